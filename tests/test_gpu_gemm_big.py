"""fp64 parity of the GEMM instantiations the ogbn-arxiv bench actually launches: the 128x128 block
tile with its 2x2-MFMA wave tile in every operand layout, its epilogues (bias / relu / mask / beta / fused column
statistics), the split-K slabs over a gathered K, and the generated-B (BGEN) edge product at d = 128, h = 256 — each test
first ASSERTS through gs_gemm_route that the shape is routed to that tile (code 3), so a routing change cannot silently
move the coverage back to the small tiles.  Needs an MI355X."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BIG = 3     # gs_gemm_route code of the 128x128 tile
PANEL = 5   # ... of the row-panel kernel that takes the PGE layer-2 / dY1 / H1full products since round 2
EDGE = 6    # ... of the dedicated real-side edge-product kernel (csrc/gs_edge.hip)


@pytest.fixture(scope="module")
def E():
    from graphslim_amd import engine
    return engine


@pytest.fixture()
def no_small_route():
    """Send everything to the tiled kernels for the duration of a test (gs_gemm_set_small / gs_gemm_set_panel are documented
    tuning knobs)."""
    from graphslim_amd import _lib
    L = _lib.lib()
    L.gs_gemm_set_small(0, 0.0)
    L.gs_gemm_set_panel(0)
    yield
    L.gs_gemm_set_small(8192, 1.5)
    L.gs_gemm_set_panel(1)


@pytest.fixture(params=[1, 0], ids=["panel", "tile128"])
def panel(request):
    """Run a test once with the row-panel kernel (the product path) and once with panel routing off, which sends the same
    product to the 128x128 tile (still the kernel of every non-eligible big product)."""
    from graphslim_amd import _lib
    L = _lib.lib()
    L.gs_gemm_set_panel(request.param)
    yield PANEL if request.param else BIG
    L.gs_gemm_set_panel(1)


def route(E, *a, **kw):
    from graphslim_amd import _lib
    d, _keep = E.gemm_desc(*a, **kw)
    return _lib.lib().gs_gemm_route(C.byref(d))


def close(got, ref, rtol=2e-5, atol_scale=2e-6, what=""):
    ref = ref.double().cpu()
    np.testing.assert_allclose(got.double().cpu().numpy(), ref.numpy(), rtol=rtol,
                               atol=atol_scale * float(ref.abs().max() + 1e-30), err_msg=what)


def layouts(A, B):
    """(tensor, strides) of A [M, K] and B [K, N] in both memory orders."""
    M, K = A.shape
    N = B.shape[1]
    Ad, Bd = A.to(DEV), B.to(DEV)
    At, Bt = A.T.contiguous().to(DEV), B.T.contiguous().to(DEV)
    return {"KK": ((Ad, (K, 1)), (Bt, (1, K))), "KM": ((Ad, (K, 1)), (Bd, (N, 1))),
            "MK": ((At, (1, M)), (Bt, (1, K))), "MM": ((At, (1, M)), (Bd, (N, 1)))}


@pytest.mark.parametrize("M,N,K", [(3001, 1100, 200), (2048, 1536, 256), (2500, 1300, 77), (12500, 256, 256)])
def test_big_tile_all_layouts(E, no_small_route, M, N, K):
    """Edge tiles in M and N, K not a multiple of the K step, all four (A, B) memory orders: the (K,K) BK = 32 form and the
    BK = 16 forms of gs_gemm_f32_kernel<128,128,64,64,...>."""
    torch.manual_seed(M + 3 * N + 7 * K)
    A, B = torch.randn(M, K), torch.randn(K, N)
    ref = A.double() @ B.double()
    for name, ((a, sa), (b, sb)) in layouts(A, B).items():
        Cd = torch.full((M, N), float("nan"), device=DEV)
        assert route(E, a, b, Cd, M, N, K, sa, sb, N) == BIG
        E.gemm(a, b, Cd, M, N, K, sa, sb, N)
        close(Cd, ref, what=f"layout {name}")


def test_pge_layer2_shape_exact(E, panel):
    """The product the roofline line is quoted on: E2[n'^2, Ph] = A1[n'^2, Ph] W1^T + b1 at n' = 454, Ph = 256, with the
    fused BatchNorm column statistics, and dY1 = (dE2 W1^T^T) * [A1 > 0] through the mask epilogue — against fp64."""
    from graphslim_amd import _lib
    n, ph = 454, 256
    rows = n * n
    torch.manual_seed(5)
    A1 = torch.randn(rows, ph).clamp_min(0)          # post-ReLU activations (zeros included)
    W1, b1 = torch.randn(ph, ph) / 16, torch.randn(ph)
    ref = A1.double() @ W1.double().T + b1.double()
    a, w = A1.to(DEV), W1.to(DEV)
    d, _k = E.gemm_desc(a, w, a, rows, ph, ph, (ph, 1), (1, ph), ph)
    tiles = _lib.lib().gs_gemm_tiles_m(C.byref(d))
    stat = torch.full((tiles, 2, ph), float("nan"), device=DEV)
    out = torch.full((rows, ph), float("nan"), device=DEV)
    assert route(E, a, w, out, rows, ph, ph, (ph, 1), (1, ph), ph, bias=b1.to(DEV), colstat=stat) == panel
    E.gemm(a, w, out, rows, ph, ph, (ph, 1), (1, ph), ph, bias=b1.to(DEV), colstat=stat)
    close(out, ref, what="E2")
    close(stat[:, 0].double().sum(0), ref.sum(0), rtol=1e-4, atol_scale=1e-5, what="column sums")
    close(stat[:, 1].double().sum(0), (ref * ref).sum(0), rtol=1e-4, atol_scale=1e-5, what="column sums of squares")
    # dY1 = (dE2 W1) * [A1 > 0], W1 handed over transposed so that both operands are k-contiguous (the engine's form)
    dE2 = torch.randn(rows, ph)
    W1T = W1.T.contiguous().to(DEV)
    out2 = torch.full((rows, ph), float("nan"), device=DEV)
    assert route(E, dE2.to(DEV), W1T, out2, rows, ph, ph, (ph, 1), (1, ph), ph, mask=a, ldmask=ph) == panel
    E.gemm(dE2.to(DEV), W1T, out2, rows, ph, ph, (ph, 1), (1, ph), ph, mask=a, ldmask=ph)
    close(out2, (dE2.double() @ W1.double()) * (A1 > 0), what="dY1")


def test_h1full_shape_bias_relu(E, panel):
    """H1full = relu(X W1 + b1) over all N = 169 343 nodes (d = 128 -> h = 256): the (K,K) tile with bias + relu, X rows
    padded to the engine's leading dimension."""
    N, d, h = 169343, 128, 256
    torch.manual_seed(6)
    X, W1, b1 = torch.randn(N, d), torch.randn(d, h) / 11, torch.randn(h)
    W1T = W1.T.contiguous().to(DEV)
    out = torch.full((N, h), float("nan"), device=DEV)
    assert route(E, X.to(DEV), W1T, out, N, h, d, (d, 1), (1, d), h, bias=b1.to(DEV), relu=1) == panel
    E.gemm(X.to(DEV), W1T, out, N, h, d, (d, 1), (1, d), h, bias=b1.to(DEV), relu=1)
    close(out, (X.double() @ W1.double() + b1.double()).clamp_min(0), what="H1full")


def test_big_tile_epilogues(E, no_small_route):
    torch.manual_seed(7)
    M, N, K = 3100, 1050, 96
    A, B, bias, mask, C0 = torch.randn(M, K), torch.randn(K, N), torch.randn(N), torch.randn(M, N), torch.randn(M, N)
    a, b = A.to(DEV), B.to(DEV)
    ref = A.double() @ B.double()
    Cd = C0.clone().to(DEV)
    assert route(E, a, b, Cd, M, N, K, (K, 1), (N, 1), N, alpha=0.5, beta=2.0, bias=bias.to(DEV)) == BIG
    E.gemm(a, b, Cd, M, N, K, (K, 1), (N, 1), N, alpha=0.5, beta=2.0, bias=bias.to(DEV))
    close(Cd, 0.5 * ref + bias.double() + 2.0 * C0.double(), what="alpha/beta/bias")
    Cd = torch.full((M, N), float("nan"), device=DEV)
    E.gemm(a, b, Cd, M, N, K, (K, 1), (N, 1), N, bias=bias.to(DEV), relu=1, mask=mask.to(DEV), ldmask=N)
    close(Cd, (ref + bias.double()).clamp_min(0) * (mask > 0), what="bias+relu+mask")
    # output with a leading dimension larger than N (a column block of a wider matrix)
    wide = torch.full((M, N + 40), 7.0, device=DEV)
    E.gemm(a, b, wide, M, N, K, (K, 1), (N, 1), N + 40)
    close(wide[:, :N], ref, what="ldc > N")
    assert float((wide[:, N:] - 7.0).abs().max()) == 0.0


def test_big_tile_split_k_weight_gradient(E):
    """dW1 = dE2^T A1 over K = n'^2 pair rows as split-K slabs ((M,M) layout, BK = 16) + fixed-order slab reduction."""
    from graphslim_amd import _lib
    ph, rows = 256, 454 * 454
    torch.manual_seed(8)
    dE2, A1 = torch.randn(rows, ph), torch.randn(rows, ph).clamp_min(0)
    kchunk = 1072
    nsplit = -(-rows // kchunk)
    slabs = torch.full((nsplit, ph, ph), float("nan"), device=DEV)
    kw = dict(nb2=nsplit, kchunk=kchunk, sCb2=ph * ph)
    assert route(E, dE2.to(DEV), A1.to(DEV), slabs, ph, ph, rows, (1, ph), (ph, 1), ph, **kw) == BIG
    E.gemm(dE2.to(DEV), A1.to(DEV), slabs, ph, ph, rows, (1, ph), (ph, 1), ph, **kw)
    out = torch.empty(ph, ph, device=DEV)
    _lib.check(_lib.lib().gs_reduce_slabs_f32(_lib.ptr(out), _lib.ptr(slabs), ph * ph, nsplit, ph * ph, 0.0, _lib.stream()))
    close(out, dE2.double().T @ A1.double(), rtol=1e-4, atol_scale=1e-5, what="dW1")


def _edge_case(Cl, cap1, f2, Nn, d, h, seed):
    g = torch.Generator().manual_seed(seed)
    K = cap1 * f2
    X = torch.randn(Nn, d, generator=g)
    H1 = torch.randn(Nn, h, generator=g)                     # the mask rows (pre-activation sign)
    Y1 = torch.randn(Cl, cap1, h, generator=g)
    idx = torch.randint(0, Nn, (Cl, K), generator=g, dtype=torch.int32)
    w = torch.rand(Cl, K, generator=g)
    n1 = torch.randint(cap1 // 2, cap1 + 1, (Cl,), generator=g, dtype=torch.int32)
    n1[0] = cap1
    kcount = n1 * f2
    for c in range(Cl):
        w[c, int(kcount[c]):] = 0
        idx[c, int(kcount[c]):] = 0
    return X, H1, Y1, idx, w, kcount.to(torch.int32), K


@pytest.mark.parametrize("edge_kernel,prec", [(1, 0), (0, 0), (1, 1), (1, 2)], ids=["edge-f32", "tile128-f32", "edge-bf16x3", "edge-bf16"])
def test_generated_b_edge_product_d128_h256(E, edge_kernel, prec):
    """The real-side first-layer gradient gW1_c = sum_e x_col(e) (x) (w_e [H1[col(e)] > 0] Y1[row(e)]) exactly as
    gs_gcond_stage_real_grads launches it at ogbn-arxiv sizes: gathered K, split-K slabs of 512, per-class K extents,
    B generated in the loader (BGEN) — and the same product with B materialised (plain gathered-K path)."""
    from graphslim_amd import _lib
    _lib.lib().gs_gemm_set_edge_kernel(edge_kernel)
    try:
        _edge_product_case(E, EDGE if edge_kernel else BIG, prec)
    finally:
        _lib.lib().gs_gemm_set_edge_kernel(1)


def _edge_product_case(E, want_route, prec):
    from graphslim_amd import _lib
    rtol, atol = {0: (1e-4, 1e-5), 1: (5e-4, 1e-4), 2: (1e-1, 2e-2)}[prec]
    Cl, cap1, f2, Nn, d, h = 5, 2816, 5, 20000, 128, 256
    X, H1, Y1, idx, w, kcount, K = _edge_case(Cl, cap1, f2, Nn, d, h, 9)
    kchunk = 512
    nsplit = -(-K // kchunk)
    Xd, H1d, Y1d, idxd, wd, kcd = X.to(DEV), H1.to(DEV), Y1.to(DEV), idx.to(DEV), w.to(DEV), kcount.to(DEV)
    ref = []
    for c in range(Cl):
        kc = int(kcount[c])
        rows = torch.arange(kc) // f2
        Bm = w[c, :kc, None].double() * (H1[idx[c, :kc].long()] > 0) * Y1[c, rows].double()
        ref.append(X[idx[c, :kc].long()].double().T @ Bm)
    ref = torch.stack(ref)
    common = dict(gatherA_k=idxd, sGb1=K, nb1=Cl, nb2=nsplit, kchunk=kchunk, Kb=kcd, sCb1=nsplit * d * h, sCb2=d * h)
    # generated B
    slabs = torch.full((Cl, nsplit, d, h), float("nan"), device=DEV)
    kw = dict(common, sBb1=cap1 * h, bgen_mask=H1d, bgen_ldm=h, bgen_idx=idxd, bgen_w=wd, bgen_div=f2, prec=prec)
    assert route(E, Xd, Y1d, slabs, d, h, K, (1, d), (h, 1), h, **kw) == want_route
    E.gemm(Xd, Y1d, slabs, d, h, K, (1, d), (h, 1), h, **kw)
    out = torch.empty(Cl, d, h, device=DEV)
    _lib.check(_lib.lib().gs_reduce_slabs_batched_f32(_lib.ptr(out), d * h, _lib.ptr(slabs), d * h, nsplit, d * h, Cl, 0.0,
                                                      _lib.stream()))
    close(out, ref, rtol=rtol, atol_scale=atol, what="BGEN edge product")
    if prec != 0 or want_route != BIG:
        return
    # materialised B (the path the 'mse' / 'cos' metrics take)
    Bmat = torch.zeros(Cl, K, h)
    for c in range(Cl):
        kc = int(kcount[c])
        rows = torch.arange(kc) // f2
        Bmat[c, :kc] = w[c, :kc, None] * (H1[idx[c, :kc].long()] > 0) * Y1[c, rows]
    slabs2 = torch.full((Cl, nsplit, d, h), float("nan"), device=DEV)
    kw2 = dict(common, sBb1=K * h)
    assert route(E, Xd, Bmat.to(DEV), slabs2, d, h, K, (1, d), (h, 1), h, **kw2) == BIG
    E.gemm(Xd, Bmat.to(DEV), slabs2, d, h, K, (1, d), (h, 1), h, **kw2)
    _lib.check(_lib.lib().gs_reduce_slabs_batched_f32(_lib.ptr(out), d * h, _lib.ptr(slabs2), d * h, nsplit, d * h, Cl, 0.0,
                                                      _lib.stream()))
    close(out, ref, rtol=1e-4, atol_scale=1e-5, what="gathered-K edge product")
