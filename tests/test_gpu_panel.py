"""The row-panel kernel (csrc/gs_panel.hip) in its three operand precisions against fp64, and the measured error of each mode:
   f32    exact fp32 MFMA                       gate 2e-5 relative (fp32 re-association only)
   bf16x3 split bf16 x3, fp32 accumulate        gate 1e-4 of the output's max  (operands represented to 2^-16)
   bf16   bf16 operands, fp32 accumulate        gate 2e-2 of the output's max  (operands rounded to 8 significant bits)
   bf16x6 EXACT 3-way bf16 split, 6 products    the fp32 gate (2e-6): the operands are represented exactly (8 + 8 + 8 bits),
          fp32 accumulate                       the dropped cross terms are <= 2^-23 of a product
Needs an MI355X."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MODES = {"f32": (0, 2e-5, 2e-6), "bf16x3": (1, 2e-4, 1e-4), "bf16": (2, 5e-2, 2e-2), "bf16x6": (3, 2e-5, 2e-6)}


@pytest.fixture(scope="module")
def E():
    from graphslim_amd import engine
    return engine


def run(E, A, W, mode, **kw):
    """C = A W^T (+...) through gs_gemm_f32 with desc.prec = mode; asserts the panel route."""
    from graphslim_amd import _lib
    M, K = A.shape
    N = W.shape[0]
    out = torch.full((M, N), float("nan"), device=DEV)
    scratch = torch.empty(N * K * 3 // 2 + 16, device=DEV)      # 4 B (modes 1, 2) or 6 B (mode 3) per element of B
    d, keep = E.gemm_desc(A, W, out, M, N, K, (K, 1), (1, K), N, prec=mode, scratch=scratch, **kw)
    assert _lib.lib().gs_gemm_route(C.byref(d)) == 5
    _lib.check(_lib.lib().gs_gemm_f32(C.byref(d), _lib.stream()))
    return out


def err(got, ref):
    ref = ref.double().cpu()
    return float((got.double().cpu() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("M,K", [(206116, 256), (169343, 128), (2048 + 5, 32), (4096, 64), (10007, 512)])
def test_panel_against_fp64(E, mode, M, K):
    code, rtol, gate = MODES[mode]
    torch.manual_seed(M + K)
    A, W, b = torch.randn(M, K), torch.randn(256, K) / np.sqrt(K), torch.randn(256)
    ref = A.double() @ W.double().T + b.double()
    out = run(E, A.to(DEV), W.to(DEV), code, bias=b.to(DEV))
    e = err(out, ref)
    print(f"panel {mode} M={M} K={K}: max|err| / max|ref| = {e:.3e}")
    assert e <= gate, (mode, M, K, e)
    if mode == "f32":
        np.testing.assert_allclose(out.double().cpu().numpy(), ref.numpy(), rtol=rtol, atol=gate * float(ref.abs().max()))


@pytest.mark.parametrize("mode", list(MODES))
def test_panel_epilogues(E, mode):
    """bias + relu, the mask epilogue (ReLU backward), fused column statistics, a row count that is not a panel multiple."""
    from graphslim_amd import _lib
    code, rtol, gate = MODES[mode]
    M, K = 5000 + 37, 256
    torch.manual_seed(3)
    A, W, b, mask = torch.randn(M, K), torch.randn(256, K) / 16, torch.randn(256), torch.randn(M, 256)
    ref = A.double() @ W.double().T
    a, w = A.to(DEV), W.to(DEV)
    out = run(E, a, w, code, bias=b.to(DEV), relu=1)
    assert err(out, (ref + b.double()).clamp_min(0)) <= gate
    out = run(E, a, w, code, mask=mask.to(DEV), ldmask=256)
    assert err(out, ref * (mask > 0)) <= gate
    d, _k = E.gemm_desc(a, w, a, M, 256, K, (K, 1), (1, K), 256)
    tiles = _lib.lib().gs_gemm_tiles_m(C.byref(d))
    assert tiles == -(-M // 32)          # fewer rows than one round of the chip: the whole product runs in 32-row tiles
    stat = torch.full((tiles, 2, 256), float("nan"), device=DEV)
    out = run(E, a, w, code, bias=b.to(DEV), colstat=stat)
    got = out.double().cpu()
    # the statistics are those of the values the kernel WROTE (whatever the operand precision)
    np.testing.assert_allclose(stat[:, 0].double().sum(0).cpu().numpy(), got.sum(0).numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(stat[:, 1].double().sum(0).cpu().numpy(), (got * got).sum(0).numpy(), rtol=1e-4, atol=1e-3)
    assert err(out, ref + b.double()) <= gate


def test_panel_narrow_tail_is_bitwise_the_wide_tiles(E):
    """The last partial round of a launch runs in 32-row tiles (4 waves side by side over the columns) instead of 128-row
    tiles: same k order per output element, so the results and the summed column statistics are bitwise / reassociation
    equal to the 128-row-tile launch; 206 116 rows = 3 full rounds of 512 tiles + 298 narrow tiles."""
    from graphslim_amd import _lib
    L = _lib.lib()
    M, K = 206116, 256
    torch.manual_seed(9)
    A, W, b = torch.randn(M, K, device=DEV), torch.randn(256, K, device=DEV) / 16, torch.randn(256, device=DEV)
    outs, stats = [], []
    for narrow in (1, 0):
        L.gs_gemm_set_panel_narrow_tail(narrow)
        try:
            d, _k = E.gemm_desc(A, W, A, M, 256, K, (K, 1), (1, K), 256)
            tiles = L.gs_gemm_tiles_m(C.byref(d))
            assert tiles == (3 * 512 + -(-(M - 3 * 512 * 128) // 32) if narrow else 1611)      # 1536 + 298
            st = torch.zeros(tiles, 2, 256, device=DEV)
            outs.append(run(E, A, W, 0, bias=b, colstat=st))
            stats.append(st.double().sum(0))
        finally:
            L.gs_gemm_set_panel_narrow_tail(1)
    assert torch.equal(outs[0], outs[1])
    np.testing.assert_allclose(stats[0].cpu().numpy(), stats[1].cpu().numpy(), rtol=1e-6)


def test_panel_f32_is_the_tile_kernel_up_to_reassociation(E):
    """Panel routing on / off: the same fp32 product from the two kernels (different k order inside a K block)."""
    from graphslim_amd import _lib
    M, K = 20000, 256
    torch.manual_seed(4)
    A, W = torch.randn(M, K, device=DEV), torch.randn(256, K, device=DEV) / 16
    o1 = run(E, A, W, 0)
    _lib.lib().gs_gemm_set_panel(0)
    try:
        o2 = torch.empty_like(o1)
        E.gemm(A, W, o2, M, 256, K, (K, 1), (1, K), 256)
    finally:
        _lib.lib().gs_gemm_set_panel(1)
    assert float((o1 - o2).abs().max()) <= 2e-5 * float(o2.abs().max())


def test_panel_needs_scratch_for_split_modes(E):
    """prec != 0 without scratch is not panel-eligible: the product falls back to the exact fp32 tile (never silently to a
    lower precision) — and ineligible shapes ignore prec."""
    from graphslim_amd import _lib
    A, W = torch.randn(4096, 256, device=DEV), torch.randn(256, 256, device=DEV)
    out = torch.empty(4096, 256, device=DEV)
    d, _ = E.gemm_desc(A, W, out, 4096, 256, 256, (256, 1), (1, 256), 256, prec=1)
    assert _lib.lib().gs_gemm_route(C.byref(d)) != 5
    _lib.check(_lib.lib().gs_gemm_f32(C.byref(d), _lib.stream()))
    ref = A.double() @ W.double().T
    assert err(out, ref) <= 2e-6


@pytest.mark.parametrize("workload", ["ogbn-arxiv", "cora-r1.0"])
@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
@pytest.mark.parametrize("param_grads", [False, True])
def test_generated_a1_operand_and_mask_are_bitwise_the_materialised_form(mode, param_grads, workload):
    """PGE forward with A1 formed in the operand path of E2 = A1 W1^T + b1 (gs_panel_fwdgen_kernel) and the [A1 > 0] mask of the
    fused marginals recomputed from U, V — against the same engine with A1 materialised by gs_pge_make_a1 (knob
    gs_gemm_set_panel_a1gen(0)): same arithmetic per element and same k order, so pair scores, d/dX~ and d/dtheta_PGE must be
    IDENTICAL, in feature epochs (dE2 generated too) and PGE epochs (A1 materialised lazily for dW1).  ogbn-arxiv: n' = 454,
    PGE width 256 (8 column tiles per wave); cora at r = 1.0: n' = 140, PGE width 128 (4 column tiles, d = 1433)."""
    import bench
    from graphslim_amd import _lib
    L = _lib.lib()
    p = bench.build_problem(workload, seed=1)
    assert len(p["labels_syn"]) >= 128
    got = {}
    for on in (1, 0):
        L.gs_gemm_set_panel_a1gen(on)
        try:
            eng = bench.make_engine(p, DEV)
            eng.set_operand_mode(mode)
            bench.init_state(p, eng)
            torch.manual_seed(123)
            bench.new_model(p, eng)
            eng.stage_sample()
            eng.stage_real_grads()
            eng.stage_pge_forward(False)
            eng.stage_syn_match()
            eng.syn_finish()
            eng.stage_pge_backward(param_grads, True)
            torch.cuda.synchronize()
            n, d = eng.feat_syn.shape
            got[on] = (eng.buffer("e3").clone(), eng.buffer("d_feat").clone(), eng.buffer("d_pge").clone())
            del eng
        finally:
            L.gs_gemm_set_panel_a1gen(1)
    assert torch.isfinite(got[1][0]).all() and got[1][1].abs().max() > 0
    for a, b, what in zip(got[1], got[0], ("pair scores e3", "d_feat", "d_pge")):
        if what == "d_pge" and not param_grads:
            continue
        assert torch.equal(a, b), (what, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.parametrize("M,K,N", [(206116, 256, 256), (169343, 128, 256), (19600, 128, 128)])
def test_exact_three_way_split_has_the_error_of_the_fp32_mfma(E, M, K, N):
    """Operand mode 3 ('bf16x6') against fp64 next to the native fp32 MFMA on the same operands, at the bench shapes: the split
    represents every fp32 operand exactly (h + m + l == x, checked here on the host), the six bf16 products are exact in fp32,
    and what is dropped (Am Bl + Al Bm + Al Bl) is <= 2^-23 of a product — so the result must be as close to fp64 as the fp32
    pipe's (whose own rounding is 2^-24 per multiply-add).  Gate: max error <= 1.5 x the fp32 kernel's (+ one ulp of slack),
    mean error <= 1.5 x."""
    torch.manual_seed(M + K + N)
    A = torch.randn(M, K) * torch.exp(torch.randn(M, 1))         # rows of very different scale
    W, b = torch.randn(N, K) / np.sqrt(K), torch.randn(N)
    # the split itself: three bf16 parts reproduce the fp32 value bit for bit
    h = A.bfloat16().float()
    m = (A - h).bfloat16().float()
    l = (A - h - m).bfloat16().float()
    assert torch.equal((h + m) + l, A)
    ref = A.double() @ W.double().T + b.double()
    scale = float(ref.abs().max())
    errs = {}
    for mode in ("f32", "bf16x6"):
        out = run(E, A.to(DEV), W.to(DEV), MODES[mode][0], bias=b.to(DEV)).double().cpu()
        d = (out - ref).abs()
        errs[mode] = (float(d.max()) / scale, float(d.mean()) / scale)
    print(f"M={M} K={K} N={N}: max / mean error over max|ref|  fp32 MFMA {errs['f32'][0]:.3e} / {errs['f32'][1]:.3e}   "
          f"3-way split {errs['bf16x6'][0]:.3e} / {errs['bf16x6'][1]:.3e}")
    assert errs["bf16x6"][0] <= 1.5 * errs["f32"][0] + 1.2e-7, errs
    assert errs["bf16x6"][1] <= 1.5 * errs["f32"][1], errs


# ---- round 4: the persistent 256-row kernels of csrc/gs_panel2.hip against the 128-row kernels of csrc/gs_panel.hip ----------------
@pytest.mark.parametrize("M,K", [(206116, 256), (169343, 128), (2053, 64), (4096 + 37, 256), (300, 128), (64 * 4 * 3 + 1, 512)])
def test_tall_plain_product_is_bitwise_the_128_row_kernel(E, M, K):
    """Same 16-k sub-blocks in the same order, same six products small-terms-first: the persistent 256-row kernel (full, 128-row and
    64-row tiles, ragged last rows dropped by the buffer stores' bounds check) gives bitwise the 128-row kernel's product."""
    from graphslim_amd import _lib
    L = _lib.lib()
    torch.manual_seed(M + K)
    A, W, b = torch.randn(M, K, device=DEV), torch.randn(256, K, device=DEV) / np.sqrt(K), torch.randn(256, device=DEV)
    try:
        L.gs_gemm_set_panel_tall(0)
        ref = run(E, A, W, 3, bias=b, relu=1)
        L.gs_gemm_set_panel_tall(1)
        got = run(E, A, W, 3, bias=b, relu=1)
    finally:
        L.gs_gemm_set_panel_tall(1)
    assert not torch.isnan(got).any()
    assert torch.equal(got, ref), (M, K, int((got != ref).sum()))
    want = (A.double() @ W.double().T + b.double()).clamp_min(0)
    assert err(got, want) <= 2e-6


@pytest.mark.parametrize("workload", ["ogbn-arxiv", "reddit"])
def test_tall_forward_and_marginals_against_the_128_row_kernels(workload):
    """PGE layer-2 forward with the generated operand and the fused marginals of its backward on the persistent 256-row kernels (all
    call sites: mask 16 + 15) against the 128-row kernels (off).  Both use the folded layer-1 form max(Us[c] + Vs[r], 0) in this mode,
    so E2 is bitwise equal; the BatchNorm-2 partial sums are grouped per workgroup instead of per tile (statistics equal to fp32
    summation order), and on ONE forward state the fused marginals are bitwise equal."""
    import bench
    from graphslim_amd import _lib
    L = _lib.lib()
    p = bench.build_problem(workload, seed=1)
    eng = bench.make_engine(p, DEV)
    bench.init_state(p, eng)
    bench.new_model(p, eng)
    eng.set_operand_mode("bf16x6")
    eng.outer_step(10)
    res = {}
    try:
        for tall in (0, 31):
            L.gs_gemm_set_panel_tall(tall)
            for part in (1, 2, 3):
                eng.pge_forward_part(False, part)
            torch.cuda.synchronize()
            res[tall] = (eng.buffer("E2").clone(), eng.buffer("mu2").clone(), eng.buffer("is2").clone(), eng.buffer("e3").clone())
            if tall == 0:
                eng.stage_syn_match()
                eng.pge_backward_part(False, True, 1)
                margs = []
                for tb in (0, 31):
                    L.gs_gemm_set_panel_tall(tb)
                    eng.view("ar_marg").fill_(float("nan"))
                    eng.pge_backward_part(False, True, 2)
                    torch.cuda.synchronize()
                    margs.append(eng.view("ar_marg").clone())
                n, ph = eng.shapes.n_syn, eng.shapes.pge_hidden
                a, b = margs[0][:2 * n * ph], margs[1][:2 * n * ph]
                assert not torch.isnan(b).any() and torch.equal(a, b), int((a != b).sum())
    finally:
        L.gs_gemm_set_panel_tall(1)
    (e0, mu0, is0, s0), (e1, mu1, is1, s1) = res[0], res[31]
    assert torch.equal(e0, e1), int((e0 != e1).sum())
    assert float(((mu0 - mu1).abs() / (mu0.abs() + 1e-3)).max()) <= 1e-5 and float(((is0 - is1).abs() / is0.abs()).max()) <= 1e-6
    assert float((s0 - s1).abs().max()) <= 1e-5 * float(s0.abs().max())
