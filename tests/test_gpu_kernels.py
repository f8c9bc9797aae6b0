"""Parity of the stateless HIP entry points (through the C ABI) against CPU references.  Needs an MI355X."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import gcond_oracle as O
from helpers import sbm_graph

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def E():
    from graphslim_amd import engine
    return engine


def ref_mm(a, b):
    return (a.double() @ b.double())


def close(got, ref, rtol=2e-5, atol_scale=2e-6):
    ref = ref.double()
    np.testing.assert_allclose(got.double().cpu().numpy(), ref.numpy(), rtol=rtol, atol=atol_scale * float(ref.abs().max() + 1e-30))


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (7, 5, 3), (33, 65, 17), (70, 7, 1433), (128, 128, 64), (300, 40, 256),
                                   (454, 454, 454), (129, 257, 100), (1, 256, 500), (256, 1, 77), (513, 33, 31)])
def test_gemm_layouts(E, M, N, K):
    torch.manual_seed(M * 131 + N * 7 + K)
    A, B = torch.randn(M, K), torch.randn(K, N)
    ref = ref_mm(A, B)
    Ad, Bd = A.to(DEV), B.to(DEV)
    At, Bt = A.T.contiguous().to(DEV), B.T.contiguous().to(DEV)
    for (a, sa), (b, sb) in [((Ad, (K, 1)), (Bd, (N, 1))), ((At, (1, M)), (Bd, (N, 1))), ((Ad, (K, 1)), (Bt, (1, K))),
                             ((At, (1, M)), (Bt, (1, K)))]:
        Cd = torch.full((M, N), float("nan"), device=DEV)
        E.gemm(a, b, Cd, M, N, K, sa, sb, N)
        close(Cd, ref)


def test_gemm_epilogues(E):
    torch.manual_seed(0)
    M, N, K = 150, 70, 45
    A, B, bias, mask, C0 = torch.randn(M, K), torch.randn(K, N), torch.randn(N), torch.randn(M, N), torch.randn(M, N)
    Cd = C0.clone().to(DEV)
    E.gemm(A.to(DEV), B.to(DEV), Cd, M, N, K, (K, 1), (N, 1), N, alpha=0.5, beta=2.0, bias=bias.to(DEV))
    close(Cd, 0.5 * ref_mm(A, B) + bias.double() + 2.0 * C0.double())
    Cd = torch.empty(M, N, device=DEV)
    E.gemm(A.to(DEV), B.to(DEV), Cd, M, N, K, (K, 1), (N, 1), N, bias=bias.to(DEV), relu=1)
    close(Cd, (ref_mm(A, B) + bias.double()).clamp_min(0))
    E.gemm(A.to(DEV), B.to(DEV), Cd, M, N, K, (K, 1), (N, 1), N, mask=mask.to(DEV), ldmask=N)
    close(Cd, ref_mm(A, B) * (mask > 0))
    one = torch.ones(1, device=DEV)     # column sums through a stride-0 operand
    out = torch.empty(1, N, device=DEV)
    E.gemm(one, B.to(DEV), out, 1, N, K, (0, 0), (N, 1), N)
    close(out, B.double().sum(0, keepdim=True))


@pytest.mark.parametrize("M,N,K", [(1000, 256, 256), (300, 70, 33), (129, 40, 64), (50, 200, 16)])
def test_gemm_fused_column_statistics(E, M, N, K):
    """colstat: per-M-tile column sums / sums of squares of the written C (BatchNorm statistics in the epilogue)."""
    import ctypes as C
    from graphslim_amd import _lib
    torch.manual_seed(M + N)
    A, B, bias = torch.randn(M, K), torch.randn(N, K), torch.randn(N)
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.nb1, d.nb2 = M, N, K, 1, 1
    tiles = _lib.lib().gs_gemm_tiles_m(C.byref(d))
    stat = torch.full((tiles, 2, N), float("nan"), device=DEV)
    Cd = torch.empty(M, N, device=DEV)
    E.gemm(A.to(DEV), B.to(DEV), Cd, M, N, K, (K, 1), (1, K), N, bias=bias.to(DEV), colstat=stat)
    ref = ref_mm(A, B.T) + bias.double()
    close(Cd, ref)
    close(stat[:, 0].double().sum(0), ref.sum(0), rtol=1e-4, atol_scale=1e-5)
    close(stat[:, 1].double().sum(0), (ref * ref).sum(0), rtol=1e-4, atol_scale=1e-5)


def test_gemm_batched_grouped_gather_splitk(E):
    torch.manual_seed(1)
    nb, M, N, K = 5, 37, 50, 90
    A, B = torch.randn(nb, M, K), torch.randn(nb, K, N)
    Cd = torch.empty(nb, M, N, device=DEV)
    E.gemm(A.to(DEV), B.to(DEV), Cd, M, N, K, (K, 1), (N, 1), N, nb1=nb, sAb1=M * K, sBb1=K * N, sCb1=M * N)
    close(Cd, torch.stack([ref_mm(A[i], B[i]) for i in range(nb)]))
    # grouped over contiguous row ranges of one matrix (per-class rows), transposed A:  out_g = S[lo:hi]^T G[lo:hi]
    n, F, Cc = 60, 21, 6
    S, G = torch.randn(n, F), torch.randn(n, Cc)
    lo = [0, 5, 5, 30, 41]
    hi = [5, 5, 30, 41, 60]
    offA = torch.tensor([l * F for l in lo], dtype=torch.int64, device=DEV)
    offB = torch.tensor([l * Cc for l in lo], dtype=torch.int64, device=DEV)
    kb = torch.tensor([h - l for l, h in zip(lo, hi)], dtype=torch.int32, device=DEV)
    out = torch.full((5, F, Cc), float("nan"), device=DEV)
    E.gemm(S.to(DEV), G.to(DEV), out, F, Cc, max(h - l for l, h in zip(lo, hi)), (1, F), (Cc, 1), Cc, nb1=5, offA=offA,
           offB=offB, Kb=kb, sCb1=F * Cc)
    close(out, torch.stack([ref_mm(S[l:h].T, G[l:h]) for l, h in zip(lo, hi)]))
    # grouped rows of the OUTPUT (Mb / offC)
    Bc = torch.randn(5, n, Cc)
    dQ = torch.full((n, n), float("nan"), device=DEV)
    offC = torch.tensor([l * n for l in lo], dtype=torch.int64, device=DEV)
    E.gemm(G.to(DEV), Bc.to(DEV), dQ, 25, n, Cc, (Cc, 1), (1, Cc), n, nb1=5, offA=offB, Mb=kb, sBb1=n * Cc, offC=offC)
    ref = torch.cat([ref_mm(G[l:h], Bc[i].T) for i, (l, h) in enumerate(zip(lo, hi))])
    close(dQ, ref)
    # gathered K (edge-wise GEMM) with split-K slabs and fixed-order reduce
    Nn, d, h, Kc = 200, 24, 40, 333
    X, Bm = torch.randn(Nn, d), torch.randn(2, Kc, h)
    idx = torch.randint(0, Nn, (2, Kc), dtype=torch.int32)
    kcount = torch.tensor([333, 100], dtype=torch.int32, device=DEV)
    nsplit, kchunk = 6, 64
    slabs = torch.full((2, nsplit, d, h), float("nan"), device=DEV)
    E.gemm(X.to(DEV), Bm.to(DEV), slabs, d, h, Kc, (1, d), (h, 1), h, gatherA_k=idx.to(DEV), sGb1=Kc, nb1=2, nb2=nsplit,
           kchunk=kchunk, Kb=kcount, sBb1=Kc * h, sCb1=nsplit * d * h, sCb2=d * h)
    out = torch.empty(2, d, h, device=DEV)
    from graphslim_amd import _lib
    _lib.check(_lib.lib().gs_reduce_slabs_batched_f32(_lib.ptr(out), d * h, _lib.ptr(slabs), d * h, nsplit, d * h, 2, 0.0,
                                                      _lib.stream()))
    ref = torch.stack([ref_mm(X[idx[b, :kc].long()].T, Bm[b, :kc]) for b, kc in enumerate([333, 100])])
    close(out, ref)


@pytest.mark.parametrize("F", [1, 7, 40, 41, 64, 256, 602, 1433])
def test_spmm_csr(E, F):
    a, x, y = sbm_graph(500, 5, 4, 8, seed=F)
    rp, col, val = O.normalize_adj_sparse(a)
    X = torch.randn(500, F)
    ref = torch.sparse.mm(O.csr_to_torch(rp, col, val, 500, torch.float64), X.double())
    i32 = lambda v: torch.as_tensor(v, dtype=torch.int32, device=DEV)
    out = E.spmm_csr(i32(rp), i32(col), torch.as_tensor(val, device=DEV), X.to(DEV))
    close(out, ref)
    # padded leading dimension (16-byte rows) must give the same numbers
    ld = (F + 3) // 4 * 4
    Xp = torch.zeros(500, ld, device=DEV)
    Xp[:, :F] = X.to(DEV)
    outp = torch.zeros(500, ld, device=DEV)
    E.spmm_csr(i32(rp), i32(col), torch.as_tensor(val, device=DEV), Xp, F=F, out=outp)
    close(outp[:, :F], ref)


def test_spmm_csr_empty_rows_and_transpose(E):
    rng = np.random.default_rng(0)
    m = sp.random(300, 200, density=0.02, random_state=rng, format="csr", dtype=np.float32)
    m = sp.vstack([m[:100], sp.csr_matrix((50, 200), dtype=np.float32), m[100:]]).tocsr()   # 50 empty rows
    X = torch.randn(200, 33)
    i32 = lambda v: torch.as_tensor(v, dtype=torch.int32, device=DEV)
    out = E.spmm_csr(i32(m.indptr), i32(m.indices), torch.as_tensor(m.data, device=DEV), X.to(DEV))
    close(out, torch.from_numpy(m.astype(np.float64).toarray()) @ X.double())
    mt = m.T.tocsr()     # autograd of spmm wrt X = the same kernel on the CSR of A^T
    dY = torch.randn(350, 33)
    dX = E.spmm_csr(i32(mt.indptr), i32(mt.indices), torch.as_tensor(mt.data, device=DEV), dY.to(DEV))
    close(dX, torch.from_numpy(m.astype(np.float64).toarray()).T @ dY.double())


def test_gather_rows(E):
    X = torch.randn(100, 37)
    idx = torch.randint(0, 100, (250,), dtype=torch.int32)
    out = E.gather_rows(X.to(DEV), idx.to(DEV))
    assert torch.equal(out.cpu(), X[idx.long()])


@pytest.mark.parametrize("n", [93, 2300])
def test_dense_gcn_norm_fwd_bwd(E, n):
    """n = 2 300 takes the large-n backward (row part + per-chunk column partials: a block-diagonal batch of structures, MSGC)."""
    torch.manual_seed(0)
    adj = torch.rand(n, n)
    adj = ((adj + adj.T) / 2).fill_diagonal_(0)
    adj[5, :] = 0
    adj[:, 5] = 0
    a64 = adj.double().requires_grad_(True)
    w = torch.randn(n, n)
    ref = O.dense_gcn_norm(a64)
    (ref * w.double()).sum().backward()
    out, r = E.dense_gcn_norm_fwd(adj.to(DEV))
    close(out, ref.detach(), rtol=1e-5)
    dadj = E.dense_gcn_norm_bwd(adj.to(DEV), r, w.to(DEV))
    close(dadj, a64.grad, rtol=1e-4, atol_scale=1e-5)


def test_adam_matches_torch(E):
    torch.manual_seed(0)
    for wd in (0.0, 5e-4):
        p = torch.randn(1000, requires_grad=True)
        opt = torch.optim.Adam([p], lr=0.01, weight_decay=wd)
        pd_ = p.detach().clone().to(DEV)
        m, v = torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
        for step in range(1, 6):
            g = torch.randn(1000)
            p.grad = g.clone()
            opt.step()
            E.adam_step(pd_, g.to(DEV), m, v, 0.01, step, weight_decay=wd)
        np.testing.assert_allclose(pd_.cpu().numpy(), p.detach().numpy(), rtol=1e-5, atol=1e-6)


def test_sampler_properties(E):
    """dataset/loader.py:187-224 semantics: <= bcap distinct seeds from the class pool, <= f neighbours per row without
    replacement, all neighbours kept when deg <= f, frontier de-duplicated with seeds first, full-graph weights carried."""
    n, C = 3000, 6
    a, x, y = sbm_graph(n, C, 4, 14, seed=3)
    rp, col, val = O.normalize_adj_sparse(a)
    rng = np.random.default_rng(0)
    idx_train = np.sort(rng.permutation(n)[:1500])
    pools = [idx_train[y[idx_train] == c] for c in range(C)]
    pools[0] = pools[0][:40]                       # a class smaller than bcap
    cls_ptr = np.concatenate([[0], np.cumsum([len(p) for p in pools])])
    cls_nodes = np.concatenate(pools)
    i32 = lambda v: torch.as_tensor(np.asarray(v), dtype=torch.int32, device=DEV)
    bcap, f1, f2 = 64, 10, 5
    blocks = E.BlockSet(C, bcap, f1, f2, DEV)
    marker = torch.full((C * n,), 0x7fffffff, dtype=torch.int32, device=DEV)
    args = (i32(rp), i32(col), torch.as_tensor(val, device=DEV), n, i32(cls_ptr), i32(cls_nodes), marker, blocks)
    E.sample_blocks(*args, 1234, 0)
    torch.cuda.synchronize()
    assert int((marker != 0x7fffffff).sum()) == 0            # scratch restored
    A = sp.csr_matrix((val, col, rp), shape=(n, n))
    nb = {v: dict(zip(A.indices[A.indptr[v]:A.indptr[v + 1]], A.data[A.indptr[v]:A.indptr[v + 1]])) for v in range(n)}
    first = {f: getattr(blocks, f).cpu().numpy().copy() for f in blocks.FIELDS}
    for c in range(C):
        B, n1 = int(first["n_seed"][c]), int(first["n1"][c])
        assert B == min(bcap, len(pools[c]))
        nid = first["nid1"][c, :n1]
        assert len(set(nid.tolist())) == n1                   # de-duplicated frontier
        assert set(nid[:B].tolist()) <= set(pools[c].tolist()) and len(set(nid[:B].tolist())) == B
        tp, te = first["tptr1"][c], first["tedge1"][c]
        for b in range(B):
            v = int(nid[b])
            w = first["val1"][c, b]
            k = int((w != 0).sum())
            assert k == min(len(nb[v]), f1) and np.all(w[k:] == 0)
            cols = first["col1"][c, b, :k]
            assert np.all(cols < n1)
            got = nid[cols]
            assert len(set(got.tolist())) == k                # without replacement
            for u, ww in zip(got, w[:k]):
                assert nb[v][int(u)] == ww                    # full-graph normalised weight, bit exact
        # CSC lists the in-edges of every frontier node in ascending slot order
        slots = {(b * f1 + j) for b in range(B) for j in range(f1) if first["val1"][c, b, j] != 0}
        assert tp[0] == 0 and tp[n1] == len(slots)
        seen = set()
        for l in range(n1):
            lst = te[tp[l]:tp[l + 1]]
            assert np.all(np.diff(lst) > 0)
            for s_ in lst:
                assert first["col1"][c, s_ // f1, s_ % f1] == l
                seen.add(int(s_))
        assert seen == slots
        for l in range(n1):
            v = int(nid[l])
            w = first["val0"][c, l]
            k = int((w != 0).sum())
            assert k == min(len(nb[v]), f2)
            got = first["col0"][c, l, :k]
            assert len(set(got.tolist())) == k
            for u, ww in zip(got, w[:k]):
                assert nb[v][int(u)] == ww
        assert np.all(first["val0"][c, n1:] == 0)
    # replayable: same (seed, step) -> identical blocks; another step -> different seeds for the big classes
    E.sample_blocks(*args, 1234, 0)
    torch.cuda.synchronize()
    for f in blocks.FIELDS:
        assert np.array_equal(getattr(blocks, f).cpu().numpy(), first[f])
    E.sample_blocks(*args, 1234, 1)
    torch.cuda.synchronize()
    assert not np.array_equal(blocks.nid1.cpu().numpy()[1], first["nid1"][1])
    # seed frequency over many draws is uniform over the pool (chi-square-ish bound)
    cnt = np.zeros(n)
    T_ = 200
    for t in range(T_):
        E.sample_blocks(*args, 99, t)
        cnt[blocks.nid1[1, :bcap].cpu().numpy()] += 1
    pool = pools[1]
    p = bcap / len(pool)
    freq = cnt[pool] / T_
    assert abs(freq.mean() - p) < 1e-9 and freq.std() < 3.5 * np.sqrt(p * (1 - p) / T_)


@pytest.mark.parametrize("fmid", [(6,), (4, 3)])
def test_sampler_middle_hops(E, fmid):
    """`--nlayers` 3 / 4 (dataset/loader.py:205-208): the hops BETWEEN hop 1 and the outermost one.  Every middle block keeps <= fm
    neighbours per node of the frontier before it (all of them when deg <= fm, no repeats, full-graph weights bit exact), its frontier
    holds the previous frontier first and then the new nodes ONCE each in the order of their first edge slot (PyG's relabelling order),
    its CSC lists every node's in-edges in ascending slot order, the outermost block has one row per node of the last frontier, the
    marker scratch is restored, and the same (seed, step) reproduces the blocks bit for bit."""
    n, C = 3000, 5
    a, x, y = sbm_graph(n, C, 4, 9, seed=5)
    rp, col, val = O.normalize_adj_sparse(a)
    rng = np.random.default_rng(1)
    idx_train = np.sort(rng.permutation(n)[:1200])
    pools = [idx_train[y[idx_train] == c] for c in range(C)]
    pools[1] = pools[1][:9]
    cls_ptr = np.concatenate([[0], np.cumsum([len(p) for p in pools])])
    cls_nodes = np.concatenate(pools)
    i32 = lambda v: torch.as_tensor(np.asarray(v), dtype=torch.int32, device=DEV)
    bcap, f1, f2 = 32, 5, 3
    blocks = E.BlockSet(C, bcap, f1, f2, DEV, fmid=fmid)
    marker = torch.full((C * n,), 0x7fffffff, dtype=torch.int32, device=DEV)
    args = (i32(rp), i32(col), torch.as_tensor(val, device=DEV), n, i32(cls_ptr), i32(cls_nodes), marker, blocks)
    E.sample_blocks(*args, 77, 3)
    torch.cuda.synchronize()
    assert int((marker != 0x7fffffff).sum()) == 0
    A = sp.csr_matrix((val, col, rp), shape=(n, n))
    nb = {v: dict(zip(A.indices[A.indptr[v]:A.indptr[v + 1]], A.data[A.indptr[v]:A.indptr[v + 1]])) for v in range(n)}
    first = {f: getattr(blocks, f).cpu().numpy().copy() for f in blocks.FIELDS}
    for c in range(C):
        rows, nid_prev = int(first["n1"][c]), first["nid1"][c, :int(first["n1"][c])]
        for m, fm in enumerate(fmid):
            ncur = int(first[f"nm{m}"][c])
            nid = first[f"nidm{m}"][c, :ncur]
            colm, valm = first[f"colm{m}"][c], first[f"valm{m}"][c]
            assert ncur >= rows and np.array_equal(nid[:rows], nid_prev) and len(set(nid.tolist())) == ncur
            order = []                                            # first-seen order of the new nodes over the slots (row-major)
            for r in range(rows):
                v = int(nid_prev[r])
                w = valm[r]
                k = int((w != 0).sum())
                assert k == min(len(nb[v]), fm) and np.all(w[k:] == 0) and np.all(colm[r, k:] == 0)
                got = nid[colm[r, :k]]
                assert len(set(got.tolist())) == k
                for u, ww in zip(got, w[:k]):
                    assert nb[v][int(u)] == ww
                    if int(u) not in set(nid_prev.tolist()) and int(u) not in order:
                        order.append(int(u))
            assert np.all(valm[rows:] == 0)
            assert order == nid[rows:].tolist()
            tp, te = first[f"tptrm{m}"][c], first[f"tedgem{m}"][c]
            slots = {(r * fm + j) for r in range(rows) for j in range(fm) if valm[r, j] != 0}
            assert tp[0] == 0 and tp[ncur] == len(slots)
            seen = set()
            for l in range(ncur):
                lst = te[tp[l]:tp[l + 1]]
                assert np.all(np.diff(lst) > 0)
                for s_ in lst:
                    assert colm[s_ // fm, s_ % fm] == l
                    seen.add(int(s_))
            assert seen == slots
            rows, nid_prev = ncur, nid
        for l in range(rows):                                      # the outermost block: one row per node of the last frontier
            v = int(nid_prev[l])
            w = first["val0"][c, l]
            k = int((w != 0).sum())
            assert k == min(len(nb[v]), f2)
            got = first["col0"][c, l, :k]
            assert len(set(got.tolist())) == k
            for u, ww in zip(got, w[:k]):
                assert nb[v][int(u)] == ww
        assert np.all(first["val0"][c, rows:] == 0)
    E.sample_blocks(*args, 77, 3)
    torch.cuda.synchronize()
    for f in blocks.FIELDS:
        assert np.array_equal(getattr(blocks, f).cpu().numpy(), first[f]), f


def test_sampler_neighbour_subsets_are_uniform(E):
    """loader.py:216-223 / torch_sparse sample_adj semantics at the level the reference defines them (distributional):
    the <= f neighbours kept for a row are a UNIFORM f-subset of its neighbours — every neighbour is kept with probability f / deg
    (chi-square over the neighbours of each row) and every PAIR of neighbours jointly with f (f - 1) / (deg (deg - 1)) (the
    signature of sampling without replacement, as opposed to independent draws) — for the hop-1 draw of the seeds and for the
    hop-2 draw of the same rows (another fan-out, another RNG stream)."""
    n, C = 1200, 2
    a, x, y = sbm_graph(n, C, 4, 30, seed=11)
    rp, col, val = O.normalize_adj_sparse(a)
    deg = np.diff(rp)
    rng = np.random.default_rng(0)
    cand = np.where(deg >= 24)[0]
    pools = [np.sort(rng.permutation(cand)[:12]), np.sort(rng.permutation(n)[:400])]      # class 0: 12 fixed seeds (pool < bcap)
    cls_ptr = np.concatenate([[0], np.cumsum([len(p) for p in pools])])
    cls_nodes = np.concatenate(pools)
    i32 = lambda v: torch.as_tensor(np.asarray(v), dtype=torch.int32, device=DEV)
    bcap, f1, f2 = 16, 10, 5
    blocks = E.BlockSet(C, bcap, f1, f2, DEV)
    marker = torch.full((C * n,), 0x7fffffff, dtype=torch.int32, device=DEV)
    args = (i32(rp), i32(col), torch.as_tensor(val, device=DEV), n, i32(cls_ptr), i32(cls_nodes), marker, blocks)
    T_ = 1500
    seeds = pools[0]
    nbr = {int(v): col[rp[v]:rp[v + 1]] for v in seeds}
    pos = {int(v): {int(u): i for i, u in enumerate(nbr[int(v)])} for v in seeds}
    cnt1 = {int(v): np.zeros(len(nbr[int(v)])) for v in seeds}
    cnt2 = {int(v): np.zeros(len(nbr[int(v)])) for v in seeds}
    pair1 = {int(v): 0 for v in seeds}
    pair2 = {int(v): 0 for v in seeds}
    for t in range(T_):
        E.sample_blocks(*args, 4242, t)
        nid = blocks.nid1[0].cpu().numpy()
        c1, v1 = blocks.col1[0].cpu().numpy(), blocks.val1[0].cpu().numpy()
        c0, v0 = blocks.col0[0].cpu().numpy(), blocks.val0[0].cpu().numpy()
        assert int(blocks.n_seed[0]) == len(seeds) and np.array_equal(nid[:len(seeds)], seeds)      # pool <= bcap: all of it, in order
        for b, v in enumerate(seeds):
            v = int(v)
            got1 = [pos[v][int(u)] for u in nid[c1[b][v1[b] != 0]]]
            got2 = [pos[v][int(u)] for u in c0[b][v0[b] != 0]]
            assert len(got1) == f1 and len(set(got1)) == f1 and len(got2) == f2 and len(set(got2)) == f2
            cnt1[v][got1] += 1
            cnt2[v][got2] += 1
            pair1[v] += (0 in got1) and (1 in got1)
            pair2[v] += (0 in got2) and (1 in got2)
    for cnt, pair, f in ((cnt1, pair1, f1), (cnt2, pair2, f2)):
        chi_tot, dof, pair_obs, pair_exp = 0.0, 0, 0.0, 0.0
        for v in cnt:
            D = len(cnt[v])
            pr = f / D
            chi = float((((cnt[v] - T_ * pr) ** 2) / (T_ * pr * (1 - pr))).sum())
            assert chi < D + 5 * np.sqrt(2 * D), (v, D, chi)           # per row: E[chi] = D (the indicators are negatively correlated)
            chi_tot += chi
            dof += D
            pair_obs += pair[v]
            pair_exp += T_ * f * (f - 1) / (D * (D - 1))
        assert abs(chi_tot - dof) < 5 * np.sqrt(2 * dof), (chi_tot, dof)
        # joint inclusion of two fixed neighbours: without replacement f(f-1)/(D(D-1)); independent draws would give (f/D)^2 (larger)
        assert abs(pair_obs - pair_exp) < 5 * np.sqrt(pair_exp), (pair_obs, pair_exp)


def test_sampler_seed_selection_two_level_path_and_ties(E):
    """the parallel seed selection must deliver exactly bcap distinct seeds whatever the key distribution.  The second
    histogram level (pools so large that the crossing key bin overflows the ranked candidate list) is forced on an ordinary pool
    through the test hook gs_sample_set_cand_cap: the blocks must be bitwise those of the one-level path (the selection is exact
    either way), every batch complete and duplicate-free."""
    from graphslim_amd import _lib
    n, C = 40000, 3
    rng = np.random.default_rng(5)
    m = 4 * n
    src, dst = rng.integers(0, n, m), rng.integers(0, n, m)
    keep = src != dst
    a = sp.coo_matrix((np.ones(keep.sum()), (src[keep], dst[keep])), shape=(n, n)).tocsr()
    a = ((a + a.T) > 0).astype(np.int64).tocsr()
    rp, col, val = O.normalize_adj_sparse(a)
    perm = rng.permutation(n)
    pools = [np.sort(perm[:30000]), np.sort(perm[30000:30300]), np.sort(perm[30300:30310])]
    cls_ptr = np.concatenate([[0], np.cumsum([len(p) for p in pools])])
    cls_nodes = np.concatenate(pools)
    i32 = lambda v: torch.as_tensor(np.asarray(v), dtype=torch.int32, device=DEV)
    bcap, f1, f2 = 256, 4, 3
    marker = torch.full((C * n,), 0x7fffffff, dtype=torch.int32, device=DEV)
    L = _lib.lib()
    out = {}
    try:
        for cap in (2048, 3, 1):
            _lib.check(L.gs_sample_set_cand_cap(cap))
            blocks = E.BlockSet(C, bcap, f1, f2, DEV)
            res = []
            for step in range(6):
                E.sample_blocks(i32(rp), i32(col), torch.as_tensor(val, device=DEV), n, i32(cls_ptr), i32(cls_nodes), marker, blocks, 77, step)
                torch.cuda.synchronize()
                res.append({f: getattr(blocks, f).cpu().numpy().copy() for f in blocks.FIELDS})
            out[cap] = res
    finally:
        _lib.check(L.gs_sample_set_cand_cap(2048))
    assert L.gs_sample_set_cand_cap(0) == -1 and L.gs_sample_set_cand_cap(4096) == -1
    for step in range(6):
        r = out[2048][step]
        for c in range(C):
            B = int(r["n_seed"][c])
            assert B == min(bcap, len(pools[c]))
            sd = r["nid1"][c, :B]
            assert len(set(sd.tolist())) == B and set(sd.tolist()) <= set(pools[c].tolist())
        for cap in (3, 1):
            for f in r:
                assert np.array_equal(out[cap][step][f], r[f]), (cap, step, f)


def test_spmm_csr_ragged_rows_and_nonfinite_row0(E):
    """Rows of very different length inside one wave (sub-wave row groups), F not a multiple of the lane tile, and a
    non-finite X[0] that no row references: padding slots must not leak 0*inf."""
    rng = np.random.default_rng(1)
    n, ncol = 257, 300
    deg = rng.integers(0, 70, n)
    deg[::7] = 0
    indptr = np.concatenate([[0], np.cumsum(deg)])
    indices = np.concatenate([rng.choice(np.arange(1, ncol), size=d, replace=False) for d in deg]).astype(np.int32)
    data = rng.normal(size=indptr[-1]).astype(np.float32)
    m = sp.csr_matrix((data, indices, indptr), shape=(n, ncol))
    i32 = lambda v: torch.as_tensor(v, dtype=torch.int32, device=DEV)
    for F in (5, 36, 40, 132, 260):
        X = torch.randn(ncol, F)
        X[0] = float("inf")
        out = E.spmm_csr(i32(indptr), i32(indices), torch.as_tensor(data, device=DEV), X.to(DEV))
        Xf = X.clone()
        Xf[0] = 0
        close(out, torch.from_numpy(m.astype(np.float64).toarray()) @ Xf.double())


def test_grouped_small_gemms_equal_sequential_launches():
    """gs_gemm_group_f32: independent latency-bound products in one launch are bitwise the products launched one by one
    (different shapes, operand layouts, a batched one with per-batch extents, epilogues)."""
    from graphslim_amd.engine import gemm_desc, gemm_group
    torch.manual_seed(0)
    n, d, h, Cc = 454, 128, 256, 40
    A = torch.randn(n, n, device=DEV); X = torch.randn(n, d, device=DEV); H = torch.randn(n, h, device=DEV)
    G = torch.randn(n, Cc, device=DEV); W = torch.randn(h, Cc, device=DEV); ones = torch.ones(n, device=DEV)
    bias = torch.randn(h, device=DEV)
    kb = torch.tensor([5, 0, 33], dtype=torch.int32, device=DEV)

    def build(outs):
        return [gemm_desc(H, G, outs[0], h, Cc, n, (1, h), (Cc, 1), Cc),                       # S^T G        (M,M)
                gemm_desc(ones, G, outs[1], 1, Cc, n, (0, 1), (Cc, 1), Cc),                    # 1^T G        (M = 1)
                gemm_desc(G, W, outs[2], n, h, Cc, (Cc, 1), (1, Cc), h, bias=bias, relu=1),    # G W^T + b    (K,K) + epilogue
                gemm_desc(X, H, outs[3], d, h, 33, (1, d), (h, 1), h, nb1=3, Kb=kb, sAb1=40 * d, sBb1=40 * h, sCb1=d * h)]

    def outs():
        return [torch.full((h, Cc), 7.0, device=DEV), torch.full((1, Cc), 7.0, device=DEV), torch.full((n, h), 7.0, device=DEV),
                torch.full((3, d, h), 7.0, device=DEV)]
    o_seq, o_grp = outs(), outs()
    for dsc, _ in build(o_seq):
        gemm_group([(dsc, None)])                     # n = 1 -> plain launch
    gemm_group(build(o_grp))
    torch.cuda.synchronize()
    for a, b in zip(o_seq, o_grp):
        assert torch.equal(a, b)
    ref = (H.double().T @ G.double()).float()
    np.testing.assert_allclose(o_grp[0].cpu().numpy(), ref.cpu().numpy(), rtol=2e-4, atol=2e-3)
    assert float(o_grp[3][1].abs().max()) == 0.0      # a batch with K = 0 writes zeros
