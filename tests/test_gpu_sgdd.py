"""SGDD on the GPU (graphslim/condensation/sgdd.py, models/ignr.py): the IGNR structure generator against fp64 autograd of the
oracle's restatement, the optimal-transport regulariser, one outer iteration against the oracle, the replay of the reference's own
run (golden G20) and the drop-in reduce() API.  Needs an MI355X."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import gcond_oracle as O
from oracle import sgdd_oracle as S
from helpers import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ignr_case(n=14, d=10, mx=9, seed=0):
    torch.manual_seed(seed)
    p = S.init_ignr(d, n, mx)
    for t in "cx":
        for l in range(2):
            p[f"{t}{l}_g"] = 1 + 0.3 * torch.randn(128)
            p[f"{t}{l}_be"] = 0.2 * torch.randn(128)
    return p, torch.randn(n, d)


@pytest.mark.parametrize("n,d", [(14, 10), (37, 6)])
def test_ignr_generator_against_fp64_autograd(n, d):
    """Training-mode structure, every parameter gradient and d / d X~ for a random upstream gradient; the running statistics after
    the pass; the eval-mode (inference) structure from them."""
    from graphslim_amd.structure import IGNRGenerator, IGNR_ORDER
    p, x = _ignr_case(n, d)
    gen = IGNRGenerator(p, d, n, 0.4, 0.01, DEV)
    xd = x.to(DEV).contiguous()
    adj = gen.forward(xd, train=True).cpu()
    p64 = {k: v.double().requires_grad_(True) for k, v in p.items()}
    x64 = x.double().requires_grad_(True)
    run = {k: [v[0].double(), v[1].double()] for k, v in S.new_running().items()}
    ref = S.ignr_structure(x64, p64, run, 0.4, train=True)
    np.testing.assert_allclose(adj.numpy(), ref.detach().numpy(), atol=3e-6)
    assert torch.equal(adj, adj.T) and float(adj.diagonal().abs().max()) == 0.0
    up = torch.randn(n, n)
    names = [k for k in IGNR_ORDER if k not in ("P", "out_w", "out_b")]
    grads = torch.autograd.grad((ref * up.double()).sum(), [p64[k] for k in names] + [x64])
    gflat, dx = gen.backward(up.to(DEV).contiguous())
    got = gen.views(gflat)
    wmax = max(float(g.abs().max()) for g in grads[:-1])
    for k, b in zip(names, grads[:-1]):
        a = got[k].cpu()
        b = b.float().reshape(a.shape)
        if k.endswith("_b") and k[1] in "01":       # biases in front of a batch-statistics BatchNorm: the exact gradient is zero
            assert float(a.abs().max()) <= 1e-4 * max(1.0, wmax), (k, float(a.abs().max()))
            continue
        np.testing.assert_allclose(a.numpy(), b.numpy(), atol=3e-4 * float(b.abs().max()) + 1e-7, err_msg=k)
    np.testing.assert_allclose(dx.cpu().numpy(), grads[-1].float().numpy(), atol=3e-4 * float(grads[-1].abs().max()))
    for key in ("c0", "c1", "x0", "x1"):
        np.testing.assert_allclose(gen.run[key][0].cpu().numpy(), run[key][0].float().numpy(), atol=1e-5)
        np.testing.assert_allclose(gen.run[key][1].cpu().numpy(), run[key][1].float().numpy(), rtol=1e-4, atol=1e-6)
    adj_eval = gen.forward(xd, train=False).cpu()
    with torch.no_grad():
        ref_eval = S.ignr_structure(x64.detach(), {k: v.detach() for k, v in p64.items()}, run, 0.4, train=False)
    np.testing.assert_allclose(adj_eval.numpy(), ref_eval.numpy(), atol=1e-5)
    assert float((adj_eval - adj).abs().max()) > 1e-3          # the two BatchNorm modes give different structures


@pytest.mark.parametrize("n,mx,seed", [(14, 9, 0), (37, 40, 1), (60, 25, 2)])
def test_transport_regulariser_against_fp64_autograd(n, mx, seed):
    """opt_loss (ignr.py:184-209) and its gradient wrt the structure and the transport plan: closed form in the eigenbasis
    (rocSOLVER ssyevd + gs_gemm_f32 + spectrum kernels) against fp64 autograd through the reference's own formula — torch.svd,
    U f(S) V^T, torch.linalg.eigh — on a structure the IGNR produces (symmetric, zero diagonal, entries in (0, 1))."""
    from graphslim_amd.structure import TransportRegulariser, pinv_corner
    torch.manual_seed(seed)
    raw = torch.rand(n, n, dtype=torch.float64)
    adj = torch.sigmoid(2 * (raw + raw.T) - 2)
    adj = adj - torch.diag(torch.diag(adj))
    c = (torch.rand(mx, mx) < 0.3).double()
    corner = torch.triu(c, 1) + torch.triu(c, 1).T + 3 * torch.eye(mx, dtype=torch.float64)      # full rank: a well-defined pseudo-inverse
    Lx_inv = S.mx_inv(corner)
    np.testing.assert_allclose(pinv_corner(corner.numpy()), S.mx_inv(corner.float()).numpy(), atol=1e-5)
    P = torch.rand(mx, n, dtype=torch.float64)
    a64, p64 = adj.clone().requires_grad_(True), P.clone().requires_grad_(True)
    val = S.opt_loss(a64, p64, Lx_inv)
    ga, gp = torch.autograd.grad(val, [a64, p64])
    ga = (ga + ga.T) / 2                                        # the structure only moves along symmetric directions
    reg = TransportRegulariser(Lx_inv.float(), n, DEV)
    d_adj = torch.zeros(n, n, device=DEV)
    dP = torch.zeros(mx, n, device=DEV)
    scale = 1e-3
    reg.forward_backward(adj.float().to(DEV).contiguous(), P.float().to(DEV).contiguous(), scale, d_adj, dP)
    assert abs(float(reg.value) - float(val)) <= 2e-3 * abs(float(val)), (float(reg.value), float(val))
    got_a = d_adj.cpu().double() / scale
    got_a = (got_a + got_a.T) / 2
    assert float((got_a - ga).abs().max()) <= 2e-2 * float(ga.abs().max()), (float((got_a - ga).abs().max()), float(ga.abs().max()))
    got_p = dP.cpu().double() / scale
    assert float((got_p - gp).abs().max()) <= 2e-2 * float(gp.abs().max()), (float((got_p - gp).abs().max()), float(gp.abs().max()))
    assert float(reg.acc) == pytest.approx(float(reg.value), rel=1e-6)


def _golden_setup(g, tag, sizes=(4, 3), dropout=0.0, seed=1):
    from graphslim_amd import host as H
    from graphslim_amd.condensation.sgdd import SGDDLoop
    from graphslim_amd.engine import EngineShapes, GCondEngine
    n, nclass, d, hidden, t, inner, epochs, outer, mx = [int(v) for v in g[f"{tag}_meta"]]
    a = sp.coo_matrix((np.ones(len(g[f"{tag}_a_row"])), (g[f"{tag}_a_row"], g[f"{tag}_a_col"])), shape=(n, n)).tocsr()
    x, y, idx_train = g[f"{tag}_x"], g[f"{tag}_y"], g[f"{tag}_idx_train"]
    labels_train = y[idx_train]
    scale = float(g[f"{tag}_opt_scale"])
    H.seed_everything(1)
    labels_syn, ncd, ranges = H.generate_labels_syn(labels_train, 0.5)
    assert np.array_equal(labels_syn, g[f"{tag}_labels_syn"])
    params = H.init_ignr_params(d, len(labels_syn), mx)
    sel = H.random_init_indices(idx_train, labels_train, ncd)
    sh = EngineShapes(n, d, nclass, len(labels_syn), t, hidden, 4, 32, sizes[0], sizes[-1], fmid=tuple(sizes[1:-1]), dropout=dropout)
    eng = GCondEngine(sh, torch.from_numpy(x), None, None, None, labels_syn, [tuple(ranges[c]) for c in range(nclass)], lr_feat=0.01,
                      lr_adj=0.01, lr_model=0.01, inner_loop=inner, seed=seed, external_blocks=True, device=DEV)
    eng.set_feat(torch.from_numpy(x[sel]))
    loop = SGDDLoop(eng, params, g[f"{tag}_Lx_inv"], 0.5, scale, 0.01, 0.01)
    H.init_sgc_params(d, hidden, nclass, t)                       # model constructor draw
    return dict(a=a, x=x, y=y, idx_train=idx_train, labels_train=labels_train, csr=H.normalize_adj_sparse(a), labels_syn=labels_syn,
                ncd=ncd, params=params, sel=sel, eng=eng, loop=loop, scale=scale, meta=(n, nclass, d, hidden, t, inner, epochs, outer, mx))


def _blocks(s, calls, sizes=(4, 3)):
    from graphslim_amd.engine import BlockSet
    nclass = s["meta"][1]
    batches = []
    for c in range(nclass):
        rng = np.random.default_rng(100000 + calls + c)
        batches.append(O.sample_blocks(*s["csr"], rng.permutation(s["idx_train"][s["labels_train"] == c])[:256], list(sizes), rng))
    return batches, BlockSet.from_batches(batches, 32, sizes[0], sizes[-1], DEV, fmid=tuple(sizes[1:-1]))


@pytest.mark.parametrize("tag,epoch,sizes", [("t1", 0, (4, 3)), ("t1", 10, (4, 3)), ("t2", 0, (4, 3)), ("t2", 10, (4, 3)),
                                             ("t1", 0, (4, 3, 2)), ("t2", 10, (4, 3, 2)), ("t2", 0, (3, 2, 2, 2))])
def test_one_outer_iteration_against_the_oracle(golden_dir, tag, epoch, sizes):
    """sgdd.py:64-95 once (a structure epoch and a feature epoch): matching loss, regulariser value, d total / d X~ and d total / d theta
    of the IGNR incl. the transport plan, the inference structure, the model after the inner steps.  Round 6: also with `--nlayers` 3 / 4
    (three / four sampled hops on the real side, Q = A^^K of the IGNR's structure on the synthetic side and in the inner steps: the
    oracle's K-hop train_class is pinned by golden G24, its SGDD loop by G20)."""
    from graphslim_amd import host as H
    g = load_golden(golden_dir, "g20_sgdd")
    s = _golden_setup(g, tag, sizes)
    n, nclass, d, hidden, t, inner, epochs, outer, mx = s["meta"]
    eng, loop = s["eng"], s["loop"]
    model = H.init_sgc_params(d, hidden, nclass, t)
    eng.set_model(model)
    st = S.SGDDState(torch.from_numpy(s["x"][s["sel"]]), s["params"], torch.from_numpy(s["labels_syn"]), s["ncd"], nclass,
                     torch.from_numpy(g[f"{tag}_Lx_inv"]), 0.01, 0.01, 0.01, 0.5, s["scale"], nlayers=len(sizes))
    st.new_epoch(model)
    batches, blocks = _blocks(s, 0, sizes)
    eng.set_blocks(blocks)
    loop.step(epoch)
    ml, reg, g_x, g_p, adj_in = st.outer_iteration(epoch, batches, torch.from_numpy(s["x"]), torch.from_numpy(s["y"]).long(), inner)
    assert abs(float(eng.buffer("loss")[0]) - float(ml)) <= 1e-4 * abs(float(ml))
    assert abs(float(loop.reg.value) - float(reg)) <= 1e-3 * abs(float(reg)), (float(loop.reg.value), float(reg))
    got = {k: v.cpu() for k, v in loop.gen.views(loop.gen.grad).items()}
    # the transport plan's gradient is the regulariser's alone; fp32 autograd through torch.svd's backward (1 / (s_i^2 - s_j^2) terms)
    # is not a usable reference for it: compare with fp64 autograd of the same formula at the same structure
    p64 = {k: v.double() for k, v in s["params"].items()}
    P64 = p64["P"].clone().requires_grad_(True)
    adj64 = S.ignr_structure(torch.from_numpy(s["x"][s["sel"]]).double(), p64, {k: [v[0].double(), v[1].double()] for k, v in S.new_running().items()}, 0.5, True)
    gP = torch.autograd.grad(s["scale"] * S.opt_loss(adj64, P64, torch.from_numpy(g[f"{tag}_Lx_inv"]).double()), P64)[0]
    g_p = dict(g_p, P=gP.float())
    for k, b in g_p.items():
        if k in ("out_w", "out_b") or (k.endswith("_b") and k[1] in "01") or (epoch % 50 >= 10 and k != "P"):
            continue          # the unused head; zero-gradient biases in front of batch-statistics BatchNorm; feature epochs skip the unread parameter gradients
        tol = 3e-2 if k == "P" else 2e-3      # P's gradient is the regulariser's alone: fp32 eigen-decompositions of ill-conditioned spectra
        err = float((got[k] - b.reshape(got[k].shape)).abs().max())
        assert err <= tol * float(b.abs().max()) + 1e-12, (k, err, float(b.abs().max()))
    if epoch % 50 >= 10:
        assert float((loop.gen.dX.cpu() - g_x).abs().max()) <= 2e-3 * float(g_x.abs().max())
        np.testing.assert_allclose(eng.feat_syn.cpu().numpy(), st.feat_syn.detach().numpy(), atol=2e-5)
        np.testing.assert_allclose(loop.adj_inner.cpu().numpy(), adj_in.numpy(), atol=2e-5)


@pytest.mark.parametrize("epoch", [0, 10])
def test_one_outer_iteration_with_dropout_against_the_oracle(golden_dir, epoch):
    """`--dropout` with `-M sgdd` (round 6): sgdd.py:62 puts the condensation model into training mode like GCond — golden G30 counts
    upstream's masks (2 x nclass per train_class + one per inner step) — so the engine's dropout stage runs under the IGNR's structure:
    two iterations of sgdd.py:64-95 (the loop advances the masks' counter) with the engine's masks handed to the oracle — matching loss,
    d total / d X~, the IGNR's gradients, the updated tensors, the model after the inner steps under their own masks."""
    from graphslim_amd import host as H
    from graphslim_amd.engine import dropout_keep
    g, tag, p_drop, seed = load_golden(golden_dir, "g20_sgdd"), "t2", 0.4, 5
    s = _golden_setup(g, tag, dropout=p_drop, seed=seed)
    n, nclass, d, hidden, t, inner, epochs, outer, mx = s["meta"]
    eng, loop = s["eng"], s["loop"]
    model = H.init_sgc_params(d, hidden, nclass, t)
    eng.set_model(model)
    st = S.SGDDState(torch.from_numpy(s["x"][s["sel"]]), s["params"], torch.from_numpy(s["labels_syn"]), s["ncd"], nclass,
                     torch.from_numpy(g[f"{tag}_Lx_inv"]), 0.01, 0.01, 0.01, 0.5, s["scale"])
    st.new_epoch(model)

    def masks(iteration):
        def hook(where, x):
            kind, index, rows = where[:3]
            rows = np.arange(x.shape[0]) if rows is None else np.asarray(rows)
            keep = dropout_keep(seed, iteration, kind, index, rows, x.shape[1], p_drop)
            return torch.from_numpy(keep).to(x.dtype) * torch.tensor(np.float32(1.0 / (1.0 - p_drop))).to(x.dtype)
        return hook

    losses = []
    try:
        for it in range(2):
            batches, blocks = _blocks(s, it * nclass)
            eng.set_blocks(blocks)
            loop.step(epoch)
            assert eng.dropout_iter() == it
            O.DROPOUT = masks(it)
            ml, reg, g_x, g_p, adj_in = st.outer_iteration(epoch, batches, torch.from_numpy(s["x"]), torch.from_numpy(s["y"]).long(), inner)
            O.DROPOUT = None
            losses.append(float(ml))
            assert abs(float(eng.buffer("loss")[0]) - float(ml)) <= 1e-4 * abs(float(ml)), (it, float(eng.buffer("loss")[0]), float(ml))
            got = {k: v.cpu() for k, v in loop.gen.views(loop.gen.grad).items()}
            for k, b in g_p.items():
                if k in ("out_w", "out_b", "P") or (k.endswith("_b") and k[1] in "01") or epoch % 50 >= 10:
                    continue
                err = float((got[k] - b.reshape(got[k].shape)).abs().max())
                assert err <= 2e-3 * float(b.abs().max()) + 1e-12, (it, k, err, float(b.abs().max()))
            if epoch % 50 >= 10:
                assert float((loop.gen.dX.cpu() - g_x).abs().max()) <= 2e-3 * float(g_x.abs().max())
                np.testing.assert_allclose(eng.feat_syn.cpu().numpy(), st.feat_syn.detach().numpy(), atol=3e-5)
            # the oracle continues from the engine's state, so that the second iteration is compared on identical inputs
            with torch.no_grad():
                st.feat_syn.copy_(eng.feat_syn.cpu())
                for k, v in loop.gen.views().items():
                    st.p[k].copy_(v.cpu().reshape(st.p[k].shape))
            for a, b in zip(eng.get_model(), st.model):
                np.testing.assert_allclose(a.cpu().numpy(), b.detach().numpy(), atol=3e-4 if epoch % 50 >= 10 else 5e-3)
                with torch.no_grad():
                    b.copy_(a.cpu())
    finally:
        O.DROPOUT = None
    # without the masks the oracle lands elsewhere: the comparison gates the dropout path
    s0 = _golden_setup(g, tag)
    s0["eng"].set_model(model)
    batches, blocks = _blocks(s0, 0)
    s0["eng"].set_blocks(blocks)
    s0["loop"].step(epoch)
    assert abs(float(s0["eng"].buffer("loss")[0]) - losses[0]) > 1e-3 * abs(losses[0])


@pytest.mark.parametrize("tag", ["t1", "t2"])
def test_replay_of_the_reference_sgdd_run(golden_dir, tag):
    """Golden G20: the reference's own SGDD.__init__ + reduce (12 epochs x 2 outer iterations: 10 structure epochs + 2 feature
    epochs) replayed on the HIP engine + IGNR + regulariser with the reference's RNG draw order and the same deterministic blocks."""
    from graphslim_amd import host as H
    g = load_golden(golden_dir, "g20_sgdd")
    s = _golden_setup(g, tag)
    n, nclass, d, hidden, t, inner, epochs, outer, mx = s["meta"]
    eng, loop = s["eng"], s["loop"]
    calls, losses, regs = 0, [], []
    for it in range(epochs):
        eng.set_model(H.init_sgc_params(d, hidden, nclass, t))
        for _ in range(outer):
            _, blocks = _blocks(s, calls)
            calls += nclass
            eng.set_blocks(blocks)
            loop.step(it)
            losses.append(float(eng.buffer("loss")[0]))
            regs.append(float(loop.reg.value))
    np.testing.assert_allclose(np.array(losses), g[f"{tag}_losses"], rtol=5e-3)
    np.testing.assert_allclose(np.array(regs), g[f"{tag}_regs"], rtol=3e-2)
    dx = np.abs(eng.feat_syn.cpu().numpy() - g[f"{tag}_feat_syn"])
    assert (dx > 5e-4 + 1e-3 * np.abs(g[f"{tag}_feat_syn"])).mean() <= 0.02 and dx.max() <= 0.02, ((dx > 5e-4).mean(), dx.max())
    np.testing.assert_allclose(loop.adj_inner.cpu().numpy(), g[f"{tag}_adj"], atol=8e-3)
    for i, key in enumerate(["c0", "c1", "x0", "x1"]):
        rv = g[f"{tag}_rv{i}"]
        drv = np.abs(loop.gen.run[key][1].cpu().numpy() - rv) / (np.abs(rv) + 0.05)
        assert (drv > 3e-2).mean() <= 0.03 and drv.max() <= 0.2, (key, (drv > 3e-2).mean(), drv.max())


def _full_size_sgdd(dataset, rate, tmp_path):
    """SGDD at a product configuration (configs/sgdd/<dataset>.json) on the dataset-shaped synthetic graph: the product's own setup
    (SGDD.prepare: engine with the HIP sampler, IGNR, regulariser with the pseudo-inverse of the real corner) and the oracle's state
    from the same initial values."""
    import graphslim  # noqa: F401
    from graphslim.config import cli
    from graphslim.condensation import SGDD
    from graphslim.dataset import get_dataset
    from graphslim_amd import host as H
    args = cli(["-D", dataset, "-M", "sgdd", "-R", str(rate), "-G", "0", "--save_path", str(tmp_path)], standalone_mode=False)
    graph = get_dataset(args.dataset, args)
    H.seed_everything(args.seed)
    agent = SGDD(setting=args.setting, data=graph, args=args)
    eng, loop = agent.prepare(graph)
    Lx = loop.reg.Lx_inv.cpu()
    st = S.SGDDState(eng.feat_syn.cpu().clone(), {k: v.cpu().clone() for k, v in agent.ignr_init.items()},
                     torch.from_numpy(np.asarray(graph.labels_syn)).long(), agent.num_class_dict if hasattr(agent, "num_class_dict") else graph.num_class_dict,
                     graph.nclass, Lx, args.lr_feat, args.lr_adj, args.lr, args.ep_ratio, args.opt_scale)
    return args, graph, agent, eng, loop, st


@pytest.mark.parametrize("dataset,rate", [("cora", 0.5), ("ogbn-arxiv", 0.005)])
def test_full_size_iterations_against_the_oracle(tmp_path, dataset, rate):
    """VERDICT r02 "next" 3: SGDD at the sizes that matter (cora r = 0.5: n' = 70, corner 200; ogbn-arxiv r = 0.5 %: n' = 454, 206 k
    pairs, corner 1 000) against oracle/sgdd_oracle.py — the literal restatement of sgdd.py:64-95 / ignr.py:184-209 with torch.svd and
    torch.linalg.eigh in fp32 — on the blocks the HIP sampler drew.  First iteration (identical state): matching loss 1e-4, regulariser
    value 1e-3, the regulariser's own structure gradient against fp64 autograd of upstream's formula.  Then both sides run the same
    structure-epoch schedule and the test answers the question the r02 whole job left open (40 % accuracy with the shipped
    opt_scale): does upstream's formula, evaluated by the oracle from the same state on the same blocks, saturate the structure the
    same way?  It must: mean and extremes of the learned structure stay together, and the regulariser values stay within 2 %."""
    import bench
    from graphslim_amd import host as H
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    args, graph, agent, eng, loop, st = _full_size_sgdd(dataset, rate, tmp_path)
    d, nclass = graph.feat_full.shape[1], graph.nclass
    X, Y = graph.feat_full, graph.labels_full
    iters = 6
    theta = H.init_sgc_params(d, args.hidden, nclass, args.ntrans)
    eng.set_model(theta)
    st.new_epoch([t.clone() for t in theta])
    rows = []
    for k in range(iters):
        adj0 = loop.gen.forward(eng.feat_syn, train=False).clone() if k == 0 else None      # (eval-mode pass: no running-statistics update)
        loop.step(0)                                                                        # a structure epoch (it % 50 < 10)
        batches = bench.blocks_to_batches(eng.get_blocks())
        ml, reg, _, g_p, adj_in = st.outer_iteration(0, batches, X, Y, args.inner_loop)
        torch.cuda.synchronize()
        a_e, a_o = loop.adj_inner.cpu(), adj_in
        rows.append(dict(ml_e=float(eng.buffer("loss")[0]), ml_o=float(ml), reg_e=float(loop.reg.value), reg_o=float(reg),
                         mean_e=float(a_e.mean()), mean_o=float(a_o.mean()), max_e=float(a_e.max()), max_o=float(a_o.max()),
                         dadj=float((a_e - a_o).abs().max())))
        if k == 0:
            # the regulariser's structure gradient (left in reg.M) against fp64 autograd of upstream's formula at the same structure
            adj64 = S.ignr_structure(eng.feat_syn.cpu().double(), {q: v.double() for q, v in agent.ignr_init.items()},
                                     {q: [v[0].double(), v[1].double()] for q, v in S.new_running().items()}, args.ep_ratio, True)
            a64 = adj64.detach().clone().requires_grad_(True)
            val64 = S.opt_loss(a64, agent.ignr_init["P"].double(), loop.reg.Lx_inv.cpu().double())
            g64 = torch.autograd.grad(args.opt_scale * val64, a64)[0]
            g64 = 0.5 * (g64 + g64.t())                                   # (the structure is symmetric: both triangles carry the gradient)
            rows[0]["val64"] = float(val64)
            # (fp64 autograd of the clamp(min=2e-20) / "zero the smallest singular values" branches is a different sub-gradient on the
            # structurally zero part of the spectrum — DESIGN 7b — so the structure gradient is compared THROUGH the generator: every
            # IGNR parameter gradient of the total loss against the oracle's autograd, like the small-size test)
            got = {q: v.cpu() for q, v in loop.gen.views(loop.gen.grad).items()}
            worst = ("", 0.0)
            for q, b in g_p.items():
                if q in ("out_w", "out_b", "P") or (q.endswith("_b") and q[1] in "01"):
                    continue
                e = float((got[q] - b.reshape(got[q].shape)).abs().max() / (b.abs().max() + 1e-30))
                worst = max(worst, (q, e), key=lambda t: t[1])
            rows[0]["grad_worst"] = worst
    for r in rows:
        print({k: (f"{v:.6g}" if isinstance(v, float) else v) for k, v in r.items()})
    bad_eig, clamped = loop.reg.read_health()
    print(f"sgdd full size [{dataset}]: n' = {eng.shapes.n_syn}, corner {agent.mx}, rocSOLVER non-converged {bad_eig}, range-guard entries {clamped}")
    r0 = rows[0]
    assert abs(r0["ml_e"] - r0["ml_o"]) <= 1e-4 * abs(r0["ml_o"]), r0
    assert abs(r0["reg_e"] - r0["reg_o"]) <= 1e-3 * abs(r0["reg_o"]), r0
    assert abs(r0["reg_e"] - r0["val64"]) <= 1e-3 * abs(r0["val64"]), r0
    assert bad_eig == 0
    assert r0["grad_worst"][1] <= 2e-3, r0
    for k, r in enumerate(rows):
        assert abs(r["ml_e"] - r["ml_o"]) <= 5e-3 * abs(r["ml_o"]), r
        # (the regulariser is an ill-conditioned function of the structure at the arxiv corner — 1e14 reciprocals — so its VALUE
        # amplifies the 1e-3 separation of the structures; the saturation itself is what the two sides must share)
        assert abs(r["reg_e"] - r["reg_o"]) <= (2e-2 if k < 3 else 0.15) * abs(r["reg_o"]), r
        assert abs(r["mean_e"] - r["mean_o"]) <= 2e-2 and abs(r["max_e"] - r["max_o"]) <= 5e-2, r


@pytest.mark.parametrize("nlayers", [2, 3])
def test_sgdd_reduce_api(tmp_path, nlayers):
    """`-M sgdd` end to end through the reference's entry points with the shipped configs/sgdd/cora.json (opt_scale 1e-11, ep_ratio
    0.5, mx_size 200): cli -> get_dataset -> SGDD(...).reduce (checkpoints incl. intermediate evaluation) -> Evaluator.evaluate."""
    import graphslim  # noqa: F401
    from graphslim.config import cli
    from graphslim.condensation import SGDD
    from graphslim.dataset import get_dataset
    from graphslim.evaluation.eval_agent import Evaluator
    from graphslim_amd.host import seed_everything
    args = cli(["-D", "cora", "-M", "sgdd", "-R", "0.25", "-G", "0", "--epochs", "20", "--outer_loop", "2", "--inner_loop", "2",
                "--eval_epochs", "40", "--run_eval", "2", "--nlayers", str(nlayers), "--save_path", str(tmp_path)], standalone_mode=False)
    assert args.opt_scale == 1e-11 and args.ep_ratio == 0.5 and args.mx_size == 200
    # the JSON's epochs = 600 fixed the checkpoint schedule before the explicit --epochs 20 was applied (config.py:231-234,385-389,
    # upstream too): name the checkpoints of the short run
    args.checkpoints = [9, 19]
    args.run_inter_eval = 1
    graph = get_dataset(args.dataset, args)
    seed_everything(args.seed)
    agent = SGDD(setting=args.setting, data=graph, args=args)
    reduced = agent.reduce(graph, verbose=False)
    n = reduced.feat_syn.shape[0]
    adj = reduced.adj_syn.cpu()
    assert tuple(adj.shape) == (n, n) and torch.isfinite(adj).all() and torch.equal(adj, adj.T)
    assert 0 <= float(adj.min()) and float(adj.max()) < 1 and float(adj.diagonal().abs().max()) == 0
    assert os.path.exists(os.path.join(str(tmp_path), "reduced_graph", "sgdd", "adj_cora_0.25_1.pt"))
    P = agent.loop.gen.views()["P"].cpu()
    assert tuple(P.shape) == (200, n) and float((P - agent.ignr_init["P"]).abs().max()) > 0      # the transport plan was trained
    mean, _ = Evaluator(args).evaluate(reduced, model_type="GCN", verbose=False)
    # 40 Adam steps of size 1e-4 from a dense sigmoid structure: still the noisy early phase of this config (upstream runs 600
    # epochs; whole-job accuracy: profiles/r02/full_runs_sgdd.txt) — only sanity here, parity is covered by the tests above
    assert 0.0 <= mean <= 1.0


@pytest.mark.parametrize("method", ["msgc", "sgdd"])
def test_inductive_setting_runs_end_to_end(tmp_path, method):
    """The inductive branch (train sub-graph, local seed ids, sgdd.py:41-43 / msgc.py:71-73; flickr-shaped synthetic graph, fan-outs
    [15, 8]) of the two structure-generator siblings through cli -> reduce -> Evaluator with the shipped flickr configs."""
    import graphslim  # noqa: F401
    from graphslim.config import cli
    from graphslim.condensation import MSGC, SGDD
    from graphslim.dataset import get_dataset
    from graphslim.evaluation.eval_agent import Evaluator
    from graphslim_amd.host import seed_everything
    args = cli(["-D", "flickr", "-M", method, "-R", "0.001", "-G", "0", "--epochs", "12", "--outer_loop", "2", "--eval_epochs", "30",
                "--run_eval", "1", "--save_path", str(tmp_path)], standalone_mode=False)
    assert args.setting == "ind" and args.ntrans == 2
    args.checkpoints = [11]
    args.run_inter_eval = 1
    graph = get_dataset(args.dataset, args)
    seed_everything(args.seed)
    agent = {"msgc": MSGC, "sgdd": SGDD}[method](setting=args.setting, data=graph, args=args)
    reduced = agent.reduce(graph, verbose=False)
    n = int(graph.feat_train.shape[0] * 0.001) if method == "msgc" else reduced.feat_syn.shape[0]
    assert torch.isfinite(reduced.feat_syn).all() and torch.isfinite(reduced.adj_syn).all()
    assert tuple(reduced.adj_syn.shape) == ((16, n, n) if method == "msgc" else (n, n))
    mean, _ = Evaluator(args).evaluate(reduced, model_type="GCN", verbose=False)
    assert 0.0 <= mean <= 1.0
