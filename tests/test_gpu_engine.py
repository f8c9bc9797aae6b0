"""Parity of the HIP GCond engine (through the C ABI) against the golden vectors produced by the reference and
against the CPU oracle on identical pre-sampled blocks.  Needs an MI355X.

Gates (BASELINE.json north_star / SURVEY.md §8d): matching loss within 1e-4 relative; gradients wrt X~ within 1e-3
relative (of the tensor's max), fp32 everywhere.
"""
import numpy as np
import pytest
import torch

from oracle import analytic as A
from oracle import gcond_oracle as O
from helpers import T, load_golden, rebuild_batches, sbm_graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(got, ref, rtol=1e-3, what=""):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=rtol * np.abs(ref).max() + 1e-30, err_msg=what)


def close_state(got, ref, tol, what=""):
    """Two runs of several Adam-stepped iterations whose gradients differ at rounding level (another fixed partition of the same fp32
    sums): Adam's first moves are ~ lr * sign(g), so an entry whose gradient is rounding noise may move the other way — the states
    agree to `tol` except on a small share of such entries, and those differ by a few learning rates at most."""
    diff = (torch.as_tensor(got).float().cpu() - torch.as_tensor(ref).float().cpu()).abs()
    share = float((diff > tol).double().mean())
    assert share <= 0.02 and float(diff.max()) <= 0.05, (what, share, float(diff.max()))


def make_engine(n, nclass, d, hidden, t, ph, labels_syn, ranges, x, f1, f2, bcap=32, metric="ours", inner=2, lrs=(0.01, 0.01, 0.01),
                csr=None, pools=None, external=True, seed=1, model="SGC", with_bn=False, fmid=(), dropout=0.0):
    from graphslim_amd.engine import EngineShapes, GCondEngine
    sh = EngineShapes(n, d, nclass, len(labels_syn), t, hidden, ph, bcap, f1, f2, model=model, with_bn=with_bn, fmid=tuple(fmid),
                      dropout=dropout)
    cls_ptr = cls_nodes = None
    if pools is not None:
        cls_ptr = np.concatenate([[0], np.cumsum([len(p) for p in pools])])
        cls_nodes = np.concatenate(pools)
    return GCondEngine(sh, torch.as_tensor(x), csr, cls_ptr, cls_nodes, labels_syn, ranges, lr_feat=lrs[0], lr_adj=lrs[1],
                       lr_model=lrs[2], inner_loop=inner, dis_metric=metric, seed=seed, external_blocks=external, device=DEV)


@pytest.mark.parametrize("tag,metric", [("t1", "ours"), ("t2", "ours"), ("t1mse", "mse"), ("gcn", "ours"), ("k3t1", "ours"), ("k3t2", "ours"),
                                        ("t3", "ours"), ("t4", "ours"), ("gcn3", "ours"), ("gcn4", "ours")])
def test_train_class_against_reference_golden(golden_dir, tag, metric):
    """G7: loss, d loss/d A^, d loss/d X~, d loss/d theta_PGE of the reference's train_class + backward.  G24 (tags k3*, round 5): the
    same with `--nlayers 3` — three sampled hops (a MIDDLE block with its CSC between hop 1 and the outermost block) on the real side,
    Q = A^^3 with the product rule for d A^ on the synthetic side (dataset/loader.py:205-206, models/sgc.py:45-51).  G27 (tags t3, t4,
    round 6): `--ntrans` 3 / 4 — the layers walked by csrc/gs_deep.inc (models/sgc.py:32-35)."""
    from graphslim_amd.engine import BlockSet
    fname = {"k3t1": "g24_train_class_three_hops", "k3t2": "g24_train_class_three_hops", "t3": "g27_train_class_deep", "t4": "g27_train_class_deep",
             "gcn3": "g28_train_class_gcn_layers", "gcn4": "g28_train_class_gcn_layers"}.get(tag, "g7_train_class")      # (G28, round 6: K-layer GCN)
    g = load_golden(golden_dir, fname)
    batches, (n, nclass, d, hidden, t), _ = rebuild_batches(g, tag)
    labels_syn = g[f"{tag}_labels_syn"]
    _, ncd, ranges = O.generate_labels_syn(g[f"{tag}_y"][g[f"{tag}_idx_train"]], float(g[f"{tag}_rate"]))
    ranges = [tuple(ranges[c]) for c in range(nclass)]
    ph = g[f"{tag}_pge2"].shape[0]
    sizes = [int(s) for s in g[f"{tag}_sizes"]]
    f1, f2, fmid = sizes[0], sizes[-1], tuple(sizes[1:-1])
    kind = "GCN" if tag.startswith("gcn") else "SGC"
    if kind == "GCN":
        t = 2
    eng = make_engine(n, nclass, d, hidden, t, ph, labels_syn, ranges, g[f"{tag}_x"], f1, f2, metric=metric, model=kind, fmid=fmid)
    eng.set_feat(T(g[f"{tag}_feat_syn"]))
    eng.set_pge([T(g[f"{tag}_pge{i}"]) for i in range(10)])
    theta = [T(g[f"{tag}_theta{i}"]) for i in range(2 * (len(sizes) if kind == "GCN" else t))]
    eng.set_model(theta)
    eng.set_blocks(BlockSet.from_batches(batches, 32, f1, f2, DEV, fmid=fmid))
    # real side
    eng.stage_real_grads()
    gw = eng.buffer("gw_real", (nclass, -1))
    for c in range(nclass):
        th = [p.clone().requires_grad_(True) for p in theta]
        ref, _ = O.class_real_grads(kind, th, T(g[f"{tag}_x"]), T(g[f"{tag}_y"]), batches[c], len(sizes))
        if metric == "ours" and t >= 2 and kind == "SGC":   # biases get distance 0 and D = 0 under 'ours' (utils.py:98-101): the costly
            for q in range(1, 2 * t - 2, 2):    # hidden layers' bias gradients (a pass over all sampled edges each) are not formed
                ref[q] = torch.zeros_like(ref[q])
        close(gw[c], torch.cat([r.reshape(-1) for r in ref]), 2e-4, f"gw_real class {c}")
    # PGE forward + norm
    eng.stage_pge_forward(False)
    close(eng.buffer("adj_raw", (len(labels_syn), -1)), g[f"{tag}_adj_syn"], 2e-5, "adj_syn")
    close(eng.buffer("adj_norm", (len(labels_syn), -1)), g[f"{tag}_adj_norm"], 2e-5, "adj_norm")
    # matching loss + second-order backward
    eng.stage_syn_match()
    eng.syn_finish()
    loss = float(eng.buffer("loss")[0])
    assert abs(loss - float(g[f"{tag}_loss"])) <= 1e-4 * abs(float(g[f"{tag}_loss"])), (loss, float(g[f"{tag}_loss"]))
    close(eng.buffer("d_adj_norm", (len(labels_syn), -1)), g[f"{tag}_dadj_norm"], 1e-3, "d loss / d A^")
    eng.stage_pge_backward(True, True)
    close(eng.buffer("d_feat", (len(labels_syn), d)), g[f"{tag}_dfeat"], 1e-3, "d loss / d X~")
    dp = eng.buffer("d_pge")
    o = 0
    for i, shp in enumerate(eng.pge_shapes()):
        k = int(np.prod(shp))
        ref = g[f"{tag}_dpge{i}"]
        if np.abs(ref).max() > 1e-6:        # Linear biases in front of a batch-stat BatchNorm: exactly-zero gradient
            close(dp[o:o + k].reshape(shp), ref, 2e-3, f"d theta_PGE[{i}]")
        o += k


@pytest.mark.parametrize("fname,tag,metric", [("g16_with_bn", "bn", "ours"), ("g23_with_bn_mse_cos", "bn_mse", "mse"),
                                               ("g23_with_bn_mse_cos", "bn_cos", "cos")])
def test_with_bn_against_reference_golden(golden_dir, fname, tag, metric):
    """G16 / G23 (gcond_base.py:261-285 check_bn, models/sgc.py:27-33): SGC with an eval-mode BatchNorm1d between its two transforms,
    bn.weight / bn.bias away from (1, 0).  The engine folds the affine map into the first transform (gs_bn_fold_kernel) and un-folds
    the inner-step gradients; under 'ours' (G16) it prices the W1 gradient columns at their true scale in the distance statistics,
    under 'mse' / 'cos' (G23, round 5: the 1-D BatchNorm parameters are part of those distances, condensation/utils.py:41-60) it maps
    the class gradients onto the true parameters before the distance and D back onto the folded model: real-side gradients, loss,
    d loss / d A^, d loss / d X~, d loss / d theta_PGE and the parameters after two inner steps of the reference."""
    from graphslim_amd.engine import BlockSet
    g = load_golden(golden_dir, fname)
    batches, (n, nclass, d, hidden, t), _ = rebuild_batches(g, tag)
    labels_syn = g[f"{tag}_labels_syn"]
    _, ncd, ranges = O.generate_labels_syn(g[f"{tag}_y"][g[f"{tag}_idx_train"]], float(g[f"{tag}_rate"]))
    ranges = [tuple(ranges[c]) for c in range(nclass)]
    ph = g[f"{tag}_pge2"].shape[0]
    f1, f2 = [int(s) for s in g[f"{tag}_sizes"]]
    eng = make_engine(n, nclass, d, hidden, t, ph, labels_syn, ranges, g[f"{tag}_x"], f1, f2, with_bn=True, inner=2, metric=metric)
    eng.set_feat(T(g[f"{tag}_feat_syn"]))
    eng.set_pge([T(g[f"{tag}_pge{i}"]) for i in range(10)])
    theta = [T(g[f"{tag}_theta{i}"]) for i in range(6)]
    eng.set_model(theta)
    eng.set_blocks(BlockSet.from_batches(batches, 32, f1, f2, DEV))
    eng.stage_real_grads()
    eng.stage_pge_forward(False)
    close(eng.buffer("adj_norm", (len(labels_syn), -1)), g[f"{tag}_adj_norm"], 2e-5, "adj_norm")
    eng.stage_syn_match()
    eng.syn_finish()
    loss = float(eng.buffer("loss")[0])
    assert abs(loss - float(g[f"{tag}_loss"])) <= 1e-4 * abs(float(g[f"{tag}_loss"])), (loss, float(g[f"{tag}_loss"]))
    close(eng.buffer("d_adj_norm", (len(labels_syn), -1)), g[f"{tag}_dadj_norm"], 1e-3, "d loss / d A^")
    eng.stage_pge_backward(True, True)
    close(eng.buffer("d_feat", (len(labels_syn), d)), g[f"{tag}_dfeat"], 1e-3, "d loss / d X~")
    dp = eng.buffer("d_pge")
    o = 0
    for i, shp in enumerate(eng.pge_shapes()):
        k = int(np.prod(shp))
        ref = g[f"{tag}_dpge{i}"]
        if np.abs(ref).max() > 1e-6:
            close(dp[o:o + k].reshape(shp), ref, 2e-3, f"d theta_PGE[{i}]")
        o += k
    # PGE.inference + two inner steps (neither optimiser of the outer problem is stepped, as in the golden)
    eng.stage_inner()
    close(eng.adj_syn.cpu(), g[f"{tag}_adj_inner"], 2e-5, "inference adjacency")
    for i, q in enumerate(eng.get_model()):
        close(q.cpu(), g[f"{tag}_theta_after{i}"], 2e-3, f"theta[{i}] after two inner steps")
    # BatchNorm layers exist between transforms only (models/sgc.py:27-33): one transform has none — refused, not silently ignored
    with pytest.raises(RuntimeError):
        make_engine(n, nclass, d, hidden, 1, ph, labels_syn, ranges, g[f"{tag}_x"], f1, f2, with_bn=True, metric=metric)


def engine_masks(seed, iteration, p_drop):
    """oracle DROPOUT hook (oracle/gcond_oracle.py) that hands the restatement the ENGINE's masks (graphslim_amd.engine.dropout_keep:
    the counter-based function of csrc/gs_kernels.h) — torch's own draws cannot be replayed on the device."""
    from graphslim_amd.engine import dropout_keep

    def hook(where, x):
        kind, index, rows = where[:3]
        layer = where[3] if len(where) > 3 else 0      # (the GCN model's hook names the hidden layer, oracle gcn_forward)
        rows = np.arange(x.shape[0]) if rows is None else np.asarray(rows)
        keep = dropout_keep(seed, iteration, kind, index, rows, x.shape[1], p_drop, layer)
        return torch.from_numpy(keep).to(x.dtype) * torch.tensor(np.float32(1.0 / (1.0 - p_drop))).to(x.dtype)
    return hook


@pytest.mark.parametrize("tag,metric", [("gd2", "ours"), ("gd2", "mse"), ("gd3", "ours"), ("gd3", "cos")])
def test_gcn_dropout_train_class_and_inner_steps(golden_dir, tag, metric):
    """Dropout in the GCN condensation model (models/base.py:59; VERDICT r05 "missing" 3) on the cases of golden G29 (two and three
    GraphConvolution layers, dropout 0.4 — the fixture that pins the restatement's GCN dropout against the reference with torch's recorded
    masks): the engine draws its own counter-based masks — one per class, forward and hidden layer, keyed on the real side by the node id
    of the row — the restatement is handed the same masks, and everything train_class produces must agree: per-class real-side
    gradients, loss, d loss / d A^, d loss / d X~, d loss / d theta_PGE (one synthetic forward per class: per-class activations and
    adjoints, csrc/gs_gcnk.inc gcnk_syn_match_drop), and the parameters after two inner steps (fresh masks per step)."""
    from graphslim_amd.engine import BlockSet
    g = load_golden(golden_dir, "g29_gcn_dropout")
    p_drop, seed, it0 = float(g[f"{tag}_dropout"]), 11, 3
    batches, (n, nclass, d, hidden, t), _ = rebuild_batches(g, tag)
    labels_syn = g[f"{tag}_labels_syn"]
    _, ncd, ranges = O.generate_labels_syn(g[f"{tag}_y"][g[f"{tag}_idx_train"]], float(g[f"{tag}_rate"]))
    ranges = [tuple(ranges[c]) for c in range(nclass)]
    ph = g[f"{tag}_pge2"].shape[0]
    sizes = [int(v) for v in g[f"{tag}_sizes"]]
    L = len(sizes)
    f1, f2, fmid = sizes[0], sizes[-1], tuple(sizes[1:-1])
    eng = make_engine(n, nclass, d, hidden, 2, ph, labels_syn, ranges, g[f"{tag}_x"], f1, f2, inner=2, metric=metric, seed=seed,
                      dropout=p_drop, model="GCN", fmid=fmid)
    eng.set_dropout_iter(it0)
    eng.set_feat(T(g[f"{tag}_feat_syn"]))
    eng.set_pge([T(g[f"{tag}_pge{i}"]) for i in range(10)])
    theta = [T(g[f"{tag}_theta{i}"]) for i in range(2 * L)]
    eng.set_model(theta)
    eng.set_blocks(BlockSet.from_batches(batches, 32, f1, f2, DEV, fmid=fmid))
    O.DROPOUT = engine_masks(seed, it0, p_drop)
    try:
        eng.stage_real_grads()
        gw = eng.buffer("gw_real", (nclass, -1))
        for c in range(nclass):
            th = [q.clone().requires_grad_(True) for q in theta]
            ref, _ = O.class_real_grads("GCN", th, T(g[f"{tag}_x"]), T(g[f"{tag}_y"]), batches[c], L, c)
            close(gw[c], torch.cat([r.reshape(-1) for r in ref]), 2e-4, f"gw_real class {c}")
        th = [q.clone().requires_grad_(True) for q in theta]
        ps = [T(g[f"{tag}_pge{i}"]).requires_grad_(True) for i in range(10)]
        pge = {"lin": ps[:6], "bn": ps[6:]}
        feat_syn = T(g[f"{tag}_feat_syn"]).requires_grad_(True)
        adj_norm = O.dense_gcn_norm(O.pge_forward(feat_syn, pge))
        adj_norm.retain_grad()
        loss_o = O.train_class("GCN", th, batches, T(g[f"{tag}_x"]), T(g[f"{tag}_y"]), feat_syn, adj_norm, T(labels_syn), ncd, nclass, L,
                               metric)
        loss_o.backward()
        eng.stage_pge_forward(False)
        eng.stage_syn_match()
        eng.syn_finish()
        loss = float(eng.buffer("loss")[0])
        assert abs(loss - float(loss_o)) <= 1e-4 * abs(float(loss_o)), (loss, float(loss_o))
        close(eng.buffer("d_adj_norm", (len(labels_syn), -1)), adj_norm.grad.numpy(), 1e-3, "d loss / d A^")
        eng.stage_pge_backward(True, True)
        close(eng.buffer("d_feat", (len(labels_syn), d)), feat_syn.grad.numpy(), 1e-3, "d loss / d X~")
        dp = eng.buffer("d_pge")
        o = 0
        for i, shp in enumerate(eng.pge_shapes()):
            k = int(np.prod(shp))
            ref = ps[i].grad.numpy()
            if np.abs(ref).max() > 1e-6:
                close(dp[o:o + k].reshape(shp), ref, 2e-3, f"d theta_PGE[{i}]")
            o += k
        eng.stage_inner()
        th = [q.clone().requires_grad_(True) for q in theta]
        opt = O.Adam(th, 0.01)
        with torch.no_grad():
            adj_in = O.dense_gcn_norm(O.pge_forward(feat_syn.detach(), pge))
        for it in range(2):
            out = O.model_forward("GCN", feat_syn.detach(), adj_in, th, L, ("inner", it, None))
            opt.step(torch.autograd.grad(torch.nn.functional.nll_loss(out, T(labels_syn)), th))
        for i, q in enumerate(eng.get_model()):
            close(q.cpu(), th[i].detach().numpy(), 2e-3, f"theta[{i}] after two inner steps")
        # another iteration: other masks, another loss
        eng.set_model(theta)
        eng.set_dropout_iter(it0 + 1)
        eng.stage_real_grads()
        eng.stage_syn_match()
        eng.syn_finish()
        assert abs(float(eng.buffer("loss")[0]) - loss) > 1e-5 * abs(loss)
    finally:
        O.DROPOUT = None


@pytest.mark.parametrize("metric", ["ours", "mse"])
def test_dropout_train_class_and_inner_steps(golden_dir, metric):
    """Dropout in the condensation model (models/sgc.py:42; VERDICT r04 "missing" 3) on the case of golden G26 (SGC, two transforms,
    dropout 0.4 — the fixture that pins the restatement's dropout against the reference with torch's recorded masks): the engine draws
    its own counter-based masks, the restatement is handed the same masks, and everything train_class produces must agree — per-class
    real-side gradients (mask on the gathered hidden rows and on the per-edge ReLU indicator), loss, d loss / d A^, d loss / d X~,
    d loss / d theta_PGE (one synthetic forward per class under its own mask) — and the parameters after two inner steps (a fresh mask
    per step).  Also: another iteration counter gives other masks, and dropout 0 / one transform are untouched by the stage."""
    from graphslim_amd.engine import BlockSet, dropout_keep
    g, tag = load_golden(golden_dir, "g26_dropout"), "dr"
    p_drop, seed, it0 = float(g[f"{tag}_dropout"]), 7, 5
    batches, (n, nclass, d, hidden, t), _ = rebuild_batches(g, tag)
    labels_syn = g[f"{tag}_labels_syn"]
    _, ncd, ranges = O.generate_labels_syn(g[f"{tag}_y"][g[f"{tag}_idx_train"]], float(g[f"{tag}_rate"]))
    ranges = [tuple(ranges[c]) for c in range(nclass)]
    ph = g[f"{tag}_pge2"].shape[0]
    f1, f2 = [int(v) for v in g[f"{tag}_sizes"]]
    eng = make_engine(n, nclass, d, hidden, t, ph, labels_syn, ranges, g[f"{tag}_x"], f1, f2, inner=2, metric=metric, seed=seed,
                      dropout=p_drop)
    assert eng.dropout_iter() == 0
    eng.set_dropout_iter(it0)
    eng.set_feat(T(g[f"{tag}_feat_syn"]))
    eng.set_pge([T(g[f"{tag}_pge{i}"]) for i in range(10)])
    theta = [T(g[f"{tag}_theta{i}"]) for i in range(4)]
    eng.set_model(theta)
    eng.set_blocks(BlockSet.from_batches(batches, 32, f1, f2, DEV))
    keep = dropout_keep(seed, it0, "syn", 1, np.arange(len(labels_syn)), hidden, p_drop)
    assert 0.4 < keep.mean() < 0.8 and (keep != dropout_keep(seed, it0 + 1, "syn", 1, np.arange(len(labels_syn)), hidden, p_drop)).any()
    O.DROPOUT = engine_masks(seed, it0, p_drop)
    try:
        eng.stage_real_grads()
        gw = eng.buffer("gw_real", (nclass, -1))
        for c in range(nclass):
            th = [q.clone().requires_grad_(True) for q in theta]
            ref, _ = O.class_real_grads("SGC", th, T(g[f"{tag}_x"]), T(g[f"{tag}_y"]), batches[c], 2, c)
            # (under dropout the per-edge operand is materialised, so the first-layer bias gradient is formed under 'ours' too)
            close(gw[c], torch.cat([r.reshape(-1) for r in ref]), 2e-4, f"gw_real class {c}")
        th = [q.clone().requires_grad_(True) for q in theta]
        ps = [T(g[f"{tag}_pge{i}"]).requires_grad_(True) for i in range(10)]
        pge = {"lin": ps[:6], "bn": ps[6:]}
        feat_syn = T(g[f"{tag}_feat_syn"]).requires_grad_(True)
        adj_norm = O.dense_gcn_norm(O.pge_forward(feat_syn, pge))
        adj_norm.retain_grad()
        loss_o = O.train_class("SGC", th, batches, T(g[f"{tag}_x"]), T(g[f"{tag}_y"]), feat_syn, adj_norm, T(labels_syn), ncd, nclass, 2,
                               metric)
        loss_o.backward()
        eng.stage_pge_forward(False)
        eng.stage_syn_match()
        eng.syn_finish()
        loss = float(eng.buffer("loss")[0])
        assert abs(loss - float(loss_o)) <= 1e-4 * abs(float(loss_o)), (loss, float(loss_o))
        assert abs(loss - float(g[f"{tag}_loss"])) > 1e-3 * abs(loss)      # (torch's masks gave another loss: the masks matter)
        close(eng.buffer("d_adj_norm", (len(labels_syn), -1)), adj_norm.grad.numpy(), 1e-3, "d loss / d A^")
        eng.stage_pge_backward(True, True)
        close(eng.buffer("d_feat", (len(labels_syn), d)), feat_syn.grad.numpy(), 1e-3, "d loss / d X~")
        dp = eng.buffer("d_pge")
        o = 0
        for i, shp in enumerate(eng.pge_shapes()):
            k = int(np.prod(shp))
            ref = ps[i].grad.numpy()
            if np.abs(ref).max() > 1e-6:
                close(dp[o:o + k].reshape(shp), ref, 2e-3, f"d theta_PGE[{i}]")
            o += k
        # PGE.inference + two inner steps, a fresh mask each (gcond.py:63-72; the model is in training mode)
        eng.stage_inner()
        th = [q.clone().requires_grad_(True) for q in theta]
        opt = O.Adam(th, 0.01)
        with torch.no_grad():
            adj_in = O.dense_gcn_norm(O.pge_forward(feat_syn.detach(), pge))
        for it in range(2):
            out = O.model_forward("SGC", feat_syn.detach(), adj_in, th, 2, ("inner", it, None))
            opt.step(torch.autograd.grad(torch.nn.functional.nll_loss(out, T(labels_syn)), th))
        for i, q in enumerate(eng.get_model()):
            close(q.cpu(), th[i].detach().numpy(), 2e-3, f"theta[{i}] after two inner steps")
        # another iteration: other masks, another loss
        eng.set_model(theta)
        eng.set_dropout_iter(it0 + 1)
        eng.stage_real_grads()
        eng.stage_syn_match()
        eng.syn_finish()
        assert abs(float(eng.buffer("loss")[0]) - loss) > 1e-4 * abs(loss)
    finally:
        O.DROPOUT = None
    # one transform: there is no dropout layer (models/sgc.py:38-43) — accepted, no effect, as upstream
    e1 = make_engine(n, nclass, d, hidden, 1, ph, labels_syn, ranges, g[f"{tag}_x"], f1, f2, metric=metric, dropout=p_drop)
    e0 = make_engine(n, nclass, d, hidden, 1, ph, labels_syn, ranges, g[f"{tag}_x"], f1, f2, metric=metric)
    th1 = O.init_sgc(d, hidden, nclass, 1)
    for e in (e0, e1):
        e.set_feat(T(g[f"{tag}_feat_syn"]))
        e.set_pge([T(g[f"{tag}_pge{i}"]) for i in range(10)])
        e.set_model(th1)
        e.set_blocks(BlockSet.from_batches(batches, 32, f1, f2, DEV))
        e.stage_real_grads()
        e.stage_pge_forward(False)
        e.stage_syn_match()
        e.syn_finish()
    assert torch.equal(e0.buffer("d_adj_norm"), e1.buffer("d_adj_norm")) and torch.equal(e0.buffer("gw_real"), e1.buffer("gw_real"))
    # refused, not ignored: BatchNorm, p outside [0, 1)  (the GCN condensation model: test_gcn_dropout_train_class_and_inner_steps)
    with pytest.raises(RuntimeError):
        make_engine(n, nclass, d, hidden, 2, ph, labels_syn, ranges, g[f"{tag}_x"], f1, f2, with_bn=True, dropout=p_drop)
    with pytest.raises(RuntimeError):
        make_engine(n, nclass, d, hidden, 2, ph, labels_syn, ranges, g[f"{tag}_x"], f1, f2, dropout=1.0)


def _problem(t, seed, n=400, nclass=5, d=12, hidden=24, ph=16, rate=0.25, f1=4, f2=3, bcap=16):
    a, x, y = sbm_graph(n, nclass, d, 8, seed)
    rng = np.random.default_rng(seed)
    idx_train = np.sort(rng.permutation(n)[: n // 2])
    labels_train = y[idx_train]
    labels_syn, ncd, ranges = O.generate_labels_syn(labels_train, rate)
    csr = O.normalize_adj_sparse(a)
    pools = [idx_train[labels_train == c] for c in range(nclass)]
    torch.manual_seed(seed)
    theta = O.init_sgc(d, hidden, nclass, t)
    pge = O.init_pge(d, ph)
    pge["bn"] = [torch.rand(ph) + 0.5, torch.randn(ph) * 0.2, torch.rand(ph) + 0.5, torch.randn(ph) * 0.2]
    xs = torch.randn(len(labels_syn), d)
    return dict(a=a, x=x, y=y, csr=csr, pools=pools, labels_syn=labels_syn, ncd=ncd, ranges=[tuple(ranges[c]) for c in range(nclass)],
                theta=theta, pge=pge, xs=xs, n=n, nclass=nclass, d=d, hidden=hidden, ph=ph, f1=f1, f2=f2, bcap=bcap, t=t)


def _batches(p, step):
    out = []
    for c in range(p["nclass"]):
        rng = np.random.default_rng(7000 + 100 * step + c)
        seeds = rng.permutation(p["pools"][c])[: p["bcap"]]
        out.append(O.sample_blocks(*p["csr"], seeds, [p["f1"], p["f2"]], rng))
    return out


@pytest.mark.parametrize("t", [1, 2])
@pytest.mark.parametrize("metric", ["ours", "mse", "cos"])
def test_outer_iterations_follow_the_oracle(t, metric):
    """Three whole outer iterations in a PGE epoch (0) and three in a feature epoch (10), with inner SGC steps:
    X~, theta_PGE, theta and the raw adj_syn trajectory against oracle.GCondState on identical blocks."""
    from graphslim_amd.engine import BlockSet
    p = _problem(t, seed=11 + t)
    eng = make_engine(p["n"], p["nclass"], p["d"], p["hidden"], t, p["ph"], p["labels_syn"], p["ranges"], p["x"], p["f1"],
                      p["f2"], bcap=p["bcap"], metric=metric, inner=2, lrs=(0.01, 0.005, 0.02))
    st = O.GCondState(p["xs"], p["pge"], T(p["labels_syn"]), p["ncd"], p["nclass"], 0.01, 0.005, 0.02, "SGC", 2)
    eng.set_feat(p["xs"])
    eng.set_pge(p["pge"]["lin"] + p["pge"]["bn"])
    X, Y = T(p["x"]), T(p["y"])
    step = 0
    for epoch in (0, 10):
        st.new_epoch(p["theta"])
        eng.set_model(p["theta"])
        for _ in range(3):
            batches = _batches(p, step)
            step += 1
            l_ref, gfeat, gpge, adj_inner = st.outer_iteration(epoch, batches, X, Y, 2, metric)
            eng.set_blocks(BlockSet.from_batches(batches, p["bcap"], p["f1"], p["f2"], DEV))
            eng.outer_step(epoch)
            loss = float(eng.buffer("loss")[0])
            assert abs(loss - float(l_ref)) <= 1e-4 * abs(float(l_ref)), (epoch, loss, float(l_ref))
            if epoch % 50 < 10:
                got = eng.buffer("d_pge")
                ref = torch.cat([g.reshape(-1) for g in gpge])
                big = ref.abs() > 1e-3 * ref.abs().max()
                close(got[big], ref[big], 5e-3, "d theta_PGE")
            else:
                close(eng.buffer("d_feat", gfeat.shape), gfeat, 1e-3, "d X~")
            close(eng.adj_syn.cpu(), adj_inner, 2e-3, "adj_syn after the step")
        # Adam normalises the step to +-lr whatever the gradient's size, so compare on the lr scale
        np.testing.assert_allclose(eng.feat_syn.cpu().numpy(), st.feat_syn.detach().numpy(), atol=2e-3)
        for got, ref in zip(eng.get_model(), st.model):
            np.testing.assert_allclose(got.cpu().numpy(), ref.detach().numpy(), atol=4e-3)
    s, k = eng.read_loss()
    assert k == 6 and s > 0
    assert eng.read_loss() == (0.0, 0)


@pytest.mark.parametrize("metric", ["ours", "cos"])
def test_outer_iterations_with_dropout_follow_the_oracle(metric):
    """Whole outer iterations (the fused two-stream step) with dropout 0.3 in the condensation model (models/sgc.py:42): three in a
    PGE epoch and three in a feature epoch with two inner steps each, against oracle.GCondState handed the engine's masks — the
    iteration counter inside the masks' keys advances with every outer step (gs_gcond_get_dropout_iter), so every iteration, class and
    inner step draws its own mask."""
    from graphslim_amd.engine import BlockSet
    p_drop, seed = 0.3, 9
    p = _problem(2, seed=17)
    eng = make_engine(p["n"], p["nclass"], p["d"], p["hidden"], 2, p["ph"], p["labels_syn"], p["ranges"], p["x"], p["f1"],
                      p["f2"], bcap=p["bcap"], metric=metric, inner=2, lrs=(0.01, 0.005, 0.02), seed=seed, dropout=p_drop)
    st = O.GCondState(p["xs"], p["pge"], T(p["labels_syn"]), p["ncd"], p["nclass"], 0.01, 0.005, 0.02, "SGC", 2)
    eng.set_feat(p["xs"])
    eng.set_pge(p["pge"]["lin"] + p["pge"]["bn"])
    X, Y = T(p["x"]), T(p["y"])
    step = 0
    try:
        for epoch in (0, 10):
            st.new_epoch(p["theta"])
            eng.set_model(p["theta"])
            for _ in range(3):
                batches = _batches(p, step)
                assert eng.dropout_iter() == step
                O.DROPOUT = engine_masks(seed, step, p_drop)
                step += 1
                l_ref, gfeat, gpge, adj_inner = st.outer_iteration(epoch, batches, X, Y, 2, metric)
                eng.set_blocks(BlockSet.from_batches(batches, p["bcap"], p["f1"], p["f2"], DEV))
                eng.outer_step(epoch)
                loss = float(eng.buffer("loss")[0])
                assert abs(loss - float(l_ref)) <= 1e-4 * abs(float(l_ref)), (epoch, loss, float(l_ref))
                if epoch % 50 < 10:
                    got = eng.buffer("d_pge")
                    ref = torch.cat([g.reshape(-1) for g in gpge])
                    big = ref.abs() > 1e-3 * ref.abs().max()
                    close(got[big], ref[big], 5e-3, "d theta_PGE")
                else:
                    close(eng.buffer("d_feat", gfeat.shape), gfeat, 1e-3, "d X~")
                close(eng.adj_syn.cpu(), adj_inner, 2e-3, "adj_syn after the step")
            np.testing.assert_allclose(eng.feat_syn.cpu().numpy(), st.feat_syn.detach().numpy(), atol=2e-3)
            for got, ref in zip(eng.get_model(), st.model):
                np.testing.assert_allclose(got.cpu().numpy(), ref.detach().numpy(), atol=4e-3)
    finally:
        O.DROPOUT = None


@pytest.mark.parametrize("driver,pge", [("c", "rows"), ("py", "rows"), ("c", "replicated")])
def test_sharded_engines_with_dropout_reproduce_the_single_gpu_run(driver, pge):
    """Class shards under dropout: the masks are keyed by the GLOBAL class id, the node id / synthetic row and the shared iteration
    counter, so three ranks' shards draw exactly the single-GPU run's masks and the all-reduced iteration is the unsharded one."""
    from graphslim_amd.parallel import Collective, ShardedGCond
    p = _problem(2, seed=21, nclass=5)
    world = 3

    def build():
        eng = make_engine(p["n"], p["nclass"], p["d"], p["hidden"], 2, p["ph"], p["labels_syn"], p["ranges"], p["x"], p["f1"],
                          p["f2"], bcap=p["bcap"], inner=2, lrs=(0.01, 0.005, 0.02), csr=p["csr"], pools=p["pools"],
                          external=False, seed=5, dropout=0.25)
        eng.set_feat(p["xs"])
        eng.set_pge(p["pge"]["lin"] + p["pge"]["bn"])
        return eng

    ref = build()
    engs = [build() for _ in range(world)]
    shs = [ShardedGCond(e, r, world, [len(q) for q in p["pools"]], Collective(None), pge=pge) for r, e in enumerate(engs)]
    lock = _LockStep(shs, driver)
    for epoch in (0, 10):
        ref.set_model(p["theta"])
        for e in engs:
            e.set_model(p["theta"])
        for _ in range(2):
            ref.outer_step(epoch)
            lock.outer_step(epoch)
            l0 = float(ref.buffer("loss")[0])
            for e in engs:
                assert abs(float(e.buffer("loss")[0]) - l0) <= 2e-5 * abs(l0)
    assert ref.dropout_iter() == 4 and all(e.dropout_iter() == 4 for e in engs)
    for e in engs:
        np.testing.assert_allclose(e.feat_syn.cpu().numpy(), ref.feat_syn.cpu().numpy(), atol=2e-4)
        np.testing.assert_allclose(e.adj_syn.cpu().numpy(), ref.adj_syn.cpu().numpy(), atol=2e-4)
        np.testing.assert_allclose(e.model_params.cpu().numpy(), ref.model_params.cpu().numpy(), atol=5e-4)


def test_engine_sampler_mode_runs_and_is_replayable():
    """Product configuration: blocks come from gs_sample_blocks inside the engine.  Same seed -> identical trajectory;
    and feeding the sampled blocks back through the oracle reproduces the engine's loss."""
    p = _problem(2, seed=5)

    def run(overlap=True, reuse=False):
        eng = make_engine(p["n"], p["nclass"], p["d"], p["hidden"], 2, p["ph"], p["labels_syn"], p["ranges"], p["x"], p["f1"],
                          p["f2"], bcap=p["bcap"], inner=1, csr=p["csr"], pools=p["pools"], external=False, seed=77)
        eng.set_overlap(overlap)
        eng.set_forward_reuse(reuse)
        eng.set_feat(p["xs"])
        eng.set_pge(p["pge"]["lin"] + p["pge"]["bn"])
        eng.set_model(p["theta"])
        losses = []
        for i in range(4):
            eng.outer_step(10)
            losses.append(float(eng.buffer("loss")[0]))
        return eng, losses

    e1, l1 = run()
    e2, l2 = run()
    assert l1 == l2 and torch.equal(e1.feat_syn, e2.feat_syn)          # bitwise reproducible (no float atomics)
    e3, l3 = run(overlap=False)                                        # one stream instead of the fork/join: same bits
    assert l3 == l1 and torch.equal(e3.feat_syn, e1.feat_syn) and torch.equal(e3.model_params, e1.model_params)
    e4, l4 = run(reuse=True)       # training forward of step i+1 adopted from the inference forward of step i: same bits
    assert l4 == l1 and torch.equal(e4.feat_syn, e1.feat_syn) and torch.equal(e4.pge_params, e1.pge_params)
    assert torch.equal(e4.adj_syn, e1.adj_syn) and e4.forward_cached()
    e4.set_feat(p["xs"])
    assert not e4.forward_cached()                                     # host wrote X~: the cached pass is stale
    assert all(np.isfinite(l1)) and l1[0] > 0
    # oracle on the blocks the engine sampled last
    blk = e1.get_blocks()
    batches = blocks_to_batches(blk)
    for c, (bs, n_id, adjs) in enumerate(batches):
        assert set(n_id[:bs].tolist()) <= set(p["pools"][c].tolist())


def blocks_to_batches(blk):
    """Engine ELL layout -> reference-style (batch_size, n_id, [blk0, blk1]) CSR batches."""
    out = []
    f = {k: getattr(blk, k).cpu().numpy() for k in blk.FIELDS}
    for c in range(blk.C):
        B, n1 = int(f["n_seed"][c]), int(f["n1"][c])
        nid1 = f["nid1"][c, :n1]
        n_id = list(nid1)
        pos = {int(v): i for i, v in enumerate(nid1)}
        rp0, c0, v0 = [0], [], []
        for l in range(n1):
            for j in range(blk.f2):
                w = f["val0"][c, l, j]
                if w != 0:
                    u = int(f["col0"][c, l, j])
                    if u not in pos:
                        pos[u] = len(n_id)
                        n_id.append(u)
                    c0.append(pos[u])
                    v0.append(w)
            rp0.append(len(c0))
        rp1, c1, v1 = [0], [], []
        for b in range(B):
            for j in range(blk.f1):
                w = f["val1"][c, b, j]
                if w != 0:
                    c1.append(int(f["col1"][c, b, j]))
                    v1.append(w)
            rp1.append(len(c1))
        blk0 = (np.array(rp0), np.array(c0), np.array(v0, np.float32), (n1, len(n_id)))
        blk1 = (np.array(rp1), np.array(c1), np.array(v1, np.float32), (B, n1))
        out.append((B, np.array(n_id), [blk0, blk1]))
    return out


def test_engine_sampled_blocks_loss_matches_oracle():
    p = _problem(1, seed=9)
    eng = make_engine(p["n"], p["nclass"], p["d"], p["hidden"], 1, p["ph"], p["labels_syn"], p["ranges"], p["x"], p["f1"],
                      p["f2"], bcap=p["bcap"], inner=0, csr=p["csr"], pools=p["pools"], external=False, seed=3)
    eng.set_feat(p["xs"])
    eng.set_pge(p["pge"]["lin"] + p["pge"]["bn"])
    eng.set_model(p["theta"])
    eng.stage_sample()
    eng.stage_real_grads()
    eng.stage_pge_forward(False)
    eng.stage_syn_match()
    eng.syn_finish()
    loss = float(eng.buffer("loss")[0])
    batches = blocks_to_batches(eng.get_blocks())
    theta = [t.clone().requires_grad_(True) for t in p["theta"]]
    adjn = O.dense_gcn_norm(O.pge_forward(p["xs"], p["pge"]))
    ref = O.train_class("SGC", theta, batches, T(p["x"]), T(p["y"]), p["xs"], adjn, T(p["labels_syn"]), p["ncd"], p["nclass"])
    assert abs(loss - float(ref)) <= 1e-4 * abs(float(ref)), (loss, float(ref))


def test_error_paths():
    from graphslim_amd import _lib
    from graphslim_amd.engine import EngineShapes, GCondEngine
    p = _problem(1, seed=2)
    with pytest.raises(SystemExit):
        make_engine(p["n"], p["nclass"], p["d"], p["hidden"], 1, p["ph"], p["labels_syn"], p["ranges"], p["x"], 4, 3, metric="nope")
    with pytest.raises(_lib.GslimError):      # SGC with 3 transforms is outside the built path
        make_engine(p["n"], p["nclass"], p["d"], p["hidden"], 3, p["ph"], p["labels_syn"], p["ranges"], p["x"], 4, 3)
    eng = make_engine(p["n"], p["nclass"], p["d"], p["hidden"], 1, p["ph"], p["labels_syn"], p["ranges"], p["x"], 4, 3)
    with pytest.raises(_lib.GslimError):      # external blocks never supplied
        eng.outer_step(0)


class _LockStep:
    """world sharded engines on ONE GPU.  driver "c": every rank on its own thread calls the library's ONE-call sharded iteration
    (gs_gcond_outer_step_sharded) and the library issues the exchanges through the callback transport (helpers.VirtualWorld sums
    the ranks' buffers); driver "py": the host-side phase list, the all-reduce emulated by summing the buffers between phases."""

    def __init__(self, shs, driver="c"):
        self.shs, self.driver = shs, driver
        if driver == "c":
            from helpers import VirtualWorld
            self.vw = VirtualWorld(len(shs))
            for r, sh in enumerate(shs):
                sh._comm = self.vw.comm(r)

    def outer_step(self, epoch):
        if self.driver == "c":
            before = self.vw.calls
            self.vw.run([(lambda sh=sh: sh.outer_step(epoch)) for sh in self.shs])
            if self.shs[0].pge == "rows":
                assert self.vw.calls - before >= 5      # the library really exchanged (5 - 7 joins per iteration)
            else:
                assert self.vw.calls - before == 1      # replicated PGE: the matching stage's class partials, nothing else
            return
        phases = [sh.phases(epoch) for sh in self.shs]
        for i in range(len(phases[0])):
            for ph in phases:
                ph[i][0]()
            buf = phases[0][i][1]
            if buf is not None:
                views = [sh.v[buf] for sh in self.shs]
                total = torch.stack(views).sum(0)
                for v in views:
                    v.copy_(total)


@pytest.mark.parametrize("pge", ["rows", "replicated"])
@pytest.mark.parametrize("driver", ["c", "py"])
@pytest.mark.parametrize("t,world,fmid", [(1, 2, ()), (2, 2, ()), (2, 3, ()), (2, 8, ()), (2, 3, (3,)), (1, 2, (3, 2))])
def test_sharded_engines_reproduce_the_single_gpu_run(t, world, driver, pge, fmid):
    """class-parallel + PGE pair-row sharding (or the PGE replicated on every rank: ONE exchange per iteration) with
    all-reduce(sum) == the unsharded iteration (same sampled blocks because the sampler keys its RNG by the global class id).
    fmid: middle hops (`--nlayers` 3 / 4): a class shard samples, gathers and back-propagates through its own middle blocks."""
    from graphslim_amd.parallel import Collective, ShardedGCond
    p = _problem(t, seed=21, nclass=5)

    def build():
        eng = make_engine(p["n"], p["nclass"], p["d"], p["hidden"], t, p["ph"], p["labels_syn"], p["ranges"], p["x"], p["f1"],
                          p["f2"], bcap=p["bcap"], inner=2, lrs=(0.01, 0.005, 0.02), csr=p["csr"], pools=p["pools"],
                          external=False, seed=5, fmid=fmid)
        eng.set_feat(p["xs"])
        eng.set_pge(p["pge"]["lin"] + p["pge"]["bn"])
        return eng

    ref = build()
    engs = [build() for _ in range(world)]
    shs = [ShardedGCond(e, r, world, [len(q) for q in p["pools"]], Collective(None), pge=pge) for r, e in enumerate(engs)]
    if pge == "replicated":
        assert all(sh.row_ranges[sh.rank] == (0, len(p["labels_syn"])) for sh in shs)
        assert [b for _, b in shs[0].phases(0) if b] == ["ar_syn"] and [b for _, b in shs[0].phases(10) if b] == ["ar_syn"]
    lock = _LockStep(shs, driver)
    for epoch in (0, 10):
        ref.set_model(p["theta"])
        for e in engs:
            e.set_model(p["theta"])
        for _ in range(2):
            ref.outer_step(epoch)
            lock.outer_step(epoch)
            l0 = float(ref.buffer("loss")[0])
            for e in engs:
                assert abs(float(e.buffer("loss")[0]) - l0) <= 2e-5 * abs(l0)
    for e in engs:
        np.testing.assert_allclose(e.feat_syn.cpu().numpy(), ref.feat_syn.cpu().numpy(), atol=2e-4)
        np.testing.assert_allclose(e.adj_syn.cpu().numpy(), ref.adj_syn.cpu().numpy(), atol=2e-4)
        # layers.{0,1}.bias sit in front of a batch-stat BatchNorm: exactly-zero true gradient, Adam random-walks them
        # on rounding noise (in the reference too) and they cannot influence adj_syn -> excluded
        ph, d = p["ph"], p["d"]
        keep = np.ones(ref.pge_params.numel(), dtype=bool)
        keep[ph * 2 * d: ph * 2 * d + ph] = False
        keep[ph * 2 * d + ph + ph * ph: ph * 2 * d + ph + ph * ph + ph] = False
        np.testing.assert_allclose(e.pge_params.cpu().numpy()[keep], ref.pge_params.cpu().numpy()[keep], atol=5e-4)
        np.testing.assert_allclose(e.model_params.cpu().numpy(), ref.model_params.cpu().numpy(), atol=5e-4)
    # replicas hold identical state after the exchanges (no drift between ranks)
    for e in engs[1:]:
        assert torch.equal(e.feat_syn, engs[0].feat_syn) and torch.equal(e.pge_params, engs[0].pge_params)
        assert torch.equal(e.model_params, engs[0].model_params)


def test_h1_rows_of_a_class_shard_are_the_full_product():
    """A sharded rank computes only the rows of H1 = relu(X W1 + b1) its classes' blocks gather (device-side mark + compact, row-list
    form of the row-panel product): the real-side gradients of the shard must be BITWISE those computed from the full product, in
    every operand mode, and the listed rows of H1full bitwise the full product's rows."""
    import bench
    from graphslim_amd import _lib
    from graphslim_amd.host import init_sgc_params
    p = bench.build_problem("ogbn-arxiv", seed=1, scale=0.25)
    eng = bench.make_engine(p, DEV)
    bench.init_state(p, eng)
    torch.manual_seed(3)
    eng.set_model(init_sgc_params(p["data"].feat_full.shape[1], p["hidden"], p["data"].nclass, p["T"]))
    C, N = p["data"].nclass, p["data"].bench_n
    L = _lib.lib()
    c_lo, c_hi = 7, 10
    assert (c_hi - c_lo) * 256 * 11 * 5 < N          # the row-list form is the one that runs
    eng.set_shard(c_lo, c_hi, 0, len(p["labels_syn"]))
    eng.stage_sample()
    try:
        for mode in ("f32", "bf16x6", "bf16x3"):
            eng.set_operand_mode(mode)
            _lib.check(L.gs_gcond_set_h1_rows(0))
            eng.view("H1full").zero_()
            eng.stage_real_grads()
            full_h1 = eng.view("H1full").clone().reshape(N, -1)
            full_gw = eng.view("gw_real").clone().reshape(C, -1)[c_lo:c_hi]
            _lib.check(L.gs_gcond_set_h1_rows(1))
            eng.view("H1full").fill_(float("nan"))
            eng.stage_real_grads()
            torch.cuda.synchronize()
            h1 = eng.view("H1full").reshape(N, -1)
            rows = ~torch.isnan(h1[:, 0])
            assert 0 < int(rows.sum()) < N
            assert torch.equal(h1[rows], full_h1[rows]), mode
            assert torch.equal(eng.view("gw_real").reshape(C, -1)[c_lo:c_hi], full_gw), mode
            blk = eng.get_blocks()
            need = torch.zeros(N, dtype=torch.bool, device=DEV)
            for c in range(c_lo, c_hi):
                n1 = int(blk.n1[c])
                need[blk.col0[c, :n1].reshape(-1).long()] = True      # (padding slots name row 0 with weight 0: listed too)
            assert torch.equal(need, rows)               # exactly the gathered sources, nothing else
    finally:
        _lib.check(L.gs_gcond_set_h1_rows(1))
        eng.set_operand_mode("f32")


@pytest.mark.parametrize("workload,scale", [("ogbn-arxiv", 0.25), ("cora", 1.0), ("reddit", 0.2)])
def test_fused_chains_against_the_launch_per_product_form(workload, scale):
    """csrc/gs_chain.hip against the first form (gs_gcond_set_fused_chains 0 / 1) from identical state: the matching stage's
    products are the same either way, its distance is one launch instead of three (column sums re-grouped), PGE.inference + the inner SGC steps differ by fp32 re-association (T = 2: products re-associated
    through the C-wide operand, Adam in the gradient products' epilogue; T = 1: logits + softmax + NLL gradient in one launch with
    the K range dealt to 4 or 8 waves)."""
    import bench
    from graphslim_amd import _lib
    from graphslim_amd.host import init_sgc_params
    L = _lib.lib()
    p = bench.build_problem(workload, seed=1, scale=scale)
    d = p["data"].feat_full.shape[1]
    torch.manual_seed(11)
    theta = init_sgc_params(d, p["hidden"], p["data"].nclass, p["T"])

    def run(fused):
        _lib.check(L.gs_gcond_set_fused_chains(fused))
        eng = bench.make_engine(p, DEV)
        bench.init_state(p, eng)
        eng.set_model(theta)
        eng.stage_sample()
        eng.stage_real_grads()
        eng.stage_pge_forward(False)
        eng.stage_syn_match()
        eng.syn_finish()
        out = {"loss": eng.buffer("loss").clone(), "d_feat": eng.buffer("d_feat").clone(), "d_adj": eng.buffer("d_adj").clone(),
               "Ps": eng.buffer("Ps").clone()}
        eng.stage_inner()
        torch.cuda.synchronize()
        out["theta"] = eng.model_params.cpu().clone()
        out["g_model"] = eng.buffer("g_model").clone()
        return out

    try:
        a, b = run(0), run(1)
    finally:
        _lib.check(L.gs_gcond_set_fused_chains(1))
    assert torch.equal(a["Ps"], b["Ps"])
    for k in ("loss", "d_feat", "d_adj"):      # (the one-launch matching distance groups its column sums differently)
        assert float((a[k] - b[k]).abs().max()) <= 1e-5 * float(a[k].abs().max()), k
    gmax = float(a["g_model"].abs().max())
    assert float((a["g_model"] - b["g_model"]).abs().max()) <= 1e-3 * gmax      # the LAST step's gradient (after inner_loop - 1 updates)
    # Adam's first steps move every entry by ~lr: compare on that scale
    assert float((a["theta"] - b["theta"]).abs().max()) <= 0.02 * 0.01 * p["inner"], float((a["theta"] - b["theta"]).abs().max())
    assert float((a["theta"] - b["theta"]).abs().mean()) <= 1e-6


def test_edge_product_split_at_lds_write_is_bitwise_the_read_time_split():
    """Round 3's form of the exact-split edge product (B split once when it is written to LDS as three bf16 planes, fragments by the
    hardware-transposed ds_read_b64_tr_b16) against round 2's (fp32 in LDS, every wave splits what it reads): same h / m / l per
    element, same MFMA order -> the real-side gradients must be BITWISE equal, at the headline shape and on a ragged one."""
    import bench
    from graphslim_amd import _lib
    from graphslim_amd.host import init_sgc_params
    L = _lib.lib()
    for scale in (1.0, 0.13):
        p = bench.build_problem("ogbn-arxiv", seed=1, scale=scale)
        eng = bench.make_engine(p, DEV)
        eng.set_operand_mode("bf16x6")
        bench.init_state(p, eng)
        torch.manual_seed(5)
        eng.set_model(init_sgc_params(p["data"].feat_full.shape[1], p["hidden"], p["data"].nclass, p["T"]))
        eng.stage_sample()
        try:
            _lib.check(L.gs_gemm_set_edge_kernel(2))
            eng.stage_real_grads()
            a = eng.view("gw_real").clone()
            _lib.check(L.gs_gemm_set_edge_kernel(1))
            eng.view("gw_real").zero_()
            eng.stage_real_grads()
            torch.cuda.synchronize()
            b = eng.view("gw_real")
            assert float(a.abs().max()) > 0
            assert torch.equal(a, b), float((a - b).abs().max())
        finally:
            _lib.check(L.gs_gemm_set_edge_kernel(1))


@pytest.mark.parametrize("mode", ["f32", "bf16x6", "bf16x3"])
def test_sign_bit_mask_of_h1_gives_bitwise_the_same_real_side_gradients(mode):
    """The product H1 = relu(X W1 + b1) writes its sign bits in the epilogue (one ballot per accumulator register) and the edge
    product reads the ReLU mask from them (32 B instead of 1 KiB per sampled edge): the bit image must be exactly [H1full > 0] on
    every row the product wrote — the full product and the row-list product of a class shard — and the real-side gradients bitwise
    those of the fp32-mask path."""
    import bench
    from graphslim_amd import _lib
    from graphslim_amd.host import init_sgc_params
    L = _lib.lib()
    p = bench.build_problem("ogbn-arxiv", seed=1, scale=0.25)
    N, C = p["data"].bench_n, p["data"].nclass
    eng = bench.make_engine(p, DEV)
    eng.set_operand_mode(mode)
    bench.init_state(p, eng)
    torch.manual_seed(9)
    eng.set_model(init_sgc_params(p["data"].feat_full.shape[1], p["hidden"], C, p["T"]))
    try:
        for c_lo, c_hi in ((0, C), (7, 10)):
            eng.set_shard(c_lo, c_hi, 0, len(p["labels_syn"]))
            eng.stage_sample()
            _lib.check(L.gs_gcond_set_h1_bits(0))
            eng.stage_real_grads()
            ref = eng.view("gw_real").clone().reshape(C, -1)[c_lo:c_hi]
            _lib.check(L.gs_gcond_set_h1_bits(1))
            eng.view("H1bits", torch.int32).zero_()
            eng.view("H1full").fill_(float("nan"))
            eng.stage_real_grads()
            torch.cuda.synchronize()
            got = eng.view("gw_real").reshape(C, -1)[c_lo:c_hi]
            assert float(ref.abs().max()) > 0 and torch.equal(got, ref), (mode, c_lo, c_hi)
            h1 = eng.view("H1full").reshape(N, -1)
            rows = ~torch.isnan(h1[:, 0])
            assert (int(rows.sum()) == N) == (c_hi - c_lo == C)
            words = eng.view("H1bits", torch.int32).reshape(N, -1)[rows].long() & 0xFFFFFFFF
            bits = ((words[:, :, None] >> torch.arange(32, device=DEV)) & 1).reshape(int(rows.sum()), -1).bool()
            assert torch.equal(bits, h1[rows] > 0), (mode, c_lo, c_hi)
    finally:
        _lib.check(L.gs_gcond_set_h1_bits(0))


def test_two_engines_with_different_knob_settings_do_not_see_each_other():
    """engine-level measurement knobs are per engine (gs_gcond_set_knob); the int-only setters of
    include/gslim_internal.h only move the process default an engine follows while it has no value of its own."""
    import bench
    from graphslim_amd import _lib
    from graphslim_amd.host import init_sgc_params
    L = _lib.lib()
    KNOB_H1_ROWS, KNOB_H1_BITS = 1, 2
    p = bench.build_problem("ogbn-arxiv", seed=1, scale=0.25)
    C = p["data"].nclass
    engs = []
    for _ in range(2):
        eng = bench.make_engine(p, DEV)
        eng.set_operand_mode("bf16x6")
        bench.init_state(p, eng)
        torch.manual_seed(9)
        eng.set_model(init_sgc_params(p["data"].feat_full.shape[1], p["hidden"], C, p["T"]))
        engs.append(eng)
    a, b = engs
    try:
        assert L.gs_gcond_get_knob(a.h, KNOB_H1_BITS) == 0 and L.gs_gcond_get_knob(b.h, KNOB_H1_BITS) == 0      # product default
        _lib.check(L.gs_gcond_set_knob(a.h, KNOB_H1_BITS, 1))
        assert L.gs_gcond_get_knob(a.h, KNOB_H1_BITS) == 1 and L.gs_gcond_get_knob(b.h, KNOB_H1_BITS) == 0
        for eng in engs:
            eng.view("H1bits", torch.int32).zero_()
            eng.stage_sample()
            eng.stage_real_grads()
        torch.cuda.synchronize()
        assert int((a.view("H1bits", torch.int32) != 0).sum()) > 0          # engine a wrote (and read) the sign bits
        assert int((b.view("H1bits", torch.int32) != 0).sum()) == 0         # engine b never did
        assert torch.equal(a.view("gw_real"), b.view("gw_real"))            # same gradients either way (same blocks: same seed)
        # the process default moves only the engine that has no value of its own
        _lib.check(L.gs_gcond_set_knob(b.h, KNOB_H1_ROWS, 0))
        _lib.check(L.gs_gcond_set_h1_rows(1))
        assert L.gs_gcond_get_knob(a.h, KNOB_H1_ROWS) == 1 and L.gs_gcond_get_knob(b.h, KNOB_H1_ROWS) == 0
        _lib.check(L.gs_gcond_set_knob(b.h, KNOB_H1_ROWS, -1))
        assert L.gs_gcond_get_knob(b.h, KNOB_H1_ROWS) == 1
        assert L.gs_gcond_set_knob(a.h, 99, 1) < 0 and L.gs_gcond_get_knob(None, 0) == -1
    finally:
        _lib.check(L.gs_gcond_set_h1_rows(1))


@pytest.mark.parametrize("workload,scale,mode", [("ogbn-arxiv", 1.0, "bf16x6"), ("ogbn-arxiv", 0.13, "bf16x6"), ("cora", 1.0, "bf16x6"),
                                                 ("ogbn-arxiv", 0.25, "f32")])
def test_launches_folded_into_the_pge_passes_give_bitwise_the_same_values(workload, scale, mode):
    """Round 4: the BatchNorm-1 statistics launch also writes the folded layer-1 terms Us | Vs and (extra workgroups) the three-way
    split image of W1; the backward's prep launch also writes the split image of W1^T; the two marginal slab reductions are one
    launch.  Against the launch-per-piece form (knob GS_KNOB_FUSED_CHAINS 0) from identical inputs — the same d_adj is injected —
    every value of the PGE forward and backward is bitwise the same (n' >= 128 with P_h = 256: persistent kernels; n' = 59 and
    P_h = 128: the other routes; the fp32 pipe: nothing is folded)."""
    from graphslim_amd import _lib
    from graphslim_amd import workloads as W
    from graphslim_amd.host import init_sgc_params
    L = _lib.lib()
    KNOB_FUSED_CHAINS = 0
    p = W.build_problem(workload, seed=1, scale=scale)
    d, C = p["data"].feat_full.shape[1], p["data"].nclass
    eng = W.make_engine(p, DEV)
    eng.set_operand_mode(mode)
    W.init_state(p, eng)
    torch.manual_seed(3)
    eng.set_model(init_sgc_params(d, p["hidden"], C, p["T"]))
    n = len(p["labels_syn"])
    g = torch.Generator(device="cpu").manual_seed(5)
    d_adj = (torch.randn(n * n, generator=g) * 1e-3).to(DEV)
    out = {}
    for on in (0, 1):
        _lib.check(L.gs_gcond_set_knob(eng.h, KNOB_FUSED_CHAINS, on))
        for pge_epoch in (False, True):
            for name in ("E2", "e3", "adj_norm", "d_pge", "ar_marg"):
                eng.view(name).fill_(float("nan"))
            eng.stage_pge_forward(False)
            eng.view("d_adj").copy_(d_adj)
            eng.view("d_feat").zero_()
            eng.stage_pge_backward(pge_epoch, True)
            torch.cuda.synchronize()
            out[(on, pge_epoch)] = {k: eng.buffer(k).clone() for k in ("E2", "mu2", "is2", "e3", "adj_norm", "ar_marg", "d_pge", "d_feat")}
    for pge_epoch in (False, True):
        a, b = out[(0, pge_epoch)], out[(1, pge_epoch)]
        for k in a:
            if k == "d_pge" and not pge_epoch:
                continue      # (feature epochs leave the weight gradients they do not need unwritten)
            assert not torch.isnan(b[k]).any() and float(b[k].abs().max()) > 0, (k, pge_epoch)
            assert torch.equal(a[k], b[k]), (k, pge_epoch, float((a[k] - b[k]).abs().max()))


@pytest.mark.parametrize("workload,scale,shard,metric", [("ogbn-arxiv", 1.0, None, "ours"), ("ogbn-arxiv", 0.13, None, "mse"),
                                                         ("ogbn-arxiv", 0.25, (7, 19), "ours"), ("flickr", 1.0, None, "ours"),
                                                         ("arxiv-gate", 0.05, None, "cos")])
def test_chained_class_products_of_the_match_agree_with_the_batched_products(workload, scale, shard, metric):
    """csrc/gs_match.hip: the per-class products of the two-transform match (dHp_c, gW1_c, gb1_c; A1m_c -> Bc_c, dX slabs) as two
    launches that chain products through the MFMA accumulator registers, against the six batched products they replace (knob
    GS_KNOB_CHAINED_PRODUCTS): same fp32 products re-associated -> gw_syn, the matching loss, d loss / d X~ and d loss / d adj to 2e-5
    of each tensor's max.  Shapes: the headline (n' = 454, d = 128, C = 40), ragged n' / class shards / C <= 32 / d = 500, and the
    metrics that also match the bias gradients (mse, cos: gb1_c comes from the first launch)."""
    from graphslim_amd import _lib
    from graphslim_amd import workloads as W
    from graphslim_amd.engine import EngineShapes, GCondEngine
    from graphslim_amd.host import init_sgc_params
    import numpy as np
    L = _lib.lib()
    KNOB_CHAINED_PRODUCTS = 6
    p = W.build_problem(workload, seed=1, scale=scale)
    d, C = p["data"].feat_full.shape[1], p["data"].nclass
    sh = EngineShapes(p["data"].bench_n, d, C, len(p["labels_syn"]), p["T"], p["hidden"], p["ph"], 256, p["fan"][0], p["fan"][1])
    cls_ptr = np.concatenate([[0], np.cumsum([len(q) for q in p["pools"]])])
    eng = GCondEngine(sh, p["data"].bench_feat, p["csr"], cls_ptr, np.concatenate(p["pools"]), p["labels_syn"], p["ranges"],
                      lr_feat=p["lrf"], lr_adj=p["lra"], lr_model=0.01, inner_loop=p["inner"], dis_metric=metric, seed=1, device=DEV)
    W.init_state(p, eng)
    torch.manual_seed(3)
    eng.set_model(init_sgc_params(d, p["hidden"], C, p["T"]))
    if shard:
        eng.set_shard(shard[0], shard[1], 0, len(p["labels_syn"]))
    eng.stage_sample()
    eng.stage_real_grads()
    eng.stage_pge_forward(False)
    out = {}
    for on in (0, 1):
        _lib.check(L.gs_gcond_set_knob(eng.h, KNOB_CHAINED_PRODUCTS, on))
        assert L.gs_gcond_get_knob(eng.h, KNOB_CHAINED_PRODUCTS) == on
        eng.view("gw_syn").zero_()      # ('ours' never writes the bias gradients it does not match)
        for name in ("d_feat", "d_adj", "loss_c"):
            eng.view(name).fill_(float("nan"))
        eng.stage_syn_match()
        torch.cuda.synchronize()
        out[on] = {name: eng.view(name).clone() for name in ("gw_syn", "d_feat", "d_adj", "loss_c")}
    lo, hi = shard if shard else (0, C)
    for name in ("gw_syn", "d_feat", "d_adj", "loss_c"):
        a, b = out[0][name], out[1][name]
        if name == "gw_syn":
            a, b = a.reshape(C, -1)[lo:hi], b.reshape(C, -1)[lo:hi]
        elif name == "loss_c":
            a, b = a[lo:hi], b[lo:hi]
        assert not torch.isnan(b).any() and float(a.abs().max()) > 0, name
        err = float((a - b).abs().max() / a.abs().max())
        assert err < 2e-5, (name, err)


def test_fork_order_knob_changes_nothing_but_the_schedule():
    """GS_KNOB_FORK_LATE (measurement knob: the real-side branch forked after the training forward's long product instead of beside it)
    only moves an event: two engines stepped through PGE and feature epochs with either order hold bitwise the same state."""
    import bench
    from graphslim_amd import _lib
    from graphslim_amd.host import init_sgc_params
    L = _lib.lib()
    KNOB_FORK_LATE = 7
    p = bench.build_problem("ogbn-arxiv", seed=1, scale=0.25)
    d, C = p["data"].feat_full.shape[1], p["data"].nclass
    torch.manual_seed(2)
    theta = init_sgc_params(d, p["hidden"], C, p["T"])
    engs = []
    for late in (0, 1):
        eng = bench.make_engine(p, DEV)
        bench.init_state(p, eng)
        eng.set_model(theta)
        _lib.check(L.gs_gcond_set_knob(eng.h, KNOB_FORK_LATE, late))
        assert L.gs_gcond_get_knob(eng.h, KNOB_FORK_LATE) == late
        for epoch in (0, 0, 10, 10):
            eng.outer_step(epoch)
        torch.cuda.synchronize()
        engs.append(eng)
    a, b = engs
    assert a.read_loss() == b.read_loss()
    assert torch.equal(a.feat_syn, b.feat_syn) and torch.equal(a.pge_params, b.pge_params) and torch.equal(a.model_params, b.model_params)


@pytest.mark.parametrize("workload,scale", [("ogbn-arxiv", 0.25), ("cora", 1.0)])
def test_hoisted_training_forward_changes_nothing_but_the_schedule(workload, scale):
    """Round 5: gs_gcond_outer_step issues the NEXT iteration's training forward of the PGE (gcond.py:48: a function of X~ and theta_PGE
    only, final once the optimiser has stepped) on a third stream beside its inner steps, and the next call waits for it instead of
    computing it (gs_gcond_set_hoist_forward: per engine, off in the library, on in the Python host mirror).  The same function of the same inputs — on the persistent kernel the
    inference pass uses (another fixed partition of the BatchNorm-2 column sums than the un-hoisted training pass: rounding-level
    differences, as with forward reuse): engines stepped through PGE and feature epochs with and without it agree at rounding level
    (close_state); with it, host calls between two steps (a buffer read drops the hoisted pass, a write of X~
    invalidates it) and a changing update rule give BITWISE the state of the undisturbed hoisted run... up to the passes they force
    back to the top of the call, so those are compared at the same tolerances."""
    import bench
    from graphslim_amd import _lib
    from graphslim_amd.host import init_sgc_params
    L = _lib.lib()
    p = bench.build_problem(workload, seed=1, scale=scale)
    d, C = p["data"].feat_full.shape[1], p["data"].nclass
    torch.manual_seed(2)
    theta = init_sgc_params(d, p["hidden"], C, p["T"])
    states = []
    try:
        for hoist, poke in ((0, False), (1, False), (1, True), (1, False)):
            eng = bench.make_engine(p, DEV)
            eng.set_hoist_forward(hoist)
            # (one form of the inner steps in all four runs: by default an engine takes the persistent launch when no hoisted pass runs
            # beside them and the launch chain otherwise — another summation order, which Adam's first steps amplify)
            _lib.check(L.gs_gcond_set_knob(eng.h, _lib.KNOBS["INNER_PERSIST"], 0))
            bench.init_state(p, eng)
            eng.set_model(theta)
            x0 = eng.feat_syn.clone()
            for i, epoch in enumerate((0, 0, 10, 10, 0, 10)):
                eng.outer_step(epoch)
                assert eng.forward_hoisted() == bool(hoist)
                if poke and i == 1:
                    eng.buffer("adj_norm")                    # a host read between two steps: the hoisted pass is waited for and dropped
                if poke and i == 3:
                    eng.set_feat(eng.feat_syn.clone())        # a host write of X~ (same values): invalidated, recomputed
            eng.outer_step_ex(True, True, True, False)        # (DosCond's rule: no inner steps, nothing to hoist beside)
            assert not eng.forward_hoisted()
            eng.outer_step(10)
            torch.cuda.synchronize()
            assert not torch.equal(eng.feat_syn, x0)
            states.append((eng.read_loss(), eng.feat_syn.clone(), eng.pge_params.clone(), eng.model_params.clone(), eng.adj_syn.clone()))
            del eng
    finally:
        pass
    ref = states[0]
    for other in states[1:]:
        assert other[0][1] == ref[0][1] and abs(other[0][0] - ref[0][0]) <= 2e-5 * abs(ref[0][0]), (other[0], ref[0])
        close_state(other[1], ref[1], 2e-4, "X~")
        close_state(other[4], ref[4], 2e-4, "adj_syn")
        close_state(other[3], ref[3], 5e-4, "theta")
    # the schedule itself is deterministic: two undisturbed hoisted runs are bitwise equal
    assert states[1][0] == states[3][0]
    for a, b in zip(states[1][1:], states[3][1:]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("workload,scale,inner", [("ogbn-arxiv", 1.0, 3), ("ogbn-arxiv", 0.25, 3), ("arxiv-gate", 0.05, 3), ("flickr", 0.5, 1),
                                                   ("flickr", 0.25, 5), ("arxiv-gate-c8", 0.05, 16)])
def test_persistent_inner_steps_against_the_launch_chain(workload, scale, inner):
    """VERDICT r05 "next" 1: the inner SGC steps (gcond.py:63-72; two transforms) as ONE persistent launch with grid barriers between the
    four phases of a step (csrc/gs_inner.hip, knob GS_KNOB_INNER_PERSIST) against the launch chain of round 5 (seven dependent launches
    per step) from identical state — the same exact-fp32 arithmetic in another summation order.  The FIRST step is compared at rounding
    level, against the chain AND against an fp64 evaluation of the model's gradient on the host (measured: both forms within 1e-7 ... 5e-7
    of the largest gradient entry of fp64, profiles/r06/diag_inner_persist.txt): gradient, Adam's moments within 2e-6 of their maxima,
    theta within a thousandth of lr where the gradient is not at rounding level (Adam's first step moves an entry by lr * sign(g): an
    entry whose gradient IS rounding noise may land lr apart).  Later steps inherit those sign flips, so `inner` steps and a second
    call are compared at the tolerances of test_fused_chains_against_the_launch_per_product_form.  Shapes: the headline shape
    (n' = 454, 29 row blocks), n' = 113 / 136 / 111 / 55 (ragged last block), d = 500 (not a multiple of 16: flickr), C = 7 / 8 (one
    class tile) and 40 (three), 1 ... 16 steps.  Repeat launches are bitwise equal; a second call continues from the first one's state
    (the arrival counter runs on)."""
    import bench
    from graphslim_amd import _lib
    from graphslim_amd.host import init_sgc_params
    L = _lib.lib()
    p = bench.build_problem(workload, seed=1, scale=scale)
    d, C = p["data"].feat_full.shape[1], p["data"].nclass
    torch.manual_seed(5)
    theta = init_sgc_params(d, p["hidden"], C, p["T"])
    KNOB = _lib.KNOBS["INNER_PERSIST"]
    lr = 0.01

    def run(on, steps, calls):
        p["inner"] = steps
        eng = bench.make_engine(p, DEV)
        _lib.check(L.gs_gcond_set_knob(eng.h, KNOB, on))
        bench.init_state(p, eng)
        eng.set_model(theta)
        eng.stage_pge_forward(True)
        outs = []
        for _ in range(calls):
            eng.inner_steps()
            torch.cuda.synchronize()
            outs.append((eng.model_params.cpu().clone(), eng.buffer("m_model").clone(), eng.buffer("v_model").clone(), eng.buffer("g_model").clone()))
        eng.read_loss()      # (raises if a grid barrier of the persistent launch timed out)
        ctx = (eng.buffer("adj_norm_inner", (eng.shapes.n_syn, eng.shapes.n_syn)).double(), eng.feat_syn.cpu().double())
        return outs, ctx

    # ---- one step: rounding level, against the chain and against fp64 ----
    (c1,), (A, X) = run(0, 1, 1)
    (p1,), _ = run(2, 1, 1)
    W1, b1, Wl, bl = [t.double().clone().requires_grad_(True) for t in theta]
    Q = A @ A
    out = Q @ (torch.relu(X @ W1 + b1) @ Wl) + Q.sum(1, keepdim=True) * bl
    torch.nn.functional.nll_loss(torch.log_softmax(out, 1), torch.from_numpy(p["labels_syn"]).long()).backward()
    g64 = torch.cat([t.grad.reshape(-1) for t in (W1, b1, Wl, bl)])
    gmax = float(g64.abs().max())
    assert float((p1[3].double() - g64).abs().max()) <= 2e-6 * gmax
    assert float((c1[3].double() - g64).abs().max()) <= 2e-6 * gmax
    for k, tol in ((1, 2e-6), (2, 4e-6)):
        assert float((p1[k] - c1[k]).abs().max()) <= tol * float(c1[k].abs().max()), k
    big = g64.abs() > 1e-3 * gmax
    assert float((p1[0] - c1[0])[big].abs().max()) <= 1e-3 * lr
    assert not torch.equal(c1[0], theta_flat(theta))
    # ---- the step loop inside the launch: ONE launch of `inner` steps == `inner` launches of one step, bitwise (same arithmetic per
    # step: the transposed operands are re-derived from theta at the top of every launch), twice in a row (the arrival counter runs
    # on); against the chain only a sanity bound that grows with the steps taken (an error in the loop / the barriers would be O(1)) ----
    one, _ = run(2, inner, 2)
    many, _ = run(2, 1, 2 * inner)
    for call in range(2):
        for x, y in zip(one[call], many[inner * (call + 1) - 1]):
            assert torch.equal(x, y), call
    chain, _ = run(0, inner, 2)
    for call in range(2):
        assert float((one[call][0] - chain[call][0]).abs().max()) <= 0.5 * lr * inner * (call + 1), call


def theta_flat(params):
    return torch.cat([t.reshape(-1).float() for t in params])


def test_a_c_caller_may_write_feat_syn_between_default_configured_steps():
    """VERDICT r05 weak 4 / "next" 5: the hoisted forward is opt-in PER ENGINE and the library's default is OFF, so a C caller that
    writes io.feat_syn / io.pge_params between two gs_gcond_outer_step calls — correct through round 4 — is still correct: nothing
    of the engine reads those buffers after the call has returned.  Driven the way a C caller would: the raw library calls, the
    buffers written with a plain device copy on the same stream, NO gs_gcond_invalidate_forward.  The second step must compute with the
    values written (bitwise the state of an engine that was handed those values through the bracketed host path), and an engine that
    opted in must report the pass pending (the contract the header spells out)."""
    import bench
    from graphslim_amd import _lib
    from graphslim_amd.host import init_sgc_params
    L = _lib.lib()
    p = bench.build_problem("ogbn-arxiv", seed=1, scale=0.25)
    d, C = p["data"].feat_full.shape[1], p["data"].nclass
    torch.manual_seed(3)
    theta = init_sgc_params(d, p["hidden"], C, p["T"])
    out = []
    for raw in (True, False):
        eng = bench.make_engine(p, DEV)
        eng.set_hoist_forward(0)                       # what gs_gcond_create leaves a C caller with
        assert L.gs_gcond_forward_hoisted(eng.h) == 0
        bench.init_state(p, eng)
        eng.set_model(theta)
        _lib.check(L.gs_gcond_outer_step(eng.h, 10, _lib.stream()))
        assert L.gs_gcond_forward_hoisted(eng.h) == 0  # nothing of the engine is reading X~ now
        new_x = (eng.feat_syn * 1.25 + 0.5).clone()
        new_pge = (eng.pge_params * 0.75).clone()
        if raw:
            eng.feat_syn.copy_(new_x)                  # the C caller's write: same stream, no library call around it
            eng.pge_params.copy_(new_pge)
        else:
            eng.set_feat(new_x)
            eng.set_pge([new_pge])
        _lib.check(L.gs_gcond_outer_step(eng.h, 10, _lib.stream()))
        _lib.check(L.gs_gcond_outer_step(eng.h, 0, _lib.stream()))
        torch.cuda.synchronize()
        out.append((eng.read_loss(), eng.feat_syn.clone(), eng.pge_params.clone(), eng.adj_syn.clone()))
        del eng
    assert out[0][0] == out[1][0]
    for a, b in zip(out[0][1:], out[1][1:]):
        assert torch.equal(a, b)
    # opted in, the same call leaves a pass pending: the caller must bracket its writes (include/gslim.h, gs_gcond_set_hoist_forward)
    eng = bench.make_engine(p, DEV)
    eng.set_hoist_forward(1)
    bench.init_state(p, eng)
    eng.set_model(theta)
    _lib.check(L.gs_gcond_outer_step(eng.h, 10, _lib.stream()))
    assert L.gs_gcond_forward_hoisted(eng.h) == 1
    _lib.check(L.gs_gcond_invalidate_forward(eng.h))
    assert L.gs_gcond_forward_hoisted(eng.h) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("mode", ["f32", "bf16x6"])
def test_per_edge_sign_bits_of_the_hop2_gather_give_bitwise_the_same_real_side_gradients(mode):
    """VERDICT r03 "next" 2a: the hop-2 gather kernel has every gathered row of H1 = relu(X W1 + b1) in registers and writes its
    sign bits per edge slot (32 B); the edge product reads its ReLU mask from those instead of gathering the 1-KiB fp32 row of H1full
    again per edge.  The real-side gradients must be bitwise those of the fp32-mask path, for the full class range and a class shard,
    and the words must be exactly [H1full[src(e)] > 0] for every live edge slot."""
    import bench
    from graphslim_amd import _lib
    from graphslim_amd.host import init_sgc_params
    L = _lib.lib()
    p = bench.build_problem("ogbn-arxiv", seed=1, scale=0.25)
    C = p["data"].nclass
    eng = bench.make_engine(p, DEV)
    eng.set_operand_mode(mode)
    bench.init_state(p, eng)
    torch.manual_seed(9)
    eng.set_model(init_sgc_params(p["data"].feat_full.shape[1], p["hidden"], C, p["T"]))
    try:
        for c_lo, c_hi in ((0, C), (7, 10)):
            eng.set_shard(c_lo, c_hi, 0, len(p["labels_syn"]))
            eng.stage_sample()
            _lib.check(L.gs_gcond_set_edge_bits(0))
            eng.stage_real_grads()
            ref = eng.view("gw_real").clone().reshape(C, -1)[c_lo:c_hi]
            _lib.check(L.gs_gcond_set_edge_bits(1))
            eng.view("edge_bits", torch.int32).fill_(-1)
            eng.stage_real_grads()
            torch.cuda.synchronize()
            got = eng.view("gw_real").reshape(C, -1)[c_lo:c_hi]
            assert float(ref.abs().max()) > 0 and torch.equal(got, ref), (mode, c_lo, c_hi)
            # the words against the mask rows, for the live slots of the first class of the range
            blk = eng.get_blocks()
            f2 = blk.col0.shape[2]
            n1 = int(blk.n1[c_lo])
            src = blk.col0[c_lo, :n1].reshape(-1).long()
            h1 = eng.view("H1full").reshape(p["data"].bench_n, -1)
            words = eng.view("edge_bits", torch.int32).reshape(C, -1, 8)[c_lo, :n1 * f2].long() & 0xFFFFFFFF
            bits = ((words[:, :, None] >> torch.arange(32, device=DEV)) & 1).reshape(n1 * f2, -1).bool()
            assert torch.equal(bits, h1[src] > 0), (mode, c_lo, c_hi)
    finally:
        _lib.check(L.gs_gcond_set_edge_bits(0))


@pytest.mark.parametrize("mode", ["f32", "bf16x6"])
def test_balanced_edge_slabs_give_the_fixed_slab_gradients(mode):
    """Round 5: the real-side edge product (gcond_base.py:223 through relu and the hop-2 block; csrc/gs_edge.hip) on BALANCED K slabs —
    a device-side plan cuts the chip's 512 workgroup slots over the classes by their live edge counts (knob GS_KNOB_EDGE_PLAN) —
    against the fixed-length slabs of rounds 2-4: the same fp32 sums in another fixed partition, so gw_real agrees to 2e-6 of each
    class's max, for the full class range and for class shards (few classes: long lists in many slabs; one class), the plan uses every
    slot it may, and two runs of the same plan are bitwise equal."""
    import bench
    from graphslim_amd import _lib
    from graphslim_amd.host import init_sgc_params
    L = _lib.lib()
    p = bench.build_problem("ogbn-arxiv", seed=1, scale=0.25)
    C = p["data"].nclass
    eng = bench.make_engine(p, DEV)
    eng.set_operand_mode(mode)
    bench.init_state(p, eng)
    torch.manual_seed(9)
    eng.set_model(init_sgc_params(p["data"].feat_full.shape[1], p["hidden"], C, p["T"]))
    KNOB = _lib.KNOBS["EDGE_PLAN"]
    assert L.gs_gcond_get_knob(eng.h, KNOB) == 1      # the product default
    try:
        for c_lo, c_hi in ((0, C), (7, 10), (3, 4)):
            eng.set_shard(c_lo, c_hi, 0, len(p["labels_syn"]))
            eng.stage_sample()
            _lib.check(L.gs_gcond_set_knob(eng.h, KNOB, 0))
            eng.stage_real_grads()
            ref = eng.view("gw_real").clone().reshape(C, -1)[c_lo:c_hi]
            _lib.check(L.gs_gcond_set_knob(eng.h, KNOB, 1))
            eng.stage_real_grads()
            got = eng.view("gw_real").clone().reshape(C, -1)[c_lo:c_hi]
            eng.stage_real_grads()
            torch.cuda.synchronize()
            again = eng.view("gw_real").reshape(C, -1)[c_lo:c_hi]
            assert float(ref.abs().max()) > 0
            err = float(((got - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)).max())
            assert err <= 2e-6, (mode, c_lo, c_hi, err)
            assert torch.equal(got, again), (mode, c_lo, c_hi)
            target = 512      # (a staged call: the stage has the chip to itself; beside the training forward the engine aims at 7/8)
            plan = eng.view("edge_plan", torch.int32)[:2 * (c_hi - c_lo) + 1].cpu().numpy()
            Cl = c_hi - c_lo
            base, chunk = plan[:Cl + 1], plan[Cl + 1:]
            n1 = eng.get_blocks().n1[c_lo:c_hi].cpu().numpy()
            f2 = eng.get_blocks().col0.shape[2]
            steps = -(-(n1 * f2) // 16)
            slabs = np.diff(base)
            assert base[0] == 0 and (slabs >= 1).all() and (chunk % 16 == 0).all()
            assert ((slabs - 1) * chunk < n1 * f2).all() and (slabs * chunk >= n1 * f2).all()      # a partition: no empty slab, nothing left out
            assert 0.7 * min(target, steps.sum()) <= base[-1] <= target + Cl, (base[-1], steps.sum())      # (d = 128: one feature tile; a slab is >= one step)
            per = chunk // 16
            assert per.max() <= 1.25 * steps.sum() / base[-1] + 1, (per, steps.sum())          # every workgroup runs ~ total / slabs steps (rounded up per class)
    finally:
        _lib.check(L.gs_gcond_set_knob(eng.h, KNOB, -1))


def _check_dedup_against_numpy(eng, c_lo, c_hi):
    """The distinct-source structure of the engine's current blocks (dd_* buffers) against numpy: distinct sources by first live slot,
    every live slot once in its source's list, lists ascending, entries = the lists cut in pairs.  Returns slots - entries."""
    blk = eng.get_blocks()
    col0, val0, n1 = (getattr(blk, k).cpu().numpy() for k in ("col0", "val0", "n1"))
    dd = {k: eng.view(k, torch.int32).cpu().numpy() for k in ("dd_n0", "dd_src0", "dd_sptr0", "dd_sedge0")}
    cap0 = col0.shape[1] * col0.shape[2]
    fewer = 0
    for c in range(c_lo, c_hi):
        live = np.flatnonzero(val0[c, :n1[c]].reshape(-1) != 0)
        srcs = col0[c, :n1[c]].reshape(-1)[live]
        uniq, first = np.unique(srcs, return_index=True)
        order = uniq[np.argsort(first)]                                   # distinct sources by first live slot
        n0 = int(dd["dd_n0"][c])
        assert n0 == len(order) and np.array_equal(dd["dd_src0"][c * cap0:c * cap0 + n0], order), c
        sptr = dd["dd_sptr0"][c * (cap0 + 1):c * (cap0 + 1) + n0 + 1]
        sedge = dd["dd_sedge0"][c * cap0:c * cap0 + sptr[-1]]
        assert sptr[0] == 0 and sptr[-1] == len(live) and (np.diff(sptr) >= 1).all()
        assert np.array_equal(np.sort(sedge), live)                       # every live slot exactly once
        owner = np.repeat(np.arange(n0), np.diff(sptr))
        assert np.array_equal(col0[c].reshape(-1)[sedge], order[owner])   # ... in its own source's list
        asc = np.ones(len(sedge), bool)
        asc[1:] = (np.diff(sedge) > 0) | (np.diff(owner) > 0)
        assert asc.all()                                                  # ... lists ascending by slot
        # the K entries: every source's list cut in pairs, (row of Y, weight) per edge, 0.f weight behind an odd tail
        nent = int(eng.view("dd_nent", torch.int32)[c])
        deg = np.diff(sptr)
        assert nent == int(((deg + 1) // 2).sum())
        esrc = eng.view("dd_ent_src", torch.int32)[c * cap0:c * cap0 + nent].cpu().numpy()
        erow = eng.view("dd_ent_row", torch.int32)[2 * c * cap0:2 * (c * cap0 + nent)].cpu().numpy().reshape(-1, 2)
        ew = eng.view("dd_ent_w")[2 * c * cap0:2 * (c * cap0 + nent)].cpu().numpy().reshape(-1, 2)
        f2_ = col0.shape[2]
        vflat = val0[c].reshape(-1)
        starts = np.concatenate([np.arange(a, b, 2) for a, b in zip(sptr[:-1], sptr[1:])]) if n0 else np.zeros(0, int)
        has2 = (starts + 1) < np.repeat(sptr[1:], (deg + 1) // 2)
        s0 = sedge[starts]
        s1 = np.where(has2, sedge[np.minimum(starts + 1, len(sedge) - 1)], s0)
        assert np.array_equal(esrc, np.repeat(order, (deg + 1) // 2))
        assert np.array_equal(erow, np.stack([s0 // f2_, s1 // f2_], 1))
        assert np.array_equal(ew, np.stack([vflat[s0], np.where(has2, vflat[s1], 0.0)], 1).astype(np.float32))
        fewer += len(live) - nent
    return fewer


@pytest.mark.parametrize("workload,scale,mode", [("ogbn-arxiv", 0.25, "f32"), ("ogbn-arxiv", 0.25, "bf16x6"), ("ogbn-arxiv", 1.0, "bf16x6"),
                                                 ("arxiv-gate-c8", 0.05, "bf16x6")])
def test_edge_product_over_distinct_sources(workload, scale, mode):
    """VERDICT r05 "next" 2: the real-side edge product gW1_c = sum_e x_src(e) (x) (w_e [H1[src(e)] > 0] Y[row e]) (gcond_base.py:212-224
    through relu and the outermost block) depends on the source only through x and the mask, so the engine sums a source's in-edges
    in the operand path: behind every sampling gs_dedup_last_kernel builds the distinct sources, the CSC of the outermost block over
    them and the product's K ENTRIES = (source, up to two of its in-edges as (row of Y, weight)) (knob GS_KNOB_EDGE_SRC; measured, not the default: profiles/r06/edge_src_ab.txt).
    Checked: (1) the de-duplication against numpy — distinct sources in the order of their first live edge slot (PyG's relabelling
    order, dataset/loader.py:216-223), every live slot exactly once in its source's list, lists ascending, the entries = the lists cut
    in pairs; (2) gw_real against the slot form of rounds 2-5 from the same blocks: the same sum in another association, 5e-6 of each
    class's max (the slot form itself moves by 2e-6 between two slab partitions), full class range and class shards; (3) bitwise
    repeatability; (4) the same through gs_gcond_set_blocks (blocks supplied by the caller are de-duplicated on upload)."""
    import bench
    from graphslim_amd import _lib
    from graphslim_amd.engine import BlockSet
    from graphslim_amd.host import init_sgc_params
    L = _lib.lib()
    p = bench.build_problem(workload, seed=1, scale=scale)
    C = p["data"].nclass
    eng = bench.make_engine(p, DEV)
    eng.set_operand_mode(mode)
    bench.init_state(p, eng)
    torch.manual_seed(9)
    theta = init_sgc_params(p["data"].feat_full.shape[1], p["hidden"], C, p["T"])
    eng.set_model(theta)
    KNOB = _lib.KNOBS["EDGE_SRC"]
    assert L.gs_gcond_get_knob(eng.h, KNOB) == 0      # measured, not the default (profiles/r06/edge_src_ab.txt)
    shards = ((0, C), (1, 4), (C - 1, C)) if C > 8 else ((0, C), (2, 3))
    for c_lo, c_hi in shards:
        eng.set_shard(c_lo, c_hi, 0, len(p["labels_syn"]))
        # (the last shard with the one-workgroup-per-class form of the de-duplication: the numpy check below pins the structure
        # bit for bit, so both forms passing it are equal)
        _lib.check(L.gs_dedup_set_form(0 if (c_lo, c_hi) == shards[-1] else 1))
        eng.stage_sample()
        _lib.check(L.gs_gcond_set_knob(eng.h, KNOB, 0))
        eng.stage_real_grads()
        ref = eng.view("gw_real").clone().reshape(C, -1)[c_lo:c_hi]
        _lib.check(L.gs_gcond_set_knob(eng.h, KNOB, 1))
        eng.stage_real_grads()
        got = eng.view("gw_real").clone().reshape(C, -1)[c_lo:c_hi]
        eng.stage_real_grads()
        torch.cuda.synchronize()
        again = eng.view("gw_real").reshape(C, -1)[c_lo:c_hi]
        assert float(ref.abs().max()) > 0
        err = float(((got - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)).max())
        assert err <= 5e-6, (mode, c_lo, c_hi, err)
        assert torch.equal(got, again), (mode, c_lo, c_hi)
        fewer = _check_dedup_against_numpy(eng, c_lo, c_hi)
        assert fewer > 0      # (the form has something to save)
    _lib.check(L.gs_dedup_set_form(1))
    # blocks supplied by the caller are de-duplicated on upload
    eng.set_shard(0, C, 0, len(p["labels_syn"]))
    eng.stage_sample()
    eng.stage_real_grads()
    want = eng.view("gw_real").clone()
    blocks = eng.get_blocks()
    sh = eng.shapes
    from graphslim_amd.engine import EngineShapes, GCondEngine
    ext = GCondEngine(sh, p["data"].bench_feat, None, None, None, p["labels_syn"], p["ranges"], lr_feat=p["lrf"], lr_adj=p["lra"],
                      lr_model=0.01, inner_loop=p["inner"], external_blocks=True, device=DEV)
    ext.set_operand_mode(mode)
    _lib.check(L.gs_gcond_set_knob(ext.h, KNOB, 1))
    ext.set_model(theta)
    ext.set_blocks(blocks)
    ext.stage_real_grads()
    torch.cuda.synchronize()
    assert torch.equal(ext.view("gw_real"), want)


def test_dedup_long_slot_lists():
    """The de-duplication's long lists (csrc/gs_sampler.hip gs_ddm_long_kernel): a source sampled by many frontier rows.  Blocks of the
    flickr shape (fan-out [15, 8]: 32 768 edge slots per class) are rewritten so that class 0 hangs EVERY live slot on one node (a list
    beyond the 16 384 slots the workgroup sort holds in LDS: the one-thread heap sort), class 1 on two nodes alternately (two bitonic
    sorts of thousands of slots), class 2 on 40 nodes (lists of hundreds) — uploaded through gs_gcond_set_blocks and checked against
    numpy like the sampled blocks of test_edge_product_over_distinct_sources (which also compares the two forms of the de-duplication
    bitwise; the one-workgroup form's one-thread insertion sort would take seconds on these lists)."""
    import bench
    from graphslim_amd import _lib
    from graphslim_amd.engine import GCondEngine
    L = _lib.lib()
    p = bench.build_problem("flickr", seed=1, scale=0.3)
    C = p["data"].nclass
    eng = bench.make_engine(p, DEV)
    bench.init_state(p, eng)
    eng.stage_sample()
    blocks = eng.get_blocks()
    torch.cuda.synchronize()
    f2, cap1 = blocks.col0.shape[2], blocks.col0.shape[1]
    # (the de-duplication reads col0 / val0 / n1 only — nothing else of these blocks is used here: every row of classes 0 and 1 is made live)
    blocks.n1[0], blocks.n1[1] = cap1, (3 * cap1) // 4
    n1 = blocks.n1.cpu().numpy()
    assert int(n1[0]) * f2 > 16384 and int(n1[1]) * f2 // 2 > 4096, (n1, f2)      # class 0's single list exceeds the LDS sort
    for c, nodes in ((0, [7]), (1, [3, 11]), (2, list(range(100, 140)))):
        k = int(n1[c]) * f2
        pick = torch.tensor(nodes, dtype=torch.int32, device=DEV)[torch.arange(k, device=DEV) % len(nodes)]
        blocks.col0[c].view(-1)[:k] = pick
        blocks.val0[c].view(-1)[:k] = torch.where(blocks.val0[c].view(-1)[:k] != 0, blocks.val0[c].view(-1)[:k], torch.full((k,), 0.25, device=DEV))
    KNOB = _lib.KNOBS["EDGE_SRC"]
    ext = GCondEngine(eng.shapes, p["data"].bench_feat, None, None, None, p["labels_syn"], p["ranges"], lr_feat=p["lrf"], lr_adj=p["lra"],
                      lr_model=0.01, inner_loop=p["inner"], external_blocks=True, device=DEV)
    _lib.check(L.gs_gcond_set_knob(ext.h, KNOB, 1))
    ext.set_blocks(blocks)
    torch.cuda.synchronize()
    assert _check_dedup_against_numpy(ext, 0, C) > 0
    n0 = ext.view("dd_n0", torch.int32).cpu().numpy()
    assert int(n0[0]) == 1 and int(n0[1]) == 2 and int(n0[2]) == 40
    # the marker rows are FREE again: the same upload a second time gives the same structure
    before = {k: ext.view(k, torch.int32).clone() for k in ("dd_n0", "dd_nent", "dd_ent_src")}
    ext.set_blocks(blocks)
    torch.cuda.synchronize()
    assert _check_dedup_against_numpy(ext, 0, C) > 0
    assert torch.equal(before["dd_n0"], ext.view("dd_n0", torch.int32)) and torch.equal(before["dd_nent"], ext.view("dd_nent", torch.int32))


def test_every_knob_reports_its_process_default():
    """gs_gcond_get_knob's table of process defaults is hand-ordered; a fresh engine must report the documented default for
    every knob id of include/gslim_internal.h (graphslim_amd._lib.KNOBS, pinned to the enum by tests/test_abi.py)."""
    import bench
    from graphslim_amd import _lib
    L = _lib.lib()
    p = bench.build_problem("ogbn-arxiv", seed=1, scale=0.05)
    eng = bench.make_engine(p, DEV)
    want = {"FUSED_CHAINS": 1, "H1_ROWS": 1, "H1_BITS": 0, "EDGE_BITS": 0, "DW1_GENERATED": 1, "FOLDED_REDUCTIONS": 1, "CHAINED_PRODUCTS": 1,
            "FORK_LATE": 0, "EDGE_PLAN": 1, "INNER_PERSIST": 1, "EDGE_SRC": 0}
    assert set(want) == set(_lib.KNOBS)
    for name, k in _lib.KNOBS.items():
        assert L.gs_gcond_get_knob(eng.h, k) == want[name], name
        _lib.check(L.gs_gcond_set_knob(eng.h, k, 1 - want[name]))
        assert L.gs_gcond_get_knob(eng.h, k) == 1 - want[name], name
        _lib.check(L.gs_gcond_set_knob(eng.h, k, -1))
        assert L.gs_gcond_get_knob(eng.h, k) == want[name], name
    assert L.gs_gcond_get_knob(eng.h, len(_lib.KNOBS)) == -1


@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("bf16x6", 2e-5), ("bf16x3", 1e-3), ("bf16", 3e-2)])
@pytest.mark.parametrize("scale", [1.0, 0.13])
def test_pge_epoch_backward_on_generated_operands(mode, tol, scale):
    """PGE epochs (gcond.py:58-59; parametrized_adj.py:57-64 backward): dW1 = dE2^T A1 and db1 = colsum(dE2) from operands generated
    in the kernel (dE2 from E2, the layer-1 activations from U, V with the BatchNorm-1 affine folded in; gs_dw1_gen_kernel) against
    round 2's form, which writes both operands to memory and multiplies them on the fp32 matrix pipe: every PGE parameter gradient
    within the operand mode's tolerance of that reference (relative to the largest entry of its tensor), at the headline shape
    (n' = 454: 29 K steps per pair row, the last one masked) and on a small ragged one."""
    import bench
    from graphslim_amd import _lib
    from graphslim_amd.host import init_sgc_params
    L = _lib.lib()
    p = bench.build_problem("ogbn-arxiv", seed=1, scale=scale)
    eng = bench.make_engine(p, DEV)
    eng.set_operand_mode(mode)
    bench.init_state(p, eng)
    torch.manual_seed(4)
    eng.set_model(init_sgc_params(p["data"].feat_full.shape[1], p["hidden"], p["data"].nclass, p["T"]))
    eng.stage_sample()
    eng.stage_real_grads()
    eng.stage_pge_forward(False)
    eng.stage_syn_match()
    eng.syn_finish()
    shapes = eng.pge_shapes()
    out = {}
    try:
        for gen in (0, 1):
            _lib.check(L.gs_gcond_set_dw1_generated(gen))
            eng.view("d_pge").fill_(float("nan"))
            eng.stage_pge_backward(True, False)
            torch.cuda.synchronize()
            out[gen] = eng.view("d_pge").clone()
    finally:
        _lib.check(L.gs_gcond_set_dw1_generated(1))
    assert not torch.isnan(out[1]).any()
    floor = 1e-6 * float(out[0].abs().max())
    off = 0
    for shp in shapes:
        k = int(np.prod(shp))
        a, b = out[0][off:off + k], out[1][off:off + k]
        off += k
        # (b0 and b1 feed a BatchNorm: their gradients are analytically zero, both forms leave rounding noise ~1e-9 there)
        assert float((a - b).abs().max()) <= tol * float(a.abs().max()) + floor, (shp, float((a - b).abs().max()), float(a.abs().max()))
    assert off == out[0].numel()


def test_library_rccl_binding_world1():
    """gs_comm's RCCL transport (dlopen of librccl, ncclCommInitRank, ncclAllReduce on the engine's stream) with ONE rank — all a
    1-GPU box can run of it: a whole sharded outer iteration through gs_gcond_outer_step_sharded reproduces gs_gcond_outer_step
    bitwise (world 1: every all-reduce is the identity, but each one goes through ncclAllReduce on fp32 and fp64 engine buffers),
    and the bookkeeping reports the joins."""
    from graphslim_amd.parallel import Comm, ShardedGCond
    p = _problem(2, seed=3)

    def build():
        eng = make_engine(p["n"], p["nclass"], p["d"], p["hidden"], 2, p["ph"], p["labels_syn"], p["ranges"], p["x"], p["f1"],
                          p["f2"], bcap=p["bcap"], inner=1, csr=p["csr"], pools=p["pools"], external=False, seed=5)
        eng.set_feat(p["xs"])
        eng.set_pge(p["pge"]["lin"] + p["pge"]["bn"])
        eng.set_model(p["theta"])
        return eng

    ref, eng = build(), build()
    comm = Comm.rccl_world1()
    assert comm.world == 1
    v = torch.arange(1000, device=DEV, dtype=torch.float32)
    comm.allreduce(v)
    w = torch.arange(77, device=DEV, dtype=torch.float64)
    comm.allreduce(w)
    torch.cuda.synchronize()
    assert torch.equal(v, torch.arange(1000, device=DEV, dtype=torch.float32)) and torch.equal(w, torch.arange(77, device=DEV, dtype=torch.float64))
    sh = ShardedGCond(eng, 0, 1, [len(q) for q in p["pools"]], comm=comm)
    comm.set_timing(True)
    for epoch in (0, 10):
        ref.outer_step(epoch)
        sh.outer_step(epoch)
    torch.cuda.synchronize()
    rep = comm.timing_report()
    assert rep["bn2_sums"]["calls"] == 4 and rep["pair_scores"]["calls"] == 4 and rep["dfeat_dadj_loss"]["calls"] == 2, rep
    assert rep["marginals_dw1"]["calls"] == 2 and rep["bn2_bwd_sums"]["calls"] == 2, rep
    assert torch.equal(eng.feat_syn, ref.feat_syn) and torch.equal(eng.pge_params, ref.pge_params)
    assert torch.equal(eng.model_params, ref.model_params) and torch.equal(eng.adj_syn, ref.adj_syn)


def test_rccl_allreduce_on_engine_views_world1():
    """The collectives of graphslim_amd/parallel.py run on zero-copy views of library-owned buffers (fp32 and fp64):
    check that RCCL accepts them (single-rank group on this box; multi-rank equivalence is covered by the lock-step test)."""
    import os
    import torch.distributed as dist
    from graphslim_amd.parallel import Collective, ShardedGCond
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        p = _problem(2, seed=3)
        eng = make_engine(p["n"], p["nclass"], p["d"], p["hidden"], 2, p["ph"], p["labels_syn"], p["ranges"], p["x"], p["f1"],
                          p["f2"], bcap=p["bcap"], inner=1, csr=p["csr"], pools=p["pools"], external=False, seed=5)
        eng.set_feat(p["xs"])
        eng.set_pge(p["pge"]["lin"] + p["pge"]["bn"])
        eng.set_model(p["theta"])
        sh = ShardedGCond(eng, 0, 1, [len(q) for q in p["pools"]], Collective(dist))
        sh.outer_step_py(0)
        for name, v in sh.v.items():
            before = v.clone()
            dist.all_reduce(v)                      # world 1: identity, but exercises RCCL on the view
            torch.cuda.synchronize()
            assert torch.equal(v, before), name
        assert np.isfinite(float(eng.buffer("loss")[0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("variant", ["doscond", "gcondx", "doscondx"])
@pytest.mark.parametrize("t", [1, 2])
def test_sibling_methods_follow_the_oracle(variant, t):
    """DosCond / GCondX / DosCondX (doscond.py:40-50, gcondx.py:30,44-66, doscondx.py:30,42-49) reuse train_class and
    differ in the update rule / structure: six outer iterations against oracle.GCondState.outer_iteration_ex."""
    from graphslim_amd.engine import BlockSet
    metric = "ours" if variant == "doscond" else "mse"
    p = _problem(t, seed=31 + t)
    eng = make_engine(p["n"], p["nclass"], p["d"], p["hidden"], t, p["ph"], p["labels_syn"], p["ranges"], p["x"], p["f1"],
                      p["f2"], bcap=p["bcap"], metric=metric, inner=2, lrs=(0.01, 0.005, 0.02))
    st = O.GCondState(p["xs"], p["pge"], T(p["labels_syn"]), p["ncd"], p["nclass"], 0.01, 0.005, 0.02, "SGC", 2)
    eng.set_feat(p["xs"])
    eng.set_pge(p["pge"]["lin"] + p["pge"]["bn"])
    identity = variant != "doscond"
    if identity:
        eng.set_identity_structure(True)
    st.new_epoch(p["theta"])
    eng.set_model(p["theta"])
    X, Y = T(p["x"]), T(p["y"])
    for ol in range(6):
        if variant == "doscond":
            rule = dict(step_pge=True, step_feat=True, run_inference=False, run_inner=False)
        elif variant == "gcondx":
            rule = dict(step_pge=False, step_feat=not (ol % 5 < 1), run_inference=False, run_inner=True)
        else:
            rule = dict(step_pge=False, step_feat=True, run_inference=False, run_inner=False)
        batches = _batches(p, ol)
        l_ref, gfeat, _ = st.outer_iteration_ex(batches, X, Y, 2, identity=identity, dis_metric=metric, **rule)
        eng.set_blocks(BlockSet.from_batches(batches, p["bcap"], p["f1"], p["f2"], DEV))
        eng.outer_step_ex(rule["step_pge"], rule["step_feat"], rule["run_inference"], rule["run_inner"])
        loss = float(eng.buffer("loss")[0])
        assert abs(loss - float(l_ref)) <= 2e-4 * abs(float(l_ref)), (variant, ol, loss, float(l_ref))
        if rule["step_feat"]:
            close(eng.buffer("d_feat", gfeat.shape), gfeat, 2e-3, "d X~")
    np.testing.assert_allclose(eng.feat_syn.cpu().numpy(), st.feat_syn.detach().numpy(), atol=3e-3)
    for got, ref in zip(eng.get_model(), st.model):
        np.testing.assert_allclose(got.cpu().numpy(), ref.detach().numpy(), atol=4e-3)
    if identity:
        assert torch.equal(eng.adj_syn.cpu(), torch.eye(len(p["labels_syn"])))
    else:
        eng.stage_pge_forward(True)      # doscond.py:53: inference at the checkpoint
        with torch.no_grad():
            ref = O.pge_forward(st.feat_syn.detach(), {"lin": [q.detach() for q in st.pge["lin"]], "bn": [q.detach() for q in st.pge["bn"]]})
        close(eng.adj_syn.cpu(), ref, 3e-3, "adj_syn")


def test_sibling_methods_reduce_api(tmp_path):
    import graphslim  # noqa: F401
    from graphslim.config import cli
    from graphslim.condensation import DosCond, DosCondX, GCondX
    from graphslim.dataset import get_dataset
    from graphslim.evaluation.eval_agent import Evaluator
    for cls, method in ((DosCond, "doscond"), (GCondX, "gcondx"), (DosCondX, "doscondx")):
        args = cli(["-D", "pubmed", "-M", method, "-R", "0.5", "-G", "0", "--epochs", "20", "--eval_epochs", "30", "--run_eval",
                    "2", "--condense_model", "SGC", "--save_path", str(tmp_path)], standalone_mode=False)
        args.run_inter_eval = 1
        graph = get_dataset(args.dataset, args)
        __import__('graphslim_amd.host', fromlist=['seed_everything']).seed_everything(args.seed)   # train_all.py:22 (the example script itself runs unseeded)
        reduced = cls(setting=args.setting, data=graph, args=args).reduce(graph, verbose=False)
        n = reduced.feat_syn.shape[0]
        if method != "doscond":
            assert torch.equal(reduced.adj_syn.cpu(), torch.eye(n))
        mean, _ = Evaluator(args).evaluate(reduced, model_type="GCN", verbose=False)
        assert mean > 0.5, (method, mean)


def test_edge_cases_tiny_classes_and_one_row_per_class():
    """One synthetic row per class (max(int(cnt*r),1) everywhere, gcond_base.py:106), class pools smaller than the seed
    budget (down to a single train node), nodes of degree < fan-out: loss / gradients still follow the oracle."""
    from graphslim_amd.engine import BlockSet
    p = _problem(2, seed=41, n=300, nclass=6, rate=0.001, bcap=16)
    assert len(p["labels_syn"]) == 6
    p["pools"][0] = p["pools"][0][:1]            # a class with ONE train node
    p["pools"][1] = p["pools"][1][:3]
    eng = make_engine(p["n"], p["nclass"], p["d"], p["hidden"], 2, p["ph"], p["labels_syn"], p["ranges"], p["x"], p["f1"],
                      p["f2"], bcap=p["bcap"], inner=1)
    st = O.GCondState(p["xs"], p["pge"], T(p["labels_syn"]), p["ncd"], p["nclass"], 0.01, 0.01, 0.01, "SGC", 2)
    eng.set_feat(p["xs"])
    eng.set_pge(p["pge"]["lin"] + p["pge"]["bn"])
    st.new_epoch(p["theta"])
    eng.set_model(p["theta"])
    batches = _batches(p, 0)
    assert batches[0][0] == 1 and batches[1][0] == 3
    l_ref, gfeat, gpge, adj_inner = st.outer_iteration(10, batches, T(p["x"]), T(p["y"]), 1)
    eng.set_blocks(BlockSet.from_batches(batches, p["bcap"], p["f1"], p["f2"], DEV))
    eng.outer_step(10)
    loss = float(eng.buffer("loss")[0])
    assert abs(loss - float(l_ref)) <= 1e-4 * abs(float(l_ref))
    close(eng.buffer("d_feat", gfeat.shape), gfeat, 2e-3)
    close(eng.adj_syn.cpu(), adj_inner, 2e-3)


def test_retrieve_class_sampler_contract():
    """GraphData.retrieve_class_sampler (loader.py:187-224) backed by gs_sample_blocks: reference-shaped output whose blocks
    reproduce the class gradient through the oracle, i.e. the blocks are a valid targets-first two-hop sample."""
    import types
    from graphslim_amd.dataset.loader import TransAndInd, splits
    from graphslim_amd.dataset.synthetic import make_raw
    from graphslim_amd.host import normalize_adj_sparse
    data = TransAndInd(splits(make_raw("cora", seed=2)), "cora")
    args = types.SimpleNamespace(device=DEV, setting="trans", nlayers=2, dataset="cora", seed=3)
    csr = normalize_adj_sparse(data.adj_full)
    dense = O.csr_to_torch(*csr, data.num_nodes).to_dense()
    for c in (0, 4):
        bs, n_id, adjs = data.retrieve_class_sampler(c, csr, args)
        assert bs == 20 and len(adjs) == 2                      # 20 train nodes per class; outermost hop first
        (b0, e0, s0), (b1, e1, s1) = adjs
        assert e0 is None and e1 is None and s1 == (b1.shape[1], bs) and s0 == (n_id.numel(), b0.shape[0])
        n_id = n_id.cpu()
        assert set(n_id[:bs].tolist()) <= set(data.class_pools("trans")[c].tolist())
        assert len(set(n_id.tolist())) == n_id.numel()          # de-duplicated, targets first
        d0, d1 = b0.to_dense().cpu(), b1.to_dense().cpu()
        # every block entry carries the full-graph normalised weight of that edge
        n1 = d1.shape[1]
        assert torch.equal(d1[d1 != 0], dense[n_id[:bs]][:, n_id[:n1]][d1 != 0])
        assert torch.equal(d0[d0 != 0], dense[n_id[:n1]][:, n_id][d0 != 0])
        assert int((d1 != 0).sum(1).max()) <= 10 and int((d0 != 0).sum(1).max()) <= 5


@pytest.mark.parametrize("metric,L", [("ours", 2), ("mse", 2), ("ours", 3), ("mse", 3), ("cos", 3), ("ours", 4)])
def test_gcn_condense_model_follows_the_oracle(metric, L):
    """condense_model = GCN (models/gcn.py:8-23, layers.py:36-51, models/base.py:51-78): three outer iterations per optimiser against
    the oracle.  L = `nlayers` GraphConvolution layers, one per sampled hop (round 6, csrc/gs_gcnk.inc: L = 3 / 4 — a middle block per
    extra hop on the real side, the dense A^ inside every layer on the synthetic side; golden G28 pins the oracle for them)."""
    from graphslim_amd.engine import BlockSet
    p = _problem(2, seed=51)
    torch.manual_seed(5)
    theta = O.init_gcn(p["d"], p["hidden"], p["nclass"], L)
    sizes = {2: [p["f1"], p["f2"]], 3: [p["f1"], 3, p["f2"]], 4: [p["f1"], 3, 2, p["f2"]]}[L]
    fmid = tuple(sizes[1:-1])
    eng = make_engine(p["n"], p["nclass"], p["d"], p["hidden"], 2, p["ph"], p["labels_syn"], p["ranges"], p["x"], p["f1"],
                      p["f2"], bcap=p["bcap"], metric=metric, inner=2, lrs=(0.01, 0.005, 0.02), model="GCN", fmid=fmid)
    st = O.GCondState(p["xs"], p["pge"], T(p["labels_syn"]), p["ncd"], p["nclass"], 0.01, 0.005, 0.02, "GCN", L)
    eng.set_feat(p["xs"])
    eng.set_pge(p["pge"]["lin"] + p["pge"]["bn"])
    X, Y = T(p["x"]), T(p["y"])
    assert len(eng.get_model()) == 2 * L
    step = 0
    for epoch in (0, 10):
        st.new_epoch(theta)
        eng.set_model(theta)
        for _ in range(3):
            batches = []
            for c in range(p["nclass"]):
                rng = np.random.default_rng(7000 + 100 * step + c)
                seeds = rng.permutation(p["pools"][c])[: p["bcap"]]
                batches.append(O.sample_blocks(*p["csr"], seeds, sizes, rng))
            step += 1
            l_ref, gfeat, gpge, adj_inner = st.outer_iteration(epoch, batches, X, Y, 2, metric)
            eng.set_blocks(BlockSet.from_batches(batches, p["bcap"], p["f1"], p["f2"], DEV, fmid=fmid))
            if step == 1:      # the real-side class gradients of the first iteration, parameter by parameter
                eng.stage_real_grads()
                gw = eng.buffer("gw_real", (p["nclass"], -1))
                for c in range(p["nclass"]):
                    th = [q.clone().requires_grad_(True) for q in theta]
                    ref, _ = O.class_real_grads("GCN", th, X, Y, batches[c], L)
                    close(gw[c], torch.cat([r.reshape(-1) for r in ref]), 2e-4, f"gw_real class {c}")
            eng.outer_step(epoch)
            loss = float(eng.buffer("loss")[0])
            assert abs(loss - float(l_ref)) <= 1e-4 * abs(float(l_ref)), (epoch, loss, float(l_ref))
            if epoch % 50 >= 10:
                close(eng.buffer("d_feat", gfeat.shape), gfeat, 1e-3, "d X~")
            close(eng.adj_syn.cpu(), adj_inner, 2e-3, "adj_syn after the step")
        np.testing.assert_allclose(eng.feat_syn.cpu().numpy(), st.feat_syn.detach().numpy(), atol=2e-3)
        for got, ref in zip(eng.get_model(), st.model):
            np.testing.assert_allclose(got.cpu().numpy(), ref.detach().numpy(), atol=4e-3)


def test_gcn_condense_model_reduce_api(tmp_path):
    """The Planetoid configs of the sibling methods use condense_model = GCN (configs/doscond/cora.json)."""
    import graphslim  # noqa: F401
    from graphslim.config import cli
    from graphslim.condensation import DosCond
    from graphslim.dataset import get_dataset
    from graphslim.evaluation.eval_agent import Evaluator
    args = cli(["-D", "cora", "-M", "doscond", "-R", "0.5", "-G", "0", "--epochs", "20", "--eval_epochs", "30", "--run_eval", "2",
                "--save_path", str(tmp_path)], standalone_mode=False)
    assert args.condense_model == "GCN" and args.dis_metric == "mse" and args.outer_loop == 3
    args.run_inter_eval = 1
    graph = get_dataset(args.dataset, args)
    __import__('graphslim_amd.host', fromlist=['seed_everything']).seed_everything(args.seed)   # train_all.py:22 (the example script itself runs unseeded)
    reduced = DosCond(setting=args.setting, data=graph, args=args).reduce(graph, verbose=False)
    assert torch.isfinite(reduced.feat_syn).all() and torch.isfinite(reduced.adj_syn).all()
    adj = reduced.adj_syn.cpu()
    assert torch.equal(adj, adj.T) and 0 <= float(adj.min()) and float(adj.max()) < 1
    mean, _ = Evaluator(args).evaluate(reduced, model_type="GCN", verbose=False)
    # 60 Adam steps of size lr = 0.01 on L1-normalised bag-of-words rows (entries ~0.05) are still in the noisy early
    # phase of this config (upstream runs 1000 epochs): only sanity here, parity is covered by the tests above
    assert 0.0 <= mean <= 1.0


def test_with_bn_reduce_api(tmp_path):
    """`--with_bn` through the reference's own entry points: GCond.reduce with an SGC (ntrans = 2) condensation model that carries
    the eval-mode BatchNorm (gcond_base.py:261-285), checkpoints and the final Evaluator run with the evaluation GCN built
    without BatchNorm (upstream's fails there, golden G17).  Sanity of the whole path; parity is test_with_bn_against_reference_golden."""
    import graphslim  # noqa: F401
    from graphslim.config import cli
    from graphslim.condensation import GCond
    from graphslim.dataset import get_dataset
    from graphslim.evaluation.eval_agent import Evaluator
    args = cli(["-D", "cora", "-M", "gcond", "-R", "0.5", "-G", "0", "--epochs", "12", "--eval_epochs", "30", "--run_eval", "2",
                "--with_bn", "--ntrans", "2", "--save_path", str(tmp_path)], standalone_mode=False)
    assert args.with_bn and args.condense_model == "SGC" and args.ntrans == 2
    args.run_inter_eval, args.checkpoints, args.outer_loop, args.inner_loop = 1, [5, 11], 3, 2
    graph = get_dataset(args.dataset, args)
    __import__('graphslim_amd.host', fromlist=['seed_everything']).seed_everything(args.seed)
    agent = GCond(setting=args.setting, data=graph, args=args)
    reduced = agent.reduce(graph, verbose=False)
    assert torch.isfinite(reduced.feat_syn).all() and torch.isfinite(reduced.adj_syn).all()
    th = agent.engine.get_model()
    assert len(th) == 6 and th[4].shape == (args.hidden,) and torch.isfinite(th[4]).all()
    assert float((th[4] - 1).abs().max()) > 0 and float(th[5].abs().max()) > 0      # bn.weight / bn.bias were trained by the inner steps
    mean, _ = Evaluator(args).evaluate(reduced, model_type="GCN", verbose=False)
    assert 0.0 <= mean <= 1.0


def test_dropout_reduce_api(tmp_path):
    """`--dropout 0.3` through the reference's own entry points: GCond.reduce with the SGC (ntrans = 2) condensation model under dropout
    (models/sgc.py:42) — every outer iteration advances the masks' counter —, checkpoints and the final Evaluator run (evaluation models
    run without dropout: models/base.py:27-29).  Sanity of the whole path; parity is test_dropout_train_class_and_inner_steps and
    test_outer_iterations_with_dropout_follow_the_oracle.  The same with `--condense_model GCN` (two and three layers, round 6:
    models/base.py:59; parity is test_gcn_dropout_train_class_and_inner_steps).  with_bn refuses a non-zero value."""
    import graphslim  # noqa: F401
    from graphslim.config import cli
    from graphslim.condensation import GCond
    from graphslim.dataset import get_dataset
    from graphslim.evaluation.eval_agent import Evaluator
    argv = ["-D", "cora", "-M", "gcond", "-R", "0.5", "-G", "0", "--epochs", "12", "--eval_epochs", "30", "--run_eval", "2",
            "--ntrans", "2", "--save_path", str(tmp_path)]
    feats = []
    for extra in (["--dropout", "0.3"], []):
        args = cli(argv + extra, standalone_mode=False)
        assert args.dropout == (0.3 if extra else 0.0) and args.condense_model == "SGC"
        args.run_inter_eval, args.checkpoints, args.outer_loop, args.inner_loop = 1, [5, 11], 3, 2
        graph = get_dataset(args.dataset, args)
        __import__('graphslim_amd.host', fromlist=['seed_everything']).seed_everything(args.seed)
        agent = GCond(setting=args.setting, data=graph, args=args)
        reduced = agent.reduce(graph, verbose=False)
        assert torch.isfinite(reduced.feat_syn).all() and torch.isfinite(reduced.adj_syn).all()
        assert agent.engine.shapes.dropout == args.dropout and agent.engine.dropout_iter() == 12 * 3
        feats.append(reduced.feat_syn.clone())
        mean, _ = Evaluator(args).evaluate(reduced, model_type="GCN", verbose=False)
        assert 0.0 <= mean <= 1.0
    assert not torch.equal(feats[0], feats[1])      # the masks entered the condensation
    for nl in (2, 3):
        args = cli(argv + ["--dropout", "0.3", "--condense_model", "GCN", "--hidden", "64", "--nlayers", str(nl)], standalone_mode=False)
        args.run_inter_eval, args.checkpoints, args.outer_loop, args.inner_loop = 1, [5, 11], 2, 2
        graph = get_dataset(args.dataset, args)
        __import__('graphslim_amd.host', fromlist=['seed_everything']).seed_everything(args.seed)
        agent = GCond(setting=args.setting, data=graph, args=args)
        reduced = agent.reduce(graph, verbose=False)
        assert torch.isfinite(reduced.feat_syn).all() and torch.isfinite(reduced.adj_syn).all()
        assert agent.engine.shapes.dropout == args.dropout and agent.engine.dropout_iter() == 12 * 2 and len(agent.engine.get_model()) == 2 * nl
        mean, _ = Evaluator(args).evaluate(reduced, model_type="GCN", verbose=False)
        assert 0.0 <= mean <= 1.0
    bad = cli(argv + ["--dropout", "0.3", "--with_bn"], standalone_mode=False)
    graph = get_dataset(bad.dataset, bad)
    with pytest.raises(NotImplementedError):
        GCond(setting=bad.setting, data=graph, args=bad).reduce(graph, verbose=False)


@pytest.mark.parametrize("ntrans", [1, 2])
def test_three_hops_reduce_api(tmp_path, ntrans):
    """`--nlayers 3` through the reference's own entry points (round 5): GCond.reduce samples [15, 10, 5] neighbours over three hops
    (dataset/loader.py:205-206), the SGC condensation model propagates three times on both sides (models/sgc.py:45-51), checkpoints
    and the final Evaluator run train the THREE-layer evaluation GCN (models/gcn.py:8-23: the flag is shared).  Sanity of the whole
    path; parity is G24 / G25 and test_three_hops_against_the_oracle.  nlayers = 1 is refused like upstream fails (gcond_base.py:213)."""
    import graphslim  # noqa: F401
    from graphslim.config import cli
    from graphslim.condensation import GCond
    from graphslim.dataset import get_dataset
    from graphslim.evaluation.eval_agent import Evaluator
    argv = ["-D", "cora", "-M", "gcond", "-R", "0.5", "-G", "0", "--epochs", "12", "--eval_epochs", "30", "--run_eval", "2",
            "--ntrans", str(ntrans), "--save_path", str(tmp_path)]
    args = cli(argv + ["--nlayers", "3"], standalone_mode=False)
    assert args.nlayers == 3 and args.condense_model == "SGC"
    args.run_inter_eval, args.checkpoints, args.outer_loop, args.inner_loop = 1, [5, 11], 3, 2
    graph = get_dataset(args.dataset, args)
    __import__('graphslim_amd.host', fromlist=['seed_everything']).seed_everything(args.seed)
    agent = GCond(setting=args.setting, data=graph, args=args)
    reduced = agent.reduce(graph, verbose=False)
    assert agent.engine.shapes.nlayers == 3 and agent.engine.shapes.fmid == (10,)
    assert torch.isfinite(reduced.feat_syn).all() and torch.isfinite(reduced.adj_syn).all()
    blk = agent.engine.get_blocks()
    assert blk.fmid == (10,) and int(blk.n_last.min()) >= int(blk.n1.min()) > 0
    mean, _ = Evaluator(args).evaluate(reduced, model_type="GCN", verbose=False)
    assert 0.0 <= mean <= 1.0
    args1 = cli(argv + ["--nlayers", "1"], standalone_mode=False)
    graph1 = get_dataset(args1.dataset, args1)
    with pytest.raises(NotImplementedError):
        GCond(setting=args1.setting, data=graph1, args=args1).reduce(graph1, verbose=False)


@pytest.mark.parametrize("nlayers", [3, 4])
def test_gcn_condense_model_with_more_layers_reduce_api(tmp_path, nlayers):
    """`--condense_model GCN --nlayers 3 / 4` through the reference's own entry points (round 6; models/gcn.py:8-23: one
    GraphConvolution per layer, models/base.py:51-78: layer i takes block i of the `nlayers` sampled hops): sanity of the whole path —
    parity is G28 and test_gcn_condense_model_follows_the_oracle; five layers are refused."""
    import graphslim  # noqa: F401
    from graphslim.config import cli
    from graphslim.condensation import GCond
    from graphslim.dataset import get_dataset
    from graphslim.evaluation.eval_agent import Evaluator
    argv = ["-D", "cora", "-M", "gcond", "-R", "0.5", "-G", "0", "--epochs", "12", "--eval_epochs", "30", "--run_eval", "2",
            "--condense_model", "GCN", "--hidden", "64", "--save_path", str(tmp_path)]
    args = cli(argv + ["--nlayers", str(nlayers)], standalone_mode=False)
    args.run_inter_eval, args.checkpoints, args.outer_loop, args.inner_loop = 1, [5, 11], 2, 2
    graph = get_dataset(args.dataset, args)
    __import__('graphslim_amd.host', fromlist=['seed_everything']).seed_everything(args.seed)
    agent = GCond(setting=args.setting, data=graph, args=args)
    reduced = agent.reduce(graph, verbose=False)
    assert agent.engine.shapes.nlayers == nlayers and len(agent.engine.get_model()) == 2 * nlayers
    assert agent.engine.get_model()[2].shape == (64, 64)
    assert torch.isfinite(reduced.feat_syn).all() and torch.isfinite(reduced.adj_syn).all()
    mean, _ = Evaluator(args).evaluate(reduced, model_type="GCN", verbose=False)
    assert 0.0 <= mean <= 1.0
    if nlayers == 3:
        bad = cli(argv + ["--nlayers", "5"], standalone_mode=False)
        with pytest.raises(NotImplementedError):
            GCond(setting=bad.setting, data=graph, args=bad).reduce(graph, verbose=False)


def test_full_neighbour_mode_against_the_oracle_on_full_blocks():
    """Full-neighbour mode (SURVEY section 8e "mode F"; loader.py:204 `sizes = [-1, -1]`) for SGC with one transform: the engine
    draws the class seeds, gathers rows of S_full = A^ A^ X / rs_full = A^ A^ 1 (computed once with gs_spmm_csr_f32) — against the
    oracle's train_class on the 2-hop FULL-neighbourhood blocks of the same seeds (oracle.sample_blocks with fan-out -1):
    real-side gradients per class, matching loss, d loss / d X~, and two whole outer iterations.  Also: the row-sharded
    precompute (two and three row blocks) is bitwise the single-block one, and ntrans = 2 is refused."""
    from graphslim_amd.condensation.full_neighbour import full_neighbour_operators
    from graphslim_amd.engine import spmm_csr
    from graphslim_amd.parallel import RowShard
    p = _problem(1, seed=31, n=600, nclass=5, d=12, bcap=16)
    eng = make_engine(p["n"], p["nclass"], p["d"], p["hidden"], 1, p["ph"], p["labels_syn"], p["ranges"], p["x"], p["f1"], p["f2"],
                      bcap=p["bcap"], inner=2, csr=p["csr"], pools=p["pools"], external=False, seed=5)
    S_full, rs_full, _ = full_neighbour_operators(p["csr"], torch.as_tensor(p["x"]), DEV)
    # the operators themselves against scipy-free dense arithmetic
    rp, col, val = p["csr"]
    A = torch.zeros(p["n"], p["n"], dtype=torch.float64)
    rows = np.repeat(np.arange(p["n"]), np.diff(rp))
    A[torch.as_tensor(rows), torch.as_tensor(col)] = torch.as_tensor(val).double()
    ref_S = A @ (A @ torch.as_tensor(p["x"]).double())
    close(S_full[:, :p["d"]].cpu(), ref_S, 2e-5, "A^ A^ X")
    close(rs_full.cpu(), (A @ A.sum(1)), 2e-5, "A^ A^ 1")
    # row-sharded precompute: per-block products concatenated == the single-block result, bit for bit
    for world in (2, 3):
        cur = S_full.new_zeros(S_full.shape)
        cur[:, :p["d"]] = torch.as_tensor(p["x"]).to(DEV)
        cur[:, p["d"]] = 1.0
        for _ in range(2):
            parts = []
            for lo, hi in RowShard.split_rows(rp, world):
                i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int32).to(DEV)
                parts.append(spmm_csr(i32(rp[lo:hi + 1] - rp[lo]), i32(col[rp[lo]:rp[hi]]),
                                      torch.as_tensor(np.ascontiguousarray(val[rp[lo]:rp[hi]])).to(DEV), cur, F=cur.shape[1]))
            cur = torch.cat(parts)
        assert torch.equal(cur, S_full), world
    eng.set_full_neighbour(S_full, rs_full)
    eng.set_feat(p["xs"])
    eng.set_pge(p["pge"]["lin"] + p["pge"]["bn"])
    st = O.GCondState(p["xs"].clone(), {"lin": [t.clone() for t in p["pge"]["lin"]], "bn": [t.clone() for t in p["pge"]["bn"]]},
                      torch.as_tensor(p["labels_syn"]), p["ncd"], p["nclass"], 0.01, 0.01, 0.01, "SGC", 2)
    eng.set_model(p["theta"])
    st.new_epoch([t.clone() for t in p["theta"]])
    for epoch in (10, 0):
        eng.stage_sample()
        blk = eng.get_blocks()
        nseed, nid = blk.n_seed.cpu().numpy(), blk.nid1.cpu().numpy()
        batches = []
        for c in range(p["nclass"]):
            seeds = nid[c, :nseed[c]]
            assert len(set(seeds.tolist())) == len(seeds) and set(seeds.tolist()) <= set(p["pools"][c].tolist())
            batches.append(O.sample_blocks(*p["csr"], seeds, [-1, -1], np.random.default_rng(0)))
        eng.stage_real_grads()
        gw = eng.buffer("gw_real", (p["nclass"], -1))
        for c in range(p["nclass"]):
            th = [q.detach().clone().requires_grad_(True) for q in st.model]
            ref, _ = O.class_real_grads("SGC", th, torch.as_tensor(p["x"]), torch.as_tensor(p["y"]), batches[c], 2)
            close(gw[c], torch.cat([r.reshape(-1) for r in ref]), 2e-4, f"gw_real class {c}")
        eng.stage_pge_forward(False)
        eng.stage_syn_match()
        eng.syn_finish()
        loss = float(eng.buffer("loss")[0])
        eng.stage_pge_backward(True, True)
        g_feat = eng.buffer("d_feat", tuple(p["xs"].shape)).clone()
        eng.stage_update(epoch)
        eng.stage_inner()
        ref_loss, ref_gfeat, _, _ = st.outer_iteration(epoch, batches, torch.as_tensor(p["x"]), torch.as_tensor(p["y"]), 2)
        assert abs(loss - float(ref_loss)) <= 1e-4 * abs(float(ref_loss)), (epoch, loss, float(ref_loss))
        close(g_feat, ref_gfeat, 1e-3, "d loss / d X~")
        close(eng.feat_syn.cpu(), st.feat_syn.detach(), 2e-3, "X~ after the step")
    # a ReLU between two transforms breaks the linearity the mode rests on
    e2 = make_engine(p["n"], p["nclass"], p["d"], p["hidden"], 2, p["ph"], p["labels_syn"], p["ranges"], p["x"], p["f1"], p["f2"],
                     bcap=p["bcap"], csr=p["csr"], pools=p["pools"], external=False)
    with pytest.raises(RuntimeError):
        e2.set_full_neighbour(S_full, rs_full)


def test_full_neighbour_reduce_api(tmp_path):
    """args.neighbour_mode = 'full' through GCond.reduce + Evaluator.evaluate on the cora-shaped config (SGC, ntrans = 1): the run
    must condense to a graph that evaluates at least as well as chance by a wide margin, like the sampled mode does; ntrans = 2 is
    refused with the reason."""
    import graphslim  # noqa: F401
    from graphslim.config import cli
    from graphslim.condensation import GCond
    from graphslim.dataset import get_dataset
    from graphslim.evaluation.eval_agent import Evaluator
    args = cli(["-D", "cora", "-M", "gcond", "-R", "0.5", "-G", "0", "--epochs", "30", "--eval_epochs", "100", "--run_eval", "2",
                "--save_path", str(tmp_path)], standalone_mode=False)
    assert args.ntrans == 1 and args.condense_model == "SGC"
    args.run_inter_eval, args.checkpoints, args.neighbour_mode = 1, [29], "full"
    graph = get_dataset(args.dataset, args)
    __import__('graphslim_amd.host', fromlist=['seed_everything']).seed_everything(args.seed)
    reduced = GCond(setting=args.setting, data=graph, args=args).reduce(graph, verbose=False)
    assert torch.isfinite(reduced.feat_syn).all() and torch.isfinite(reduced.adj_syn).all()
    mean, _ = Evaluator(args).evaluate(reduced, model_type="GCN", verbose=False)
    assert mean > 2.0 / graph.nclass, mean
    args.ntrans = 2
    with pytest.raises(NotImplementedError, match="ntrans = 1"):
        GCond(setting=args.setting, data=graph, args=args).reduce(graph, verbose=False)


def test_full_size_arxiv_properties():
    """BASELINE.json configs[3] at FULL size (N = 169 343, n' = 454, C = 40), where the CPU oracle needs seconds per
    iteration: size-independent properties instead of an oracle comparison —
      * the closed-form total derivative d loss / d X~ (direct + through the PGE) agrees with central finite differences of
        the engine's own loss on fixed blocks and theta;
      * adj_syn is symmetric with a zero diagonal and entries in [0, 1);
      * two engines with the same seed produce bitwise identical trajectories (no float atomics anywhere)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    p = bench.build_problem("ogbn-arxiv", seed=1)

    def fresh():
        eng = bench.make_engine(p, DEV)
        bench.init_state(p, eng)
        bench.new_model(p, eng)
        return eng

    eng = fresh()
    eng.stage_sample()
    eng.stage_real_grads()

    def loss_at(x):
        eng.set_feat(x)
        eng.stage_pge_forward(False)
        eng.stage_syn_match()
        eng.syn_finish()
        return float(eng.buffer("loss")[0])

    x0 = eng.feat_syn.clone()
    l0 = loss_at(x0)
    eng.stage_pge_backward(False, True)
    g = eng.buffer("d_feat", tuple(x0.shape)).clone()
    assert np.isfinite(l0) and l0 > 0 and bool(torch.isfinite(g).all())
    flat = g.abs().flatten()
    picks = torch.topk(flat, 3).indices.tolist() + [int(flat.numel() // 2)]
    for idx in picks:
        i, j = divmod(idx, x0.shape[1])
        h = 2e-2 * max(1.0, float(x0[i, j].abs()))
        xp, xm = x0.clone(), x0.clone()
        xp[i, j] += h
        xm[i, j] -= h
        fd = (loss_at(xp) - loss_at(xm)) / (2 * h)
        an = float(g[i, j])
        # fp32 loss of magnitude ~1e2 differenced over 2h: absolute noise ~1e-5 * l0 / h
        assert abs(fd - an) <= 0.05 * abs(an) + 2e-5 * l0 / h, (i, j, fd, an)
    # the same for two PGE parameters (a layer-2 weight and a BN-2 scale): d loss / d theta_PGE
    loss_at(x0)
    eng.stage_pge_backward(True, False)
    pg = eng.buffer("d_pge", (eng.pge_params.numel(),)).clone()
    Ph, d = eng.shapes.pge_hidden, x0.shape[1]
    w1_off = Ph * 2 * d + Ph
    cand = pg[w1_off:w1_off + Ph * Ph].abs()
    k1 = w1_off + int(torch.argmax(cand))
    k2 = int(eng.pge_params.numel()) - 2 * Ph + int(torch.argmax(pg[-2 * Ph:-Ph].abs()))      # bns.1.weight
    for k in (k1, k2):
        v0 = float(eng.pge_params[k])
        h = 1e-2 * max(1.0, abs(v0))
        eng.pge_params[k] = v0 + h
        lp = loss_at(x0)
        eng.pge_params[k] = v0 - h
        lm = loss_at(x0)
        eng.pge_params[k] = v0
        fd, an = (lp - lm) / (2 * h), float(pg[k])
        assert abs(fd - an) <= 0.05 * abs(an) + 2e-5 * l0 / h, (k, fd, an)
    loss_at(x0)

    def run():
        e = fresh()
        out = []
        for it in range(3):
            e.outer_step(0 if it == 0 else 10)
            out.append(float(e.buffer("loss")[0]))
        return e, out

    e1, l1 = run()
    e2, l2 = run()
    assert l1 == l2 and torch.equal(e1.feat_syn, e2.feat_syn) and torch.equal(e1.pge_params, e2.pge_params)
    a = e1.adj_syn
    assert torch.equal(a, a.T) and float(a.diagonal().abs().max()) == 0.0 and float(a.min()) >= 0.0 and float(a.max()) < 1.0


@pytest.mark.parametrize("n,ph", [(13, 128), (37, 256)])
def test_chunked_pge_forward_backward(golden_dir, n, ph):
    """pge_chunks = 5 (parametrized_adj.py:41-55) on the engine: adjacency, d/dX~ and d/dtheta_PGE of a linear functional of
    the raw adjacency against the oracle's autograd (the oracle itself is pinned by golden G11, n = 13: that case also
    against the reference's own numbers)."""
    d, C = 6, 3
    torch.manual_seed(4)
    labels_syn = np.sort(np.arange(n) % C)
    ranges = [(int(np.searchsorted(labels_syn, c)), int(np.searchsorted(labels_syn, c, side="right"))) for c in range(C)]
    from graphslim_amd.engine import EngineShapes, GCondEngine
    sh = EngineShapes(50, d, C, n, 1, 8, ph, 8, 4, 3, pge_chunks=5)
    eng = GCondEngine(sh, torch.randn(50, d), None, None, None, labels_syn, ranges, lr_feat=0.01, lr_adj=0.01, lr_model=0.01,
                      inner_loop=0, dis_metric="ours", seed=1, external_blocks=True, device=DEV)
    if n == 13:
        g = load_golden(golden_dir, "g11_pge_chunked")
        x0, w = T(g["x"]), T(g["w"])
        params = [T(g[f"p{i}"]) for i in range(10)]
    else:
        x0, w = torch.randn(n, d), torch.randn(n, n)
        pge = O.init_pge(d, ph)
        params = list(pge["lin"]) + [torch.rand(ph) + 0.5, torch.randn(ph) * 0.2, torch.rand(ph) + 0.5, torch.randn(ph) * 0.2]
    eng.set_feat(x0)
    eng.set_pge(params)
    eng.stage_pge_forward(False)
    x = x0.clone().requires_grad_(True)
    lin = [q.clone().requires_grad_(True) for q in params[:6]]
    bn = [q.clone().requires_grad_(True) for q in params[6:]]
    ref = O.pge_forward(x, {"lin": lin, "bn": bn}, n_chunks=5)
    close(eng.buffer("adj_raw", (n, n)), ref.detach(), 2e-4, "chunked adj")
    if n == 13:
        close(eng.buffer("adj_raw", (n, n)), g["adj"], 2e-4, "chunked adj vs the reference")
    (ref * w).sum().backward()
    eng.view("d_adj").copy_(w.reshape(-1).to(DEV))
    eng.view("d_feat").zero_()
    eng.stage_pge_backward(True, True)
    close(eng.buffer("d_feat", (n, d)), x.grad, 2e-3, "d X~ through the chunked PGE")
    flat = eng.buffer("d_pge", (eng.pge_params.numel(),)).cpu()
    off = 0
    for i, q in enumerate(lin + bn):
        got = flat[off:off + q.numel()].reshape(q.shape)
        off += q.numel()
        if i in (1, 3):
            continue
        close(got, q.grad, 3e-3, f"d theta_PGE[{i}]")
    eng.stage_pge_forward(True)          # inference pass: same function
    close(eng.adj_syn.cpu(), ref.detach(), 2e-4, "chunked inference")
