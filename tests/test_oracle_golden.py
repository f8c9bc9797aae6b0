"""Pins oracle/gcond_oracle.py against fixtures produced by running the reference itself
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import gcond_oracle as O

torch.set_num_threads(4)


from helpers import T, load_golden as load, rebuild_batches as _rebuild


def test_g1_normalisation(golden_dir):
    g = load(golden_dir, "g1_norm")
    n = int(g["n"])
    a = sp.coo_matrix((np.ones(len(g["a_row"])), (g["a_row"], g["a_col"])), shape=(n, n)).tocsr()
    rp, col, val = O.normalize_adj_sparse(a)
    dense = O.csr_to_torch(rp, col, val, n).to_dense().numpy()
    np.testing.assert_array_equal(dense, g["sparse_norm_dense"])   # f64 scipy -> fp32: bit exact
    dn = O.dense_gcn_norm(T(g["adj_d"]))
    np.testing.assert_allclose(dn.numpy(), g["dense_norm"], rtol=1e-6, atol=1e-7)


def test_g2_labels_syn(golden_dir):
    g = load(golden_dir, "g2_labels")
    names = sorted({k.split("__")[0] for k in g.files})
    assert len(names) == 8
    for name in names:
        ls, ncd, _ = O.generate_labels_syn(g[name + "__labels_train"], float(g[name + "__rate"]))
        np.testing.assert_array_equal(ls, g[name + "__labels_syn"])
        np.testing.assert_array_equal(np.array(list(ncd.keys())), g[name + "__dict_keys"])
        np.testing.assert_array_equal(np.array(list(ncd.values())), g[name + "__dict_vals"])
    assert len(g["cora_0.5__labels_syn"]) == 70 and len(g["cora_0.25__labels_syn"]) == 35
    assert len(g["cora_0.1__labels_syn"]) == 14 and len(g["citeseer_0.005__labels_syn"]) == 6


def test_g3_init_draw_order(golden_dir):
    g = load(golden_dir, "g3_init")
    torch.manual_seed(123)
    w, b = O.init_linear(6, 4)
    w2, b2 = O.init_linear(4, 3)
    for got, key in ((w, "mylinear_w"), (b, "mylinear_b"), (w2, "gconv_w"), (b2, "gconv_b")):
        np.testing.assert_array_equal(got.numpy(), g[key])
    torch.manual_seed(77)
    O.init_sgc(6, 5, 3, 2)            # constructor draw
    p = O.init_sgc(6, 5, 3, 2)        # initialize() draw
    for i, t in enumerate(p):
        np.testing.assert_array_equal(t.numpy(), g[f"sgc2_p{i}"])
    torch.manual_seed(99)
    pge = O.init_pge_reference_order(6, 128)
    names = [str(g[f"pge_name{i}"]) for i in range(10)]
    assert names[:6] == ["layers.0.weight", "layers.0.bias", "layers.1.weight", "layers.1.bias",
                         "layers.2.weight", "layers.2.bias"]
    for i, t in enumerate(pge["lin"] + pge["bn"]):
        np.testing.assert_array_equal(t.numpy(), g[f"pge_p{i}"])


def test_g4_model_forward(golden_dir):
    g = load(golden_dir, "g4_models")
    x = T(g["x"])
    blocks = [O.block_to_torch((g[f"blk{k}_rowptr"], g[f"blk{k}_col"], g[f"blk{k}_val"], tuple(g[f"blk{k}_shape"])))
              for k in range(2)]
    n_id = T(g["n_id"])
    adj = T(g["adj_dense"])
    for kind in ("SGC", "GCN"):
        for t in (1, 2):
            tag = f"{kind}_T{t}"
            params = [T(g[f"{tag}_p{i}"]) for i in range(2 * (t if kind == "SGC" else 2))]
            np.testing.assert_allclose(O.model_forward(kind, x, adj, params, 2).numpy(), g[f"{tag}_dense"],
                                       rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(O.model_forward(kind, x[n_id], blocks, params, 2).numpy(), g[f"{tag}_blocks"],
                                       rtol=2e-5, atol=2e-6)


def test_g5_match_loss(golden_dir):
    g = load(golden_dir, "g5_match")
    gr = [T(g[f"gr{i}"]) for i in range(4)]
    gs = [T(g[f"gs{i}"]).requires_grad_(True) for i in range(4)]
    for metric in ("ours", "mse", "cos"):
        np.testing.assert_allclose(float(O.match_loss(gs, gr, metric)), float(g[f"loss_{metric}"]), rtol=1e-6)
    O.match_loss(gs, gr, "ours").backward()
    np.testing.assert_allclose(gs[0].grad.numpy(), g["d_ours_dgs0"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(gs[2].grad.numpy(), g["d_ours_dgs2"], rtol=1e-5, atol=1e-7)
    with pytest.raises(SystemExit):
        O.match_loss(gs, gr, "nope")


@pytest.mark.parametrize("tag", ["small", "h256"])
def test_g6_pge(golden_dir, tag):
    g = load(golden_dir, "g6_pge")
    x = T(g[f"{tag}_x"]).requires_grad_(True)
    ps = [T(g[f"{tag}_p{i}"]).requires_grad_(True) for i in range(10)]
    pge = {"lin": ps[:6], "bn": ps[6:]}
    adj = O.pge_forward(x, pge)
    np.testing.assert_allclose(adj.detach().numpy(), g[f"{tag}_adj"], rtol=1e-5, atol=1e-6)
    assert torch.equal(adj, adj.T) and float(adj.diagonal().abs().max()) == 0.0
    (adj * T(g[f"{tag}_w"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g[f"{tag}_dx"], rtol=2e-3, atol=2e-6)
    for i, p in enumerate(ps):
        ref = g[f"{tag}_dp{i}"]
        np.testing.assert_allclose(p.grad.numpy(), ref, rtol=2e-3, atol=1e-5 * max(1.0, np.abs(ref).max()))
    with torch.no_grad():
        np.testing.assert_allclose(O.pge_forward(x.detach(), pge).numpy(), g[f"{tag}_inference"], rtol=1e-5, atol=1e-6)


def rebuild_batches(g, tag):
    batches, dims, _ = _rebuild(g, tag)
    return batches, dims


@pytest.mark.parametrize("tag,kind,metric", [("t1", "SGC", "ours"), ("t2", "SGC", "ours"), ("t1mse", "SGC", "mse"),
                                             ("gcn", "GCN", "ours"), ("k3t1", "SGC", "ours"), ("k3t2", "SGC", "ours"),
                                             ("t3", "SGC", "ours"), ("t4", "SGC", "ours"), ("gcn3", "GCN", "ours"), ("gcn4", "GCN", "ours")])
def test_g7_g24_train_class(golden_dir, tag, kind, metric):
    """G7: the reference's train_class + backward (two hops); G24 (tags k3*): the same with `--nlayers 3` — three sampled hops on the
    real side, adj_syn applied three times on the synthetic side (dataset/loader.py:205-206, models/sgc.py:45-51); G27 (tags t3, t4):
    `--ntrans` 3 / 4 — hidden-to-hidden transforms between the first and the last (models/sgc.py:32-35); G28 (tags gcn3, gcn4): the GCN
    condensation model with three / four GraphConvolution layers, one per sampled hop (models/gcn.py:8-23, models/base.py:51-78)."""
    fname = {"k3t1": "g24_train_class_three_hops", "k3t2": "g24_train_class_three_hops", "t3": "g27_train_class_deep", "t4": "g27_train_class_deep",
             "gcn3": "g28_train_class_gcn_layers", "gcn4": "g28_train_class_gcn_layers"}.get(tag, "g7_train_class")
    g = load(golden_dir, fname)
    nlayers = len(g[f"{tag}_sizes"])
    batches, (n, nclass, d, hidden, t) = rebuild_batches(g, tag)
    nth = 2 * max(2, nlayers) if kind == "GCN" else 2 * t
    theta = [T(g[f"{tag}_theta{i}"]).requires_grad_(True) for i in range(nth)]
    ps = [T(g[f"{tag}_pge{i}"]).requires_grad_(True) for i in range(10)]
    pge = {"lin": ps[:6], "bn": ps[6:]}
    feat_syn = T(g[f"{tag}_feat_syn"]).requires_grad_(True)
    labels_syn = T(g[f"{tag}_labels_syn"])
    _, ncd, _ = O.generate_labels_syn(g[f"{tag}_y"][g[f"{tag}_idx_train"]], float(g[f"{tag}_rate"]))
    adj_syn = O.pge_forward(feat_syn, pge)
    np.testing.assert_allclose(adj_syn.detach().numpy(), g[f"{tag}_adj_syn"], rtol=1e-5, atol=1e-6)
    adj_norm = O.dense_gcn_norm(adj_syn)
    adj_norm.retain_grad()
    loss = O.train_class(kind, theta, batches, T(g[f"{tag}_x"]), T(g[f"{tag}_y"]), feat_syn, adj_norm, labels_syn,
                         ncd, nclass, nlayers, metric)
    loss.backward()
    # north_star gate: gradient-match loss within 1e-4 relative
    np.testing.assert_allclose(float(loss), float(g[f"{tag}_loss"]), rtol=1e-4)

    def close(a, ref, rtol=2e-3):
        np.testing.assert_allclose(a, ref, rtol=rtol, atol=1e-4 * np.abs(ref).max())

    close(adj_norm.grad.numpy(), g[f"{tag}_dadj_norm"])
    close(feat_syn.grad.numpy(), g[f"{tag}_dfeat"])
    for i, p in enumerate(ps):
        ref = g[f"{tag}_dpge{i}"]
        if np.abs(ref).max() > 1e-6:     # Linear biases feeding BatchNorm have ~0 gradient (noise)
            close(p.grad.numpy(), ref)


@pytest.mark.parametrize("fname,tag,metric", [("g16_with_bn", "bn", "ours"), ("g23_with_bn_mse_cos", "bn_mse", "mse"),
                                               ("g23_with_bn_mse_cos", "bn_cos", "cos")])
def test_g16_g23_with_bn_train_class_and_inner_steps(golden_dir, fname, tag, metric):
    """with_bn (gcond_base.py:261-285, models/sgc.py:27-33): the reference's train_class on an SGC with an eval-mode BatchNorm
    between its two transforms (trainable weight / bias away from (1, 0)), and two inner steps with torch Adam — under 'ours' (G16;
    the 1-D BatchNorm parameters contribute no distance, utils.py:98-101) and under 'mse' / 'cos' (G23; every parameter's gradient
    enters the distance, condensation/utils.py:41-60)."""
    g = load(golden_dir, fname)
    assert [str(v) for v in g[f"{tag}_param_names"]] == ["layers.0.weight", "layers.0.bias", "layers.1.weight", "layers.1.bias",
                                                      "bns.0.weight", "bns.0.bias"]
    np.testing.assert_array_equal(g[f"{tag}_bn_running"], np.stack([np.zeros(12, np.float32), np.ones(12, np.float32)]))
    batches, (n, nclass, d, hidden, t) = rebuild_batches(g, tag)
    theta = [T(g[f"{tag}_theta{i}"]).requires_grad_(True) for i in range(6)]
    ps = [T(g[f"{tag}_pge{i}"]).requires_grad_(True) for i in range(10)]
    pge = {"lin": ps[:6], "bn": ps[6:]}
    feat_syn = T(g[f"{tag}_feat_syn"]).requires_grad_(True)
    labels_syn = T(g[f"{tag}_labels_syn"])
    _, ncd, _ = O.generate_labels_syn(g[f"{tag}_y"][g[f"{tag}_idx_train"]], float(g[f"{tag}_rate"]))
    adj_norm = O.dense_gcn_norm(O.pge_forward(feat_syn, pge))
    adj_norm.retain_grad()
    loss = O.train_class("SGC_BN", theta, batches, T(g[f"{tag}_x"]), T(g[f"{tag}_y"]), feat_syn, adj_norm, labels_syn, ncd, nclass,
                         2, metric)
    loss.backward()
    np.testing.assert_allclose(float(loss), float(g[f"{tag}_loss"]), rtol=1e-4)

    def close(a, ref, rtol=2e-3):
        np.testing.assert_allclose(a, ref, rtol=rtol, atol=1e-4 * np.abs(ref).max())

    close(adj_norm.grad.numpy(), g[f"{tag}_dadj_norm"])
    close(feat_syn.grad.numpy(), g[f"{tag}_dfeat"])
    for i, p in enumerate(ps):
        ref = g[f"{tag}_dpge{i}"]
        if np.abs(ref).max() > 1e-6:
            close(p.grad.numpy(), ref)
    # inner steps (gcond.py:63-72)
    th = [T(g[f"{tag}_theta{i}"]).requires_grad_(True) for i in range(6)]
    opt = O.Adam(th, 0.01)
    with torch.no_grad():
        adj_in = O.dense_gcn_norm(O.pge_forward(feat_syn.detach(), pge))
    np.testing.assert_allclose(O.pge_forward(feat_syn.detach(), pge).detach().numpy(), g[f"{tag}_adj_inner"], rtol=1e-5, atol=1e-6)
    for it in range(2):
        out = O.model_forward("SGC_BN", feat_syn.detach(), adj_in, th, 2)
        li = torch.nn.functional.nll_loss(out, labels_syn)
        np.testing.assert_allclose(float(li), float(g[f"{tag}_inner_losses"][it]), rtol=1e-5)
        opt.step(torch.autograd.grad(li, th))
    for i, q in enumerate(th):
        np.testing.assert_allclose(q.detach().numpy(), g[f"{tag}_theta_after{i}"], rtol=1e-4, atol=2e-5)


def replay_masks(g, tag, nclass, p_drop, dtype=torch.float32):
    """oracle DROPOUT hook that replays the multipliers the reference drew (golden G26: one per class on the real batch, one per class on
    the synthetic graph — in train_class's call order — then one per inner step)."""
    def hook(where, x):
        kind, index, _rows = where
        i = {"real": 2 * index, "syn": 2 * index + 1, "inner": 2 * nclass + index}[kind]
        m = torch.from_numpy(g[f"{tag}_mask{i}"].astype(np.float32)).to(dtype) / (1.0 - p_drop)
        assert m.shape == x.shape, (where[:2], m.shape, x.shape)
        return m
    return hook


def test_g26_dropout_train_class_and_inner_steps(golden_dir):
    """Dropout in the condensation model (models/sgc.py:42; training mode inside train_class and the inner steps, gcond.py:45): the
    reference's train_class + backward and two inner steps on an SGC with two transforms and dropout 0.4, with the masks torch drew
    recorded in the fixture and replayed here — pins WHERE the restatement applies dropout (behind relu, before the second transform),
    WHEN it draws (per class: real batch then synthetic graph; once per inner step) and the 1 / (1 - p) scaling."""
    g, tag = load(golden_dir, "g26_dropout"), "dr"
    p_drop = float(g[f"{tag}_dropout"])
    batches, (n, nclass, d, hidden, t) = rebuild_batches(g, tag)
    theta = [T(g[f"{tag}_theta{i}"]).requires_grad_(True) for i in range(4)]
    ps = [T(g[f"{tag}_pge{i}"]).requires_grad_(True) for i in range(10)]
    pge = {"lin": ps[:6], "bn": ps[6:]}
    feat_syn = T(g[f"{tag}_feat_syn"]).requires_grad_(True)
    labels_syn = T(g[f"{tag}_labels_syn"])
    _, ncd, _ = O.generate_labels_syn(g[f"{tag}_y"][g[f"{tag}_idx_train"]], float(g[f"{tag}_rate"]))
    adj_norm = O.dense_gcn_norm(O.pge_forward(feat_syn, pge))
    adj_norm.retain_grad()
    O.DROPOUT = replay_masks(g, tag, nclass, p_drop)
    try:
        loss = O.train_class("SGC", theta, batches, T(g[f"{tag}_x"]), T(g[f"{tag}_y"]), feat_syn, adj_norm, labels_syn, ncd, nclass, 2,
                             "ours")
        loss.backward()
        np.testing.assert_allclose(float(loss), float(g[f"{tag}_loss"]), rtol=1e-4)

        def close(a, ref, rtol=2e-3):
            np.testing.assert_allclose(a, ref, rtol=rtol, atol=1e-4 * np.abs(ref).max())

        close(adj_norm.grad.numpy(), g[f"{tag}_dadj_norm"])
        close(feat_syn.grad.numpy(), g[f"{tag}_dfeat"])
        for i, p in enumerate(ps):
            ref = g[f"{tag}_dpge{i}"]
            if np.abs(ref).max() > 1e-6:
                close(p.grad.numpy(), ref)
        th = [T(g[f"{tag}_theta{i}"]).requires_grad_(True) for i in range(4)]
        opt = O.Adam(th, 0.01)
        with torch.no_grad():
            adj_in = O.dense_gcn_norm(O.pge_forward(feat_syn.detach(), pge))
        for it in range(2):
            out = O.model_forward("SGC", feat_syn.detach(), adj_in, th, 2, ("inner", it, None))
            li = torch.nn.functional.nll_loss(out, labels_syn)
            np.testing.assert_allclose(float(li), float(g[f"{tag}_inner_losses"][it]), rtol=1e-5)
            opt.step(torch.autograd.grad(li, th))
        for i, q in enumerate(th):
            np.testing.assert_allclose(q.detach().numpy(), g[f"{tag}_theta_after{i}"], rtol=1e-4, atol=2e-5)
    finally:
        O.DROPOUT = None
    # without the masks the restatement lands somewhere else: the fixture gates the dropout path, not the dropout-free one
    loss0 = O.train_class("SGC", [q.detach().requires_grad_(True) for q in theta], batches, T(g[f"{tag}_x"]), T(g[f"{tag}_y"]),
                          feat_syn, O.dense_gcn_norm(O.pge_forward(feat_syn, pge)), labels_syn, ncd, nclass, 2, "ours")
    assert abs(float(loss0) - float(g[f"{tag}_loss"])) > 1e-2 * abs(float(g[f"{tag}_loss"]))


def replay_masks_gcn(g, tag, nclass, hid_layers, p_drop, dtype=torch.float32):
    """oracle DROPOUT hook that replays golden G29's recorded multipliers: per class the real batch's `hid_layers` masks then the
    synthetic graph's (train_class's call order), then `hid_layers` per inner step."""
    def hook(where, x):
        kind, index, _rows, layer = where
        i = {"real": 2 * index * hid_layers, "syn": (2 * index + 1) * hid_layers, "inner": (2 * nclass + index) * hid_layers}[kind] + layer
        m = torch.from_numpy(g[f"{tag}_mask{i}"].astype(np.float32)).to(dtype) / (1.0 - p_drop)
        assert m.shape == x.shape, (where[:2], layer, m.shape, x.shape)
        return m
    return hook


@pytest.mark.parametrize("tag", ["gd2", "gd3"])
def test_g29_gcn_dropout_train_class_and_inner_steps(golden_dir, tag):
    """Dropout in the GCN condensation model (models/base.py:51-78): the reference's train_class + backward and two inner steps for two
    and three GraphConvolution layers with dropout 0.4, torch's masks recorded and replayed — pins where the restatement applies
    dropout (behind relu in every layer but the last, on the rows of that layer's output), the order of the draws and the scaling."""
    g = load(golden_dir, "g29_gcn_dropout")
    p_drop = float(g[f"{tag}_dropout"])
    nlayers = len(g[f"{tag}_sizes"])
    batches, (n, nclass, d, hidden, t) = rebuild_batches(g, tag)
    theta = [T(g[f"{tag}_theta{i}"]).requires_grad_(True) for i in range(2 * nlayers)]
    ps = [T(g[f"{tag}_pge{i}"]).requires_grad_(True) for i in range(10)]
    pge = {"lin": ps[:6], "bn": ps[6:]}
    feat_syn = T(g[f"{tag}_feat_syn"]).requires_grad_(True)
    labels_syn = T(g[f"{tag}_labels_syn"])
    _, ncd, _ = O.generate_labels_syn(g[f"{tag}_y"][g[f"{tag}_idx_train"]], float(g[f"{tag}_rate"]))
    adj_norm = O.dense_gcn_norm(O.pge_forward(feat_syn, pge))
    adj_norm.retain_grad()
    O.DROPOUT = replay_masks_gcn(g, tag, nclass, nlayers - 1, p_drop)
    try:
        loss = O.train_class("GCN", theta, batches, T(g[f"{tag}_x"]), T(g[f"{tag}_y"]), feat_syn, adj_norm, labels_syn, ncd, nclass,
                             nlayers, "ours")
        loss.backward()
        np.testing.assert_allclose(float(loss.detach()), float(g[f"{tag}_loss"]), rtol=1e-4)

        def close(a, ref, rtol=2e-3):
            np.testing.assert_allclose(a, ref, rtol=rtol, atol=1e-4 * np.abs(ref).max())

        close(adj_norm.grad.numpy(), g[f"{tag}_dadj_norm"])
        close(feat_syn.grad.numpy(), g[f"{tag}_dfeat"])
        for i, p in enumerate(ps):
            ref = g[f"{tag}_dpge{i}"]
            if np.abs(ref).max() > 1e-6:
                close(p.grad.numpy(), ref)
        th = [T(g[f"{tag}_theta{i}"]).requires_grad_(True) for i in range(2 * nlayers)]
        opt = O.Adam(th, 0.01)
        with torch.no_grad():
            adj_in = O.dense_gcn_norm(O.pge_forward(feat_syn.detach(), pge))
        for it in range(2):
            out = O.model_forward("GCN", feat_syn.detach(), adj_in, th, nlayers, ("inner", it, None))
            li = torch.nn.functional.nll_loss(out, labels_syn)
            np.testing.assert_allclose(float(li), float(g[f"{tag}_inner_losses"][it]), rtol=1e-5)
            opt.step(torch.autograd.grad(li, th))
        for i, q in enumerate(th):
            np.testing.assert_allclose(q.detach().numpy(), g[f"{tag}_theta_after{i}"], rtol=1e-4, atol=2e-5)
    finally:
        O.DROPOUT = None
    loss0 = O.train_class("GCN", [q.detach().requires_grad_(True) for q in theta], batches, T(g[f"{tag}_x"]), T(g[f"{tag}_y"]),
                          feat_syn, O.dense_gcn_norm(O.pge_forward(feat_syn, pge)), labels_syn, ncd, nclass, nlayers, "ours")
    assert abs(float(loss0) - float(g[f"{tag}_loss"])) > 1e-3 * abs(float(g[f"{tag}_loss"]))      # (ten times the gate above)


def test_g30_dropout_in_the_sibling_methods(golden_dir):
    """Golden G30 — WHEN the reference's sibling methods draw dropout masks (two transforms, dropout 0.4 against 0): MSGC never does
    (msgc.py:99 `basic_model.eval()` before every train_class, never back to training mode): no mask drawn, losses and X~ bit for bit
    those of the run without dropout — so graphslim_amd's MSGC accepts the value without effect; SGDD (sgdd.py:62 model.train()) draws
    2 x nclass masks per train_class and one per inner step, as the restatement does (sgdd_oracle.SGDDState: train_class + the inner
    loop's ("inner", i) hook) and the engine's dropout stage under the IGNR structure."""
    g = load(golden_dir, "g30_siblings_dropout")
    nclass, epochs, outer, inner = [int(v) for v in g["sgdd_meta"]]
    assert int(g["msgc_p0_masks_drawn"]) == 0 and int(g["msgc_p4_masks_drawn"]) == 0
    assert np.array_equal(g["msgc_p0_losses"], g["msgc_p4_losses"]) and np.array_equal(g["msgc_p0_feat_syn"], g["msgc_p4_feat_syn"])
    assert int(g["sgdd_p0_masks_drawn"]) == 0 and int(g["sgdd_p4_masks_drawn"]) == epochs * outer * (2 * nclass + inner)
    assert np.abs(g["sgdd_p0_losses"] - g["sgdd_p4_losses"]).min() > 1e-2
    import inspect
    import oracle.msgc_oracle as M
    import oracle.sgdd_oracle as S
    assert "DROPOUT" not in inspect.getsource(M.sgc_forward_batched)           # MSGC's restatement has no dropout stage
    assert '("inner", i, None)' in inspect.getsource(S.SGDDState.outer_iteration)


@pytest.mark.parametrize("tag", ["cora_like", "arxiv_like"])
def test_g8_full_reduce_and_evaluate(golden_dir, tag):
    """Replays a whole reference run (GCond.__init__ + reduce + intermediate evals + Evaluator.evaluate)."""
    g = load(golden_dir, "g8_reduce")
    n, nclass, d, hidden, t, inner, epochs, outer = [int(v) for v in g[f"{tag}_meta"]]
    a = sp.coo_matrix((np.ones(len(g[f"{tag}_a_row"])), (g[f"{tag}_a_row"], g[f"{tag}_a_col"])), shape=(n, n)).tocsr()
    csr = O.normalize_adj_sparse(a)
    idx_train, y = g[f"{tag}_idx_train"], g[f"{tag}_y"]
    labels_train = y[idx_train]

    def batch_fn(c, call):
        rng = np.random.default_rng(100000 + call)
        batch = rng.permutation(idx_train[labels_train == c])[:256]
        return O.sample_blocks(*csr, batch, [4, 3], rng)

    out = O.reduce_and_evaluate(a, g[f"{tag}_x"], y, idx_train, g[f"{tag}_idx_val"], g[f"{tag}_idx_test"], nclass,
                                reduction_rate=0.5, hidden=hidden, ntrans=t, pge_hid=256 if tag == "arxiv_like" else 128,
                                epochs=epochs, outer_loop=outer, inner_loop=inner, checkpoints=[5, 11], lr_feat=0.01,
                                lr_adj=0.01, lr_model=0.01, sizes=[4, 3], threshold=0.05, seed=1, run_inter_eval=2,
                                eval_epochs=20, run_eval=3, batch_fn=batch_fn)
    np.testing.assert_array_equal(out["saved"][2].numpy(), g[f"{tag}_saved_label"])
    np.testing.assert_allclose(np.array(out["loss_avg"]), g[f"{tag}_loss_avg"], rtol=1e-4)
    np.testing.assert_allclose(out["feat_syn"].numpy(), g[f"{tag}_feat_syn"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(out["adj_syn"].numpy(), g[f"{tag}_adj_syn"], rtol=1e-3, atol=2e-4)
    for p, i in zip(out["state"].pge_param_list(), range(10)):
        if i in (1, 3):
            # layers.{0,1}.bias feed a batch-stat BatchNorm: their true gradient is exactly 0, so Adam
            # random-walks them on rounding noise in the reference too; they cannot influence adj_syn.
            continue
        # Adam turns rounding-level gradient noise on near-dead units into O(lr) steps: allow 1e-3 abs
        # (each parameter moved by up to epochs*outer*lr = 0.24 over the run).
        np.testing.assert_allclose(p.detach().numpy(), g[f"{tag}_pge{i}"], rtol=1e-3, atol=1e-3)
    # eval-GCN accuracy gate: +-0.3 pp  (here the trajectories agree, so the accuracies coincide)
    assert abs(out["eval_mean"] - float(g[f"{tag}_eval_mean"])) <= 0.003
    log = str(g[f"{tag}_log"])
    vals = [float(l.split()[1]) for l in log.splitlines() if l.startswith("Val:") or l.startswith("\nVal:")]
    vals = [float(l.split("Val:")[1].split()[0]) for l in log.split("\n") if "Val:" in l]
    got = [100 * m[0] for m in out["inter"]]
    np.testing.assert_allclose(got, vals, atol=0.3)
    assert list(g[f"{tag}_files"]) == ["adj_cora_0.5_1.pt", "feat_cora_0.5_1.pt", "label_cora_0.5_1.pt"] or \
        list(g[f"{tag}_files"]) == ["adj_ogbn-arxiv_0.5_1.pt", "feat_ogbn-arxiv_0.5_1.pt", "label_ogbn-arxiv_0.5_1.pt"]


@pytest.mark.parametrize("fname", ["g9_fit_with_val", "g25_fit_with_val_three_layers"])
@pytest.mark.parametrize("setting", ["trans", "ind"])
def test_g9_g25_fit_with_val_trace(golden_dir, setting, fname):
    """models/base.py:80-225 replayed by the oracle from the reference's own initial weights: per-epoch validation
    accuracy, best-validation restore, final weights and test accuracy — the two-layer evaluation GCN (G9) and the three-layer one of
    `--nlayers 3` (G25, models/gcn.py:8-23)."""
    g = load(golden_dir, fname)
    nw = 2 * (int(g["nlayers"]) if "nlayers" in g else 2)
    n, nclass, d, hidden, ns, iters = [int(v) for v in g["meta"]]
    a = sp.coo_matrix((np.ones(len(g["a_row"])), (g["a_row"], g["a_col"])), shape=(n, n)).tocsr()
    x, y = T(g["x"]), T(g["y"])
    idx_val, idx_test = g["idx_val"], g["idx_test"]
    if setting == "trans":
        feat, adj = x, O.csr_to_torch(*O.normalize_adj_sparse(a), n)
        iv, it = T(idx_val), T(idx_test)
    else:
        feat = x[idx_val]
        adj = O.csr_to_torch(*O.normalize_adj_sparse(a[np.ix_(idx_val, idx_val)]), len(idx_val))
        iv, it = torch.arange(len(idx_val)), torch.arange(len(idx_test))
    init = [T(g[f"{setting}_init{i}"]) for i in range(nw)]
    trace = []
    best, w = O.fit_with_val(init, T(g["adj_syn"]), T(g["feat_syn"]), T(g["labels_syn"]), feat, adj, iv, y[idx_val], iters,
                             lr=0.01, weight_decay=5e-4, trace=trace)
    ref = g[f"{setting}_val_trace"]
    got = np.array([t[1] for t in trace])
    # accuracies are k / n_val: allow one validation node to flip on an fp32 tie
    assert np.abs(got - ref).max() <= 1.0 / len(idx_val) + 1e-12
    assert abs(best - float(g[f"{setting}_best"])) <= 1.0 / len(idx_val) + 1e-12
    if np.array_equal(got, ref):
        for i in range(nw):
            np.testing.assert_allclose(w[i].numpy(), g[f"{setting}_final{i}"], rtol=1e-3, atol=1e-5)
    if setting == "trans":
        acc = O.eval_test(w, x, adj, it, y[idx_test])
    else:
        adj_t = O.csr_to_torch(*O.normalize_adj_sparse(a[np.ix_(idx_test, idx_test)]), len(idx_test))
        acc = O.eval_test(w, x[idx_test], adj_t, it, y[idx_test])
    assert abs(acc - float(g[f"{setting}_test"])) <= 1.0 / len(idx_test) + 1e-12


def test_g11_chunked_pge(golden_dir):
    """parametrized_adj.py:41-55 (reddit, r >= 0.01): five np.array_split chunks with per-chunk BatchNorm statistics."""
    g = load(golden_dir, "g11_pge_chunked")
    x = T(g["x"]).clone().requires_grad_(True)
    lin = [T(g[f"p{i}"]).clone().requires_grad_(True) for i in range(6)]
    bn = [T(g[f"p{i}"]).clone().requires_grad_(True) for i in range(6, 10)]
    adj = O.pge_forward(x, {"lin": lin, "bn": bn}, n_chunks=5)
    np.testing.assert_allclose(adj.detach().numpy(), g["adj"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(g["inference"], g["adj"], rtol=0, atol=0)
    (adj * T(g["w"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=1e-3, atol=1e-5)
    for i, q in enumerate(lin + bn):
        if i in (1, 3):      # biases in front of a batch-stat BatchNorm: exact-zero gradient, rounding noise on both sides
            continue
        np.testing.assert_allclose(q.grad.numpy(), g[f"dp{i}"], rtol=2e-3, atol=2e-5, err_msg=f"p{i}")
    # and it is NOT the single-batch forward
    assert np.abs(O.pge_forward(x.detach(), {"lin": [q.detach() for q in lin], "bn": [q.detach() for q in bn]}).numpy() - g["adj"]).max() > 1e-3


@pytest.mark.parametrize("setting", ["trans", "ind"])
def test_g12_whole_graph_fit_with_val(golden_dir, setting):
    """models/base.py:80-225 with reduced=False (training on the original graph) replayed from the reference's initial
    weights: validation trace, best-validation restore, test accuracy."""
    g = load(golden_dir, "g12_fit_full")
    n, nclass, d, hidden, iters = [int(v) for v in g["meta"]]
    a = sp.coo_matrix((np.ones(len(g["a_row"])), (g["a_row"], g["a_col"])), shape=(n, n)).tocsr()
    x, y = T(g["x"]), T(g["y"])
    it, iv, ie = g["idx_train"], g["idx_val"], g["idx_test"]
    norm = lambda m: O.csr_to_torch(*O.normalize_adj_sparse(m), m.shape[0])
    init = [T(g[f"{setting}_init{i}"]) for i in range(4)]
    trace = []
    if setting == "trans":
        A = norm(a)
        best, w = O.fit_with_val_full(init, x, A, T(it), y[it], x, A, T(iv), y[iv], iters, lr=0.01, weight_decay=5e-4, trace=trace)
        acc = O.eval_test(w, x, A, T(ie), y[ie])
    else:
        best, w = O.fit_with_val_full(init, x[it], norm(a[np.ix_(it, it)]), None, y[it], x[iv], norm(a[np.ix_(iv, iv)]), None, y[iv],
                                      iters, lr=0.01, weight_decay=5e-4, trace=trace)
        acc = O.eval_test(w, x[ie], norm(a[np.ix_(ie, ie)]), torch.arange(len(ie)), y[ie])
    got, ref = np.array([t[1] for t in trace]), g[f"{setting}_val_trace"]
    assert np.abs(got - ref).max() <= 1.0 / len(iv) + 1e-12
    assert abs(best - float(g[f"{setting}_best"])) <= 1.0 / len(iv) + 1e-12
    assert abs(acc - float(g[f"{setting}_test"])) <= 1.0 / len(ie) + 1e-12
    if np.array_equal(got, ref):
        for i in range(4):
            np.testing.assert_allclose(w[i].numpy(), g[f"{setting}_final{i}"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("tag", ["doscond_t1", "doscond_t2", "gcondx_t1", "gcondx_t2", "doscondx_t1"])
def test_g13_sibling_methods_replay_the_reference(golden_dir, tag):
    """G13: the reference's own DosCond / GCondX / DosCondX .reduce (doscond.py:40-53, gcondx.py:44-69, doscondx.py:42-49), 24
    outer iterations each: pins oracle.GCondState.outer_iteration_ex (the update rules of the sibling methods) — per-iteration
    matching losses, final X~, final PGE parameters / inference adjacency."""
    g = load(golden_dir, "g13_siblings")
    n, nclass, d, hidden, t, inner, epochs, outer = [int(v) for v in g[f"{tag}_meta"]]
    a = sp.coo_matrix((np.ones(len(g[f"{tag}_a_row"])), (g[f"{tag}_a_row"], g[f"{tag}_a_col"])), shape=(n, n)).tocsr()
    csr = O.normalize_adj_sparse(a)
    idx_train, y = g[f"{tag}_idx_train"], g[f"{tag}_y"]
    labels_train = y[idx_train]

    def batch_fn(c, call):
        rng = np.random.default_rng(100000 + call)
        batch = rng.permutation(idx_train[labels_train == c])[:256]
        return O.sample_blocks(*csr, batch, [4, 3], rng)

    out = O.reduce_sibling(tag.split("_")[0], a, g[f"{tag}_x"], y, idx_train, nclass, reduction_rate=0.5, hidden=hidden, ntrans=t,
                           pge_hid=256 if t == 2 else 128, epochs=epochs, outer_loop=outer, inner_loop=inner, lr_feat=0.01,
                           lr_adj=0.01, lr_model=0.01, seed=1, batch_fn=batch_fn)
    np.testing.assert_array_equal(out["labels_syn"], g[f"{tag}_labels_syn"])
    np.testing.assert_allclose(np.array(out["losses"]), g[f"{tag}_losses"], rtol=2e-4)
    np.testing.assert_allclose(out["feat_syn"].numpy(), g[f"{tag}_feat_syn"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(out["adj_syn"].numpy(), g[f"{tag}_adj_syn"], rtol=1e-3, atol=2e-4)
    if tag.startswith("doscond_"):
        for p, i in zip(out["state"].pge_param_list(), range(10)):
            if i in (1, 3):
                continue     # zero-gradient biases in front of batch-statistics BatchNorm (see test_g8)
            # Adam turns rounding-level gradient noise on near-dead units into O(lr) steps (each parameter moved by up to
            # epochs * outer * lr = 0.24): all but a vanishing share of the entries within 1e-3, none further than 0.02
            diff = np.abs(p.detach().numpy() - g[f"{tag}_pge{i}"])
            tol = 1e-3 + 1e-3 * np.abs(g[f"{tag}_pge{i}"])
            assert (diff > tol).mean() <= 2e-4 and diff.max() <= 0.02, (i, (diff > tol).mean(), diff.max())


@pytest.mark.parametrize("tag", ["t1", "t2"])
def test_g18_msgc_replays_the_reference(golden_dir, tag):
    """G18: the reference's own MSGC.__init__ + reduce (msgc.py:18-130,178-236; 3 epochs x 4 outer iterations, batch_adj 3 / 2):
    pins oracle/msgc_oracle.py — label allocation, the greedy edge lists incl. their randperm tie breaks (bit exact), the
    per-iteration matching losses on the batched structure, final X~, edge-MLP parameters, the normalised [B, n', n'] batch and
    the window averages a checkpoint hands to the evaluation."""
    from oracle import msgc_oracle as M
    g = load(golden_dir, "g18_msgc")
    n, nclass, d, hidden, t, inner, epochs, outer, B = [int(v) for v in g[f"{tag}_meta"]]
    a = sp.coo_matrix((np.ones(len(g[f"{tag}_a_row"])), (g[f"{tag}_a_row"], g[f"{tag}_a_col"])), shape=(n, n)).tocsr()
    csr = O.normalize_adj_sparse(a)
    idx_train, y = g[f"{tag}_idx_train"], g[f"{tag}_y"]
    labels_train = y[idx_train]

    def batch_fn(c, call):
        rng = np.random.default_rng(100000 + call)
        batch = rng.permutation(idx_train[labels_train == c])[:256]
        return O.sample_blocks(*csr, batch, [4, 3], rng)

    out = M.reduce_msgc(a, g[f"{tag}_x"], y, idx_train, nclass, reduction_rate=0.5, hidden=hidden, ntrans=t, epochs=epochs,
                        outer_loop=outer, inner_loop=inner, batch_adj=B, lr_feat=0.01, lr_adj=0.01, lr_model=0.01, seed=1,
                        batch_fn=batch_fn)
    np.testing.assert_array_equal(out["y_syn"].numpy(), g[f"{tag}_y_syn"])
    for k in ("rows", "cols", "batch"):
        np.testing.assert_array_equal(out[k].numpy(), g[f"{tag}_{k}"])
    np.testing.assert_allclose(np.array(out["losses"]), g[f"{tag}_losses"], rtol=2e-4)
    np.testing.assert_allclose(out["feat_syn"].numpy(), g[f"{tag}_feat_syn"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(out["adj"].numpy(), g[f"{tag}_adj"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(out["best_x"].numpy(), g[f"{tag}_best_x"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(out["best_adj"].numpy(), g[f"{tag}_best_adj"], rtol=1e-3, atol=2e-4)
    np.testing.assert_array_equal(np.tile(out["y_syn"].numpy(), B), g[f"{tag}_labels_out"])
    for i, p in enumerate(out["params"]):
        if i in (1, 5):
            continue         # biases in front of a batch-statistics BatchNorm: zero gradient up to rounding (see test_g8)
        diff = np.abs(p.numpy() - g[f"{tag}_pge{i}"])
        tol = 1e-3 + 1e-3 * np.abs(g[f"{tag}_pge{i}"])
        assert (diff > tol).mean() <= 2e-4 and diff.max() <= 0.02, (i, (diff > tol).mean(), diff.max())


@pytest.mark.parametrize("tag", ["t1", "t2"])
def test_g20_sgdd_replays_the_reference(golden_dir, tag):
    """G20: the reference's own SGDD.__init__ + reduce (sgdd.py:17-105, models/ignr.py:95-213; 12 epochs x 2 iterations = 10
    structure epochs + 2 feature epochs): pins oracle/sgdd_oracle.py — the IGNR draw order, both BatchNorm modes with their
    running statistics, the optimal-transport regulariser value of every iteration, matching losses, final X~ / parameters /
    inference structure."""
    from oracle import sgdd_oracle as S
    g = load(golden_dir, "g20_sgdd")
    n, nclass, d, hidden, t, inner, epochs, outer, mx = [int(v) for v in g[f"{tag}_meta"]]
    a = sp.coo_matrix((np.ones(len(g[f"{tag}_a_row"])), (g[f"{tag}_a_row"], g[f"{tag}_a_col"])), shape=(n, n)).tocsr()
    csr = O.normalize_adj_sparse(a)
    idx_train, y = g[f"{tag}_idx_train"], g[f"{tag}_y"]
    labels_train = y[idx_train]

    def batch_fn(c, call):
        rng = np.random.default_rng(100000 + call)
        batch = rng.permutation(idx_train[labels_train == c])[:256]
        return O.sample_blocks(*csr, batch, [4, 3], rng)

    out = S.reduce_sgdd(a, g[f"{tag}_x"], y, idx_train, nclass, reduction_rate=0.5, hidden=hidden, ntrans=t, epochs=epochs,
                        outer_loop=outer, inner_loop=inner, mx_size=mx, opt_scale=float(g[f"{tag}_opt_scale"]), ep_ratio=0.5,
                        lr_feat=0.01, lr_adj=0.01, lr_model=0.01, seed=1, batch_fn=batch_fn)
    np.testing.assert_array_equal(out["labels_syn"], g[f"{tag}_labels_syn"])
    np.testing.assert_allclose(out["state"].Lx_inv.numpy(), g[f"{tag}_Lx_inv"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(np.array(out["losses"]), g[f"{tag}_losses"], rtol=2e-3)
    np.testing.assert_allclose(np.array(out["regs"]), g[f"{tag}_regs"], rtol=2e-2)
    # four Adam steps of lr 0.01 on X~: an entry whose gradient sits at the rounding level may take a step of the other sign
    dx = np.abs(out["feat_syn"].numpy() - g[f"{tag}_feat_syn"])
    assert (dx > 5e-4 + 1e-3 * np.abs(g[f"{tag}_feat_syn"])).mean() <= 0.02 and dx.max() <= 0.02, ((dx > 5e-4).mean(), dx.max())
    np.testing.assert_allclose(out["adj"].numpy(), g[f"{tag}_adj"], atol=5e-3)
    for i, key in enumerate(["c0", "c1", "x0", "x1"]):
        # the running mean contains the layer's bias, whose exact gradient is zero in front of a batch-statistics BatchNorm: Adam turns
        # its rounding-level gradient into +-lr steps on both sides (24 iterations x lr 0.01), independently
        drm = np.abs(out["running"][key][0].numpy() - g[f"{tag}_rm{i}"])
        assert (drm > 3e-2).mean() <= 0.03 and drm.max() <= 0.1, (key, (drm > 3e-2).mean(), drm.max())
        rv = g[f"{tag}_rv{i}"]
        drv = np.abs(out["running"][key][1].numpy() - rv) / (np.abs(rv) + 0.05)
        assert (drv > 3e-2).mean() <= 0.03 and drv.max() <= 0.2, (key, (drv > 3e-2).mean(), drv.max())
