"""Checks the closed-form second-order math (oracle/analytic.py — the spec the HIP kernels follow)
against the autograd oracle in fp64.  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.nn.functional as F

from oracle import analytic as A
from oracle import gcond_oracle as O

DT = torch.float64


def tiny_problem(T, seed=0, n=70, nclass=4, d=6, hidden=9, rate=0.4, sizes=(3, 2)):
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    y = rng.integers(0, nclass, n)
    src, dst = rng.integers(0, n, 4 * n), rng.integers(0, n, 4 * n)
    a = sp.coo_matrix((np.ones(4 * n), (src, dst)), shape=(n, n)).tocsr()
    a = ((a + a.T) > 0).astype(np.int64).tolil()
    a.setdiag(0)
    a = a.tocsr()
    x = torch.randn(n, d, dtype=DT)
    idx_train = np.sort(rng.permutation(n)[: n // 2])
    labels_train = y[idx_train]
    labels_syn, ncd, ranges = O.generate_labels_syn(labels_train, rate)
    csr = O.normalize_adj_sparse(a)
    batches = []
    for c in range(nclass):
        seeds = rng.permutation(idx_train[labels_train == c])[:8]
        batches.append(O.sample_blocks(*csr, seeds, list(sizes), rng))
    theta = [p.to(DT) for p in O.init_sgc(d, hidden, nclass, T)]
    ns = len(labels_syn)
    xs = torch.randn(ns, d, dtype=DT)
    pge = O.init_pge(d, 8, dtype=DT)
    pge["bn"] = [torch.rand(8, dtype=DT) + 0.5, torch.randn(8, dtype=DT) * 0.2, torch.rand(8, dtype=DT) + 0.5,
                 torch.randn(8, dtype=DT) * 0.2]
    return dict(x=x, y=torch.from_numpy(y), batches=batches, theta=theta, xs=xs, pge=pge, nclass=nclass,
                labels_syn=torch.from_numpy(labels_syn), ncd=ncd, ranges=ranges)


def dense_batch(b):
    bs, n_id, adjs = b
    return bs, torch.from_numpy(n_id), [O.block_to_torch(a, DT).to_dense() for a in adjs]


@pytest.mark.parametrize("T", [1, 2])
@pytest.mark.parametrize("sizes", [(3, 2), (3, 2, 2), (2, 2, 2, 2)])
def test_real_class_grad(T, sizes):
    p = tiny_problem(T, sizes=sizes)
    for c in range(p["nclass"]):
        theta = [t.clone().requires_grad_(True) for t in p["theta"]]
        ref, _ = O.class_real_grads("SGC", theta, p["x"], p["y"], p["batches"][c], len(sizes))
        got = A.real_class_grad(p["theta"], p["x"], dense_batch(p["batches"][c]), c)
        for r, g in zip(ref, got):
            np.testing.assert_allclose(g.numpy(), r.numpy(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("T", [1, 2])
@pytest.mark.parametrize("metric,K", [("ours", 2), ("mse", 2), ("cos", 2), ("ours", 3), ("mse", 3), ("ours", 4), ("ours", 1)])
def test_syn_match_second_order(T, metric, K):
    """K = nlayers propagations (models/sgc.py:45-51): only Q = A^^K and the product rule for d A^ change (oracle/analytic.py)."""
    p = tiny_problem(T, seed=3, sizes=(3, 2) if K <= 2 else (3, 2) + (2,) * (K - 2))
    theta = [t.clone().requires_grad_(True) for t in p["theta"]]
    xs = p["xs"].clone().requires_grad_(True)
    adj = O.pge_forward(xs.detach(), p["pge"])
    an = O.dense_gcn_norm(adj).clone().requires_grad_(True)
    if K == 1:      # (no reference run exists for one hop — PyG hands train_class a bare Adj tuple —; the math is checked all the same)
        p["batches"] = [(b[0], b[1][:b[2][-1][3][1]], b[2][-1:]) for b in p["batches"]]
    loss = O.train_class("SGC", theta, p["batches"], p["x"], p["y"], xs, an, p["labels_syn"], p["ncd"], p["nclass"],
                         K, metric)
    gx, ga = torch.autograd.grad(loss, [xs, an])
    if K == 1:
        gw_real = [[g.detach() for g in O.class_real_grads("SGC", [t.clone().requires_grad_(True) for t in p["theta"]], p["x"], p["y"],
                                                           p["batches"][c], 1)[0]] for c in range(p["nclass"])]
    else:
        gw_real = [A.real_class_grad(p["theta"], p["x"], dense_batch(p["batches"][c]), c) for c in range(p["nclass"])]
    ns = len(p["labels_syn"])
    ranges = [tuple(p["ranges"][c]) for c in range(p["nclass"])]
    coeffs = [p["ncd"][c] / ns for c in range(p["nclass"])]
    l, dX, dA = A.syn_match(p["theta"], p["xs"], an.detach(), p["labels_syn"], ranges, coeffs, gw_real, metric, K=K)
    np.testing.assert_allclose(float(l), float(loss), rtol=1e-10)
    np.testing.assert_allclose(dX.numpy(), gx.numpy(), rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(dA.numpy(), ga.numpy(), rtol=1e-7, atol=1e-11)


@pytest.mark.parametrize("T", [2, 3, 4])
@pytest.mark.parametrize("metric,K,sizes", [("ours", 2, (3, 2)), ("mse", 2, (3, 2)), ("cos", 2, (3, 2)), ("ours", 3, (3, 2, 2))])
def test_deep_closed_forms(T, metric, K, sizes):
    """`--ntrans` >= 3 (models/sgc.py:32-35): the closed forms as loops over the layers (oracle/analytic.py *_deep; at T = 2 they must
    reproduce the two-transform forms) — real-side class gradients, the matching loss with its second-order gradients wrt X~ and A^,
    and the inner steps' gradient — against autograd through the oracle, fp64."""
    p = tiny_problem(T, seed=7, sizes=sizes)
    gw_real = []
    for c in range(p["nclass"]):
        theta = [t.clone().requires_grad_(True) for t in p["theta"]]
        ref, _ = O.class_real_grads("SGC", theta, p["x"], p["y"], p["batches"][c], K)
        got = A.real_class_grad_deep(p["theta"], p["x"], dense_batch(p["batches"][c]), c)
        for r, g in zip(ref, got):
            np.testing.assert_allclose(g.numpy(), r.numpy(), rtol=1e-9, atol=1e-12)
        gw_real.append(got)
    theta = [t.clone().requires_grad_(True) for t in p["theta"]]
    xs = p["xs"].clone().requires_grad_(True)
    an = O.dense_gcn_norm(O.pge_forward(xs.detach(), p["pge"])).clone().requires_grad_(True)
    loss = O.train_class("SGC", theta, p["batches"], p["x"], p["y"], xs, an, p["labels_syn"], p["ncd"], p["nclass"], K, metric)
    gx, ga = torch.autograd.grad(loss, [xs, an])
    ns = len(p["labels_syn"])
    ranges = [tuple(p["ranges"][c]) for c in range(p["nclass"])]
    coeffs = [p["ncd"][c] / ns for c in range(p["nclass"])]
    l, dX, dA = A.syn_match_deep(p["theta"], p["xs"], an.detach(), p["labels_syn"], ranges, coeffs, gw_real, metric, K=K)
    np.testing.assert_allclose(float(l), float(loss), rtol=1e-10)
    np.testing.assert_allclose(dX.numpy(), gx.numpy(), rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(dA.numpy(), ga.numpy(), rtol=1e-7, atol=1e-11)
    if T == 2:
        l2, dX2, dA2 = A.syn_match(p["theta"], p["xs"], an.detach(), p["labels_syn"], ranges, coeffs, gw_real, metric, K=K)
        np.testing.assert_allclose(dX.numpy(), dX2.numpy(), rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(dA.numpy(), dA2.numpy(), rtol=1e-10, atol=1e-13)
    # the inner steps' gradient
    theta = [t.clone().requires_grad_(True) for t in p["theta"]]
    out = O.sgc_forward(p["xs"], an.detach(), theta, K)
    ref = torch.autograd.grad(F.nll_loss(out, p["labels_syn"]), theta)
    for r, g in zip(ref, A.sgc_dense_grad_deep(p["theta"], p["xs"], an.detach(), p["labels_syn"], K)):
        np.testing.assert_allclose(g.numpy(), r.numpy(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("L,sizes", [(2, (3, 2)), (3, (3, 2, 2)), (4, (2, 2, 2, 2))])
@pytest.mark.parametrize("metric", ["ours", "mse", "cos"])
def test_gcn_deep_closed_forms(L, sizes, metric):
    """K-layer GCN as condensation model (models/gcn.py:8-23, models/base.py:51-78): the closed forms as loops over the layers
    (oracle/analytic.py *_gcn_deep; at L = 2 they must reproduce the two-layer forms) against autograd through the oracle, fp64."""
    p = tiny_problem(1, seed=9, sizes=sizes)
    d, hidden = p["x"].shape[1], 9
    torch.manual_seed(4)
    th0 = [t.to(DT) for t in O.init_gcn(d, hidden, p["nclass"], L)]
    gw_real = []
    for c in range(p["nclass"]):
        theta = [t.clone().requires_grad_(True) for t in th0]
        ref, _ = O.class_real_grads("GCN", theta, p["x"], p["y"], p["batches"][c], L)
        got = A.real_class_grad_gcn_deep(th0, p["x"], dense_batch(p["batches"][c]), c)
        for r, g in zip(ref, got):
            np.testing.assert_allclose(g.numpy(), r.numpy(), rtol=1e-9, atol=1e-12)
        gw_real.append(got)
    theta = [t.clone().requires_grad_(True) for t in th0]
    xs = p["xs"].clone().requires_grad_(True)
    an = O.dense_gcn_norm(O.pge_forward(xs.detach(), p["pge"])).clone().requires_grad_(True)
    loss = O.train_class("GCN", theta, p["batches"], p["x"], p["y"], xs, an, p["labels_syn"], p["ncd"], p["nclass"], L, metric)
    gx, ga = torch.autograd.grad(loss, [xs, an])
    ns = len(p["labels_syn"])
    ranges = [tuple(p["ranges"][c]) for c in range(p["nclass"])]
    coeffs = [p["ncd"][c] / ns for c in range(p["nclass"])]
    l, dX, dA = A.syn_match_gcn_deep(th0, p["xs"], an.detach(), p["labels_syn"], ranges, coeffs, gw_real, metric)
    np.testing.assert_allclose(float(l), float(loss.detach()), rtol=1e-10)
    np.testing.assert_allclose(dX.numpy(), gx.numpy(), rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(dA.numpy(), ga.numpy(), rtol=1e-7, atol=1e-11)
    if L == 2:
        l2, dX2, dA2 = A.syn_match_gcn(th0, p["xs"], an.detach(), p["labels_syn"], ranges, coeffs, gw_real, metric)
        np.testing.assert_allclose(dX.numpy(), dX2.numpy(), rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(dA.numpy(), dA2.numpy(), rtol=1e-10, atol=1e-13)
    theta = [t.clone().requires_grad_(True) for t in th0]
    out = O.gcn_forward(p["xs"], an.detach(), theta)
    ref = torch.autograd.grad(F.nll_loss(out, p["labels_syn"]), theta)
    for r, g in zip(ref, A.gcn_dense_grad_deep(th0, p["xs"], an.detach(), p["labels_syn"])):
        np.testing.assert_allclose(g.numpy(), r.numpy(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("L,sizes,metric", [(2, (3, 2), "ours"), (2, (3, 2), "mse"), (3, (3, 2, 2), "ours"), (4, (2, 2, 2, 2), "cos")])
def test_gcn_dropout_closed_forms(L, sizes, metric):
    """GCN condensation model with dropout (models/base.py:59: F.dropout behind relu in every layer but the last; one forward per class in
    train_class, so one multiplier per class, forward and hidden layer): the closed forms of oracle/analytic.py *_gcn_deep with `mults`
    — per-class forwards, per-class adjoints through them — against autograd through the oracle with the same multipliers, fp64."""
    p = tiny_problem(1, seed=9, sizes=sizes)
    d, hidden, pd = p["x"].shape[1], 9, 0.35
    ns = len(p["labels_syn"])
    torch.manual_seed(4)
    th0 = [t.to(DT) for t in O.init_gcn(d, hidden, p["nclass"], L)]
    g = torch.Generator().manual_seed(13)
    draw = lambda rows: (torch.rand(rows, hidden, generator=g, dtype=DT) > pd).to(DT) / (1 - pd)
    # rows of layer k's output on the real batch: the targets of block k (dense_batch: [targets, sources])
    real = [[draw(blk.shape[0]) for blk in dense_batch(p["batches"][c])[2][:-1]] for c in range(p["nclass"])]
    syn = [[draw(ns) for _ in range(L - 1)] for c in range(p["nclass"])]
    inner = [draw(ns) for _ in range(L - 1)]
    O.DROPOUT = lambda where, x: {"real": real, "syn": syn}[where[0]][where[1]][where[3]] if where[0] != "inner" else inner[where[3]]
    try:
        theta = [t.clone().requires_grad_(True) for t in th0]
        xs = p["xs"].clone().requires_grad_(True)
        an = O.dense_gcn_norm(O.pge_forward(xs.detach(), p["pge"])).clone().requires_grad_(True)
        loss = O.train_class("GCN", theta, p["batches"], p["x"], p["y"], xs, an, p["labels_syn"], p["ncd"], p["nclass"], L, metric)
        gx, ga = torch.autograd.grad(loss, [xs, an])
        ref_real = [O.class_real_grads("GCN", [t.clone().requires_grad_(True) for t in th0], p["x"], p["y"], p["batches"][c], L, c)[0]
                    for c in range(p["nclass"])]
        theta = [t.clone().requires_grad_(True) for t in th0]
        out = O.gcn_forward(p["xs"], an.detach(), theta, ("inner", 0, None))
        ref_inner = torch.autograd.grad(F.nll_loss(out, p["labels_syn"]), theta)
    finally:
        O.DROPOUT = None
    gw_real = [A.real_class_grad_gcn_deep(th0, p["x"], dense_batch(p["batches"][c]), c, mults=real[c]) for c in range(p["nclass"])]
    for ref, got in zip(ref_real, gw_real):
        for r, q in zip(ref, got):
            np.testing.assert_allclose(q.numpy(), r.numpy(), rtol=1e-9, atol=1e-12)
    ranges = [tuple(p["ranges"][c]) for c in range(p["nclass"])]
    coeffs = [p["ncd"][c] / ns for c in range(p["nclass"])]
    l, dX, dA = A.syn_match_gcn_deep(th0, p["xs"], an.detach(), p["labels_syn"], ranges, coeffs, gw_real, metric, mults=syn)
    np.testing.assert_allclose(float(l), float(loss), rtol=1e-10)
    np.testing.assert_allclose(dX.numpy(), gx.numpy(), rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(dA.numpy(), ga.numpy(), rtol=1e-7, atol=1e-11)
    l0, _, _ = A.syn_match_gcn_deep(th0, p["xs"], an.detach(), p["labels_syn"], ranges, coeffs, gw_real, metric)
    assert abs(float(l0) - float(l)) > 1e-6 * abs(float(l))
    for r, q in zip(ref_inner, A.gcn_dense_grad_deep(th0, p["xs"], an.detach(), p["labels_syn"], mults=inner)):
        np.testing.assert_allclose(q.numpy(), r.numpy(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("metric,K", [("ours", 2), ("mse", 2), ("cos", 3)])
def test_dropout_closed_forms(metric, K):
    """Dropout in the condensation model (models/sgc.py:42): the closed forms with one multiplier per class forward — real batch and
    synthetic graph — against autograd through the oracle with the same multipliers (oracle.gcond_oracle.DROPOUT)."""
    p = tiny_problem(2, seed=5, sizes=(3, 2) if K == 2 else (3, 2, 2))
    ns, h, pd = len(p["labels_syn"]), p["theta"][0].shape[1], 0.4
    g = torch.Generator().manual_seed(11)
    real = [(torch.rand(len(p["batches"][c][1]), h, generator=g, dtype=DT) > pd).to(DT) / (1 - pd) for c in range(p["nclass"])]
    syn = [(torch.rand(ns, h, generator=g, dtype=DT) > pd).to(DT) / (1 - pd) for c in range(p["nclass"])]
    O.DROPOUT = lambda where, x: {"real": real, "syn": syn}[where[0]][where[1]]
    try:
        theta = [t.clone().requires_grad_(True) for t in p["theta"]]
        xs = p["xs"].clone().requires_grad_(True)
        an = O.dense_gcn_norm(O.pge_forward(xs.detach(), p["pge"])).clone().requires_grad_(True)
        loss = O.train_class("SGC", theta, p["batches"], p["x"], p["y"], xs, an, p["labels_syn"], p["ncd"], p["nclass"], K, metric)
        gx, ga = torch.autograd.grad(loss, [xs, an])
        ref_real = [O.class_real_grads("SGC", [t.clone().requires_grad_(True) for t in p["theta"]], p["x"], p["y"], p["batches"][c], K, c)[0]
                    for c in range(p["nclass"])]
    finally:
        O.DROPOUT = None
    gw_real = [A.real_class_grad(p["theta"], p["x"], dense_batch(p["batches"][c]), c, mult=real[c]) for c in range(p["nclass"])]
    for ref, got in zip(ref_real, gw_real):
        for r, q in zip(ref, got):
            np.testing.assert_allclose(q.numpy(), r.numpy(), rtol=1e-9, atol=1e-12)
    ranges = [tuple(p["ranges"][c]) for c in range(p["nclass"])]
    coeffs = [p["ncd"][c] / ns for c in range(p["nclass"])]
    l, dX, dA = A.syn_match(p["theta"], p["xs"], an.detach(), p["labels_syn"], ranges, coeffs, gw_real, metric, K=K, mults=syn)
    np.testing.assert_allclose(float(l), float(loss), rtol=1e-10)
    np.testing.assert_allclose(dX.numpy(), gx.numpy(), rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(dA.numpy(), ga.numpy(), rtol=1e-7, atol=1e-11)
    # without the multipliers the closed form is the dropout-free one (a different loss)
    l0, _, _ = A.syn_match(p["theta"], p["xs"], an.detach(), p["labels_syn"], ranges, coeffs, gw_real, metric, K=K)
    assert abs(float(l0) - float(l)) > 1e-6 * abs(float(l))


def test_gcn_norm_backward():
    torch.manual_seed(0)
    adj = torch.rand(9, 9, dtype=DT)
    adj = ((adj + adj.T) / 2).fill_diagonal_(0).requires_grad_(True)
    w = torch.randn(9, 9, dtype=DT)
    (O.dense_gcn_norm(adj) * w).sum().backward()
    an, ctx = A.gcn_norm_fwd(adj.detach())
    np.testing.assert_allclose(an.numpy(), O.dense_gcn_norm(adj.detach()).numpy(), rtol=1e-12)
    np.testing.assert_allclose(A.gcn_norm_bwd(w, ctx).numpy(), adj.grad.numpy(), rtol=1e-9, atol=1e-12)


def test_pge_forward_backward():
    p = tiny_problem(1, seed=5)
    xs = p["xs"].clone().requires_grad_(True)
    lin = [t.clone().requires_grad_(True) for t in p["pge"]["lin"]]
    bn = [t.clone().requires_grad_(True) for t in p["pge"]["bn"]]
    adj = O.pge_forward(xs, {"lin": lin, "bn": bn})
    w = torch.randn_like(adj)
    (adj * w).sum().backward()
    a2, ctx = A.pge_fwd(p["xs"], p["pge"]["lin"], p["pge"]["bn"])
    np.testing.assert_allclose(a2.numpy(), adj.detach().numpy(), rtol=1e-10, atol=1e-13)
    dX, dlin, dbn = A.pge_bwd(w, p["xs"], p["pge"]["lin"], p["pge"]["bn"], ctx)
    np.testing.assert_allclose(dX.numpy(), xs.grad.numpy(), rtol=1e-7, atol=1e-12)
    for got, ref in zip(dlin + dbn, lin + bn):
        np.testing.assert_allclose(got.numpy(), ref.grad.numpy(), rtol=1e-7, atol=1e-11)


@pytest.mark.parametrize("T", [1, 2])
def test_inner_model_grad(T):
    p = tiny_problem(T, seed=7)
    theta = [t.clone().requires_grad_(True) for t in p["theta"]]
    an = O.dense_gcn_norm(O.pge_forward(p["xs"], p["pge"]))
    loss = F.nll_loss(O.sgc_forward(p["xs"], an, theta), p["labels_syn"])
    ref = torch.autograd.grad(loss, theta)
    got = A.sgc_dense_grad(p["theta"], p["xs"], an, p["labels_syn"])
    for r, g in zip(ref, got):
        np.testing.assert_allclose(g.numpy(), r.numpy(), rtol=1e-9, atol=1e-13)


def test_gcn_condense_model_real_and_inner_grads():
    p = tiny_problem(2, seed=9)
    theta = [t.to(DT) for t in O.init_gcn(6, 9, p["nclass"], 2)]
    for c in range(p["nclass"]):
        th = [t.clone().requires_grad_(True) for t in theta]
        ref, _ = O.class_real_grads("GCN", th, p["x"], p["y"], p["batches"][c])
        got = A.real_class_grad_gcn(theta, p["x"], dense_batch(p["batches"][c]), c)
        for r, g in zip(ref, got):
            np.testing.assert_allclose(g.numpy(), r.numpy(), rtol=1e-9, atol=1e-12)
    th = [t.clone().requires_grad_(True) for t in theta]
    an = O.dense_gcn_norm(O.pge_forward(p["xs"], p["pge"]))
    loss = F.nll_loss(O.gcn_forward(p["xs"], an, th), p["labels_syn"])
    for r, g in zip(torch.autograd.grad(loss, th), A.gcn_dense_grad(theta, p["xs"], an, p["labels_syn"])):
        np.testing.assert_allclose(g.numpy(), r.numpy(), rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("metric", ["ours", "mse", "cos"])
def test_gcn_condense_model_second_order(metric):
    p = tiny_problem(2, seed=13)
    theta0 = [t.to(DT) for t in O.init_gcn(6, 9, p["nclass"], 2)]
    theta = [t.clone().requires_grad_(True) for t in theta0]
    xs = p["xs"].clone().requires_grad_(True)
    an = O.dense_gcn_norm(O.pge_forward(xs.detach(), p["pge"])).clone().requires_grad_(True)
    loss = O.train_class("GCN", theta, p["batches"], p["x"], p["y"], xs, an, p["labels_syn"], p["ncd"], p["nclass"], 2, metric)
    gx, ga = torch.autograd.grad(loss, [xs, an])
    gw_real = [A.real_class_grad_gcn(theta0, p["x"], dense_batch(p["batches"][c]), c) for c in range(p["nclass"])]
    ns = len(p["labels_syn"])
    ranges = [tuple(p["ranges"][c]) for c in range(p["nclass"])]
    coeffs = [p["ncd"][c] / ns for c in range(p["nclass"])]
    l, dX, dA = A.syn_match_gcn(theta0, p["xs"], an.detach(), p["labels_syn"], ranges, coeffs, gw_real, metric)
    np.testing.assert_allclose(float(l), float(loss), rtol=1e-10)
    np.testing.assert_allclose(dX.numpy(), gx.numpy(), rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(dA.numpy(), ga.numpy(), rtol=1e-7, atol=1e-11)
