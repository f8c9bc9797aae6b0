"""CPU restatement of SGDD — "Does Graph Distillation See Like Vision Dataset Counterpart?" —
/root/reference/graphslim/condensation/sgdd.py:14-105 and its structure generator /root/reference/graphslim/models/ignr.py:9-213.

TEST INFRASTRUCTURE ONLY (see oracle/gcond_oracle.py header).  Pinned against the reference's own SGDD.reduce run in this
container (golden G20, tests/test_oracle_golden.py).

SGDD shares train_class / match_loss and the update rule with GCond (sgdd.py:69,80-83) and replaces the PGE by the IGNR graphon
learner: two MLP towers over all n'^2 ordered pairs — net0 on cat(x_i, x_j), net1 on the pair's grid coordinates — coupled by
element-wise products (ignr.py:158-170), BatchNorm in TRAINING mode for the structure train_class differentiates and in EVAL
mode (running statistics) for the structure of the inner steps (ignr.py:151-154,211-213), plus an optimal-transport regulariser
between the learned structure and a fixed corner of the real adjacency (ignr.py:184-209) scaled by args.opt_scale (sgdd.py:70-74).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import gcond_oracle as O

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
SV_EPS = 0.009                   # ignr.py:19,32


def mgrid(n: int) -> torch.Tensor:
    """ignr.py:46-71 get_mgrid(n) for dim = 2: pair (i, j) -> (2 i/(n-1) - 1, 2 j/(n-1) - 1), i slow, float32."""
    g = np.stack(np.mgrid[:n, :n], axis=-1)[None, ...].astype(np.float32)
    g[0, :, :, 0] = g[0, :, :, 0] / (n - 1)
    g[0, :, :, 1] = g[0, :, :, 1] / (n - 1)
    g -= 0.5
    g *= 2.0
    return torch.from_numpy(g).view(-1, 2).to(torch.float32)


def init_ignr(d: int, nnodes: int, mx_size: int, nfeat: int = 128):
    """GraphonLearner.__init__ + reset_parameters (ignr.py:95-149) with the reference's RNG draw order: the six nn.Linear of
    net0 / net1 in constructor order, P ~ U(0, 1), the unused `output` Linear, then reset_parameters() re-draws every Linear in
    module order (net0.{0,1,2}, net1.{0,1,2}, output) — P keeps its first draw.  Returns a dict of tensors."""
    dims = [(2 * d, nfeat), (nfeat, nfeat), (nfeat, 1), (2, nfeat), (nfeat, nfeat), (nfeat, 1)]
    for fi, fo in dims:
        O.draw_nn_linear(fi, fo)
    P = torch.empty(mx_size, nnodes).uniform_(0, 1)
    O.draw_nn_linear(nfeat, 1)
    lin = [O.draw_nn_linear(fi, fo) for fi, fo in dims]
    out = O.draw_nn_linear(nfeat, 1)
    p = {"P": P, "out_w": out[0], "out_b": out[1]}
    for t, base in (("c", 0), ("x", 3)):          # tower c = net0 (pair features), tower x = net1 (coordinates)
        for l in range(3):
            p[f"{t}{l}_w"], p[f"{t}{l}_b"] = lin[base + l]
        for l in range(2):
            p[f"{t}{l}_g"], p[f"{t}{l}_be"] = torch.ones(nfeat), torch.zeros(nfeat)
    return p


# nn.Module.parameters() order of GraphonLearner: own parameter P, then net0, net1, output
PARAM_ORDER = (["P"] + [f"c{l}_{k}" for l in range(2) for k in ("w", "b", "g", "be")] + ["c2_w", "c2_b"]
               + [f"x{l}_{k}" for l in range(2) for k in ("w", "b", "g", "be")] + ["x2_w", "x2_b", "out_w", "out_b"])


def new_running(nfeat=128):
    return {f"{t}{l}": [torch.zeros(nfeat), torch.ones(nfeat)] for t in "cx" for l in range(2)}


def _bn(x, g, be, run, train):
    if train:
        mu, var = x.mean(0), x.var(0, unbiased=False)
        with torch.no_grad():
            n = x.shape[0]
            run[0].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mu.detach())
            run[1].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * x.detach().var(0, unbiased=True) if n > 1 else var.detach())
    else:
        mu, var = run[0], run[1]
    return (x - mu) / torch.sqrt(var + BN_EPS) * g + be


def ignr_structure(feat, p, run, ep_ratio, train):
    """ignr.py:151-177: the two coupled towers -> (A + A^T)/2 -> sigmoid -> zero diagonal.  `run` (running statistics) is
    updated in training mode and read in eval mode."""
    n = feat.shape[0]
    ii = torch.arange(n).repeat_interleave(n)
    jj = torch.arange(n).repeat(n)
    c = torch.cat([feat[ii], feat[jj]], dim=1)
    x = mgrid(n).to(feat.dtype)
    for l in range(2):
        c = F.relu(_bn(c @ p[f"c{l}_w"].T + p[f"c{l}_b"], p[f"c{l}_g"], p[f"c{l}_be"], run[f"c{l}"], train))
        x = F.relu(_bn(x @ p[f"x{l}_w"].T + p[f"x{l}_b"], p[f"x{l}_g"], p[f"x{l}_be"], run[f"x{l}"], train))
        x = x * c
    c = c @ p["c2_w"].T + p["c2_b"]
    x = x @ p["x2_w"].T + p["x2_b"]
    x = (1 - ep_ratio) * x + ep_ratio * c
    adj = x.reshape(n, n)
    adj = (adj + adj.T) / 2
    adj = torch.sigmoid(adj)
    return adj - torch.diag(torch.diag(adj, 0))


def mx_inv(mx: torch.Tensor) -> torch.Tensor:
    """ignr.py:13-27: pseudo-inverse through torch.svd; when the smallest singular value is below 0.009 the entries EQUAL to it
    are zeroed instead of inverted."""
    U, D, V = torch.svd(mx)
    d_min = torch.min(D)
    if d_min < SV_EPS:
        d1 = torch.zeros_like(D)
        d1[D > d_min] = 1 / D[D > d_min]
    else:
        d1 = 1 / D
    return U @ d1.diag() @ V.t()


def mx_inv_sqrt(mx):
    """ignr.py:30-41."""
    U, D, V = torch.svd(mx)
    d_min = torch.min(D)
    if d_min < SV_EPS:
        d1 = torch.zeros_like(D)
        d1[D > d_min] = 1 / D[D > d_min]
    else:
        d1 = 1 / D
    return U @ d1.sqrt().diag() @ V.t(), U @ d1.diag() @ V.t()


def opt_loss(adj, P, Lx_inv):
    """ignr.py:184-209 (the Sinkhorn-normalised copy of P computed there is never used: the raw parameter enters)."""
    ly_rt, ly = mx_inv_sqrt(adj)
    m = adj.shape[0]
    ev = torch.linalg.eigh(ly_rt @ P.t() @ Lx_inv @ P @ ly_rt)[0]
    return torch.abs(ly.diag().sum() * m - 2 * torch.sqrt(ev.clamp(min=2e-20)).sum())


class SGDDState:
    """X~, the IGNR parameters and running statistics, their Adams, the per-epoch model (sgdd.py:17-28,56-60)."""

    def __init__(self, feat_syn, params, labels_syn, num_class_dict, nclass, Lx_inv, lr_feat, lr_adj, lr_model, ep_ratio, opt_scale,
                 nlayers=2):
        self.feat_syn = feat_syn.clone().requires_grad_(True)
        self.p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        self.run = new_running(params["c0_g"].shape[0])
        self.labels_syn, self.ncd, self.nclass = labels_syn, num_class_dict, nclass
        self.Lx_inv, self.ep, self.opt_scale, self.lr_model, self.nlayers = Lx_inv, ep_ratio, opt_scale, lr_model, nlayers
        self.opt_feat = O.Adam([self.feat_syn], lr_feat)
        self.plist = [self.p[k] for k in PARAM_ORDER]
        self.opt_pge = O.Adam(self.plist, lr_adj)
        self.model = self.opt_model = None

    def new_epoch(self, model_params):
        self.model = [q.clone().requires_grad_(True) for q in model_params]
        self.opt_model = O.Adam(self.model, self.lr_model)

    def outer_iteration(self, epoch, batches, features, labels, inner_loop, dis_metric="ours"):
        """sgdd.py:64-95.  Returns (matching loss, regulariser value, d total / d X~, {name: d total / d param}, inference structure)."""
        adj = ignr_structure(self.feat_syn, self.p, self.run, self.ep, train=True)
        reg = opt_loss(adj, self.p["P"], self.Lx_inv)
        ml = O.train_class("SGC", self.model, batches, features, labels, self.feat_syn, O.dense_gcn_norm(adj), self.labels_syn,
                           self.ncd, self.nclass, self.nlayers, dis_metric)
        total = ml + (self.opt_scale * reg if self.opt_scale > 0 else 0.0)
        grads = torch.autograd.grad(total, [self.feat_syn] + self.plist, allow_unused=True)
        gp = [g if g is not None else torch.zeros_like(q) for g, q in zip(grads[1:], self.plist)]
        if epoch % 50 < 10:
            self.opt_pge.step(gp)
        else:
            self.opt_feat.step([grads[0]])
        with torch.no_grad():
            x_in = self.feat_syn.detach()
            adj_in = ignr_structure(x_in, {k: v.detach() for k, v in self.p.items()}, self.run, self.ep, train=False)
            adj_in_norm = O.dense_gcn_norm(adj_in)
        for i in range(inner_loop):      # (sgdd.py:62 model.train(): dropout draws a fresh mask per step, like GCond — golden G30)
            out = O.sgc_forward(x_in, adj_in_norm, self.model, self.nlayers, ("inner", i, None))
            gm = torch.autograd.grad(F.nll_loss(out, self.labels_syn), self.model)
            self.opt_model.step(list(gm))
        return ml.detach(), reg.detach(), grads[0].detach(), dict(zip(PARAM_ORDER, [g.detach() for g in gp])), adj_in


def reduce_sgdd(adj_full, x, y, idx_train, nclass, *, reduction_rate, hidden, ntrans, epochs, outer_loop, inner_loop, mx_size,
                opt_scale, ep_ratio, lr_feat, lr_adj, lr_model, seed, batch_fn, dis_metric="ours"):
    """SGDD.__init__ + reduce without checkpoints in the reference's RNG draw order: seed_everything -> IGNR constructor +
    reset_parameters -> Random init -> model constructor -> model.initialize() per epoch."""
    O.seed_everything(seed)
    feat_full = torch.from_numpy(x).float()
    labels_full = torch.from_numpy(y).long()
    labels_train = y[idx_train]
    labels_syn_np, ncd, _ = O.generate_labels_syn(labels_train, reduction_rate)
    d = x.shape[1]
    mx = min(mx_size, adj_full.shape[0])
    params = init_ignr(d, len(labels_syn_np), mx)
    sel = O.random_init_indices(idx_train, labels_train, ncd)
    Lx_inv = mx_inv(torch.from_numpy(np.asarray(adj_full[:mx, :mx].todense(), dtype=np.float32)))
    st = SGDDState(feat_full[sel], params, torch.from_numpy(labels_syn_np), ncd, nclass, Lx_inv, lr_feat, lr_adj, lr_model, ep_ratio,
                   opt_scale)
    O.init_sgc(d, hidden, nclass, ntrans)            # model constructor draw (sgdd.py:54)
    calls, losses, regs = 0, [], []
    adj_in = None
    for it in range(epochs):
        st.new_epoch(O.init_sgc(d, hidden, nclass, ntrans))
        for _ in range(outer_loop):
            batches = []
            for c in range(nclass):
                batches.append(batch_fn(c, calls))
                calls += 1
            ml, reg, _, _, adj_in = st.outer_iteration(it, batches, feat_full, labels_full, inner_loop, dis_metric)
            losses.append(float(ml))
            regs.append(float(reg))
    return dict(losses=losses, regs=regs, feat_syn=st.feat_syn.detach(), params={k: v.detach() for k, v in st.p.items()},
                running={k: [v[0].clone(), v[1].clone()] for k, v in st.run.items()}, adj=adj_in, labels_syn=labels_syn_np, state=st)
