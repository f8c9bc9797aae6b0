"""CPU restatement of MSGC — "Multiple sparse graphs condensation" — /root/reference/graphslim/condensation/msgc.py.

TEST INFRASTRUCTURE ONLY (see oracle/gcond_oracle.py header): imported by tests/ and tests/golden/make_golden.py, never by the
product path.  Pinned against the reference's own MSGC.reduce run in this container (golden G18, tests/test_oracle_golden.py).

MSGC shares train_class / match_loss with GCond (gcond_base.py:156-241) and replaces the dense PGE by `batch_adj` sparse
structures over the SAME n' synthetic nodes: a fixed edge list per structure (reset_adj_batch, msgc.py:178-222), an edge MLP
with training-mode BatchNorm that scores every directed edge (msgc.py:31-39), and a batched dense normalisation
(get_adj_t_syn, msgc.py:224-236).  The condensation model sees the [B, n', n'] batch as B graphs over the same features; its
output is flattened to [B n', C] and matched against labels repeated B times (msgc.py:77, models/sgc.py:45-53).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import gcond_oracle as O

BN_EPS = 1e-5


def msgc_labels(labels_train, n_syn: int, nclass: int):
    """msgc.py:41-64: one row per class, the rest by floor(rate * (n' - C)), the remainder handed out one by one to the class
    with the smallest share-to-rate ratio.  All arithmetic in torch float32 like the reference.  Returns (y_syn int64 [n'],
    per-class counts int64 [C]); y_syn is class-major, classes ascending."""
    lab = torch.as_tensor(np.asarray(labels_train)).long()
    n = lab.shape[0]
    base = torch.ones(nclass)
    rate = F.one_hot(lab, num_classes=nclass).sum(0) / n
    n_each = torch.floor((n_syn - base.sum()) * rate) + base
    left = n_syn - n_each.sum()
    for _ in range(int(left.item())):
        more = n_each / n_each.sum() / rate
        n_each[more.argmin()] += 1
    n_each = n_each.to(torch.int64)
    y_syn = torch.cat([torch.full((int(k),), c, dtype=torch.int64) for c, k in enumerate(n_each)])
    return y_syn, n_each


def msgc_edges(y_syn: torch.Tensor, nclass: int, batch_size: int):
    """msgc.py:178-222 reset_adj_batch: per structure, every node links to (at most) one node of each class — the candidate
    of that class that so far has the fewest links towards the node's own class, ties broken by torch.randperm (the only RNG
    draw, taken from the global torch generator in this order).  Each link is stored in both directions.  Returns
    (rows, cols, batch) int64, concatenated over the structures."""
    n = y_syn.shape[0]
    index = torch.arange(n)
    rows, cols, batch = [], [], []
    for b in range(batch_size):
        n_neighbor = torch.zeros(n, nclass)
        row, col = [], []
        for r in range(n):
            for c in range(nclass):
                c_mask = y_syn == c
                c_mask[r] = False
                if c_mask.sum() == 0 or n_neighbor[r][c] > 1:
                    continue
                link_coef = n_neighbor[c_mask, y_syn[r]]
                sel = link_coef.argmin()
                cand = link_coef == link_coef[sel]
                if cand.sum() == 1:
                    col_id = index[c_mask][sel].item()
                else:
                    cands = index[c_mask][cand]
                    col_id = cands[torch.randperm(cands.shape[0])[0]].item()
                n_neighbor[r, c] += 1
                n_neighbor[col_id, y_syn[r]] += 1
                row += [r, col_id]
                col += [col_id, r]
        rows.append(torch.tensor(row, dtype=torch.int64))
        cols.append(torch.tensor(col, dtype=torch.int64))
        batch.append(torch.full((len(row),), b, dtype=torch.int64))
    return torch.cat(rows), torch.cat(cols), torch.cat(batch)


def init_edge_mlp(d: int, hid: int = 256):
    """msgc.py:31-39 nn.Sequential(Linear(2d, hid), BN, ReLU, Linear(hid, hid), BN, ReLU, Linear(hid, 1)) in
    nn.Module.parameters() order: [W0, b0, g1, be1, W1, b1, g2, be2, w2, b2]; nn.Linear default draws in constructor order."""
    w0, b0 = O.draw_nn_linear(2 * d, hid)
    w1, b1 = O.draw_nn_linear(hid, hid)
    w2, b2 = O.draw_nn_linear(hid, 1)
    return [w0, b0, torch.ones(hid), torch.zeros(hid), w1, b1, torch.ones(hid), torch.zeros(hid), w2, b2]


def _bn_train(x, g, be):
    mu = x.mean(0)
    var = x.var(0, unbiased=False)
    return (x - mu) / torch.sqrt(var + BN_EPS) * g + be


def edge_scores(feat_syn, params, rows, cols):
    """The edge MLP of msgc.py:226-227 (the module is never put into eval mode: batch statistics over all edges of all
    structures)."""
    w0, b0, g1, be1, w1, b1, g2, be2, w2, b2 = params
    e = torch.cat([feat_syn[rows], feat_syn[cols]], dim=1)
    e = F.relu(_bn_train(e @ w0.T + b0, g1, be1))
    e = F.relu(_bn_train(e @ w1.T + b1, g2, be2))
    return torch.sigmoid((e @ w2.T + b2).flatten())


def msgc_adj(feat_syn, params, rows, cols, batch, batch_size):
    """msgc.py:224-236 get_adj_t_syn: scatter, symmetrise, add self loops, D^-1/2 A D^-1/2 per structure.  [B, n', n']."""
    n = feat_syn.shape[0]
    adj = torch.zeros(batch_size, n, n, dtype=feat_syn.dtype)
    adj = adj.index_put((batch, rows, cols), edge_scores(feat_syn, params, rows, cols))
    adj = (adj.transpose(1, 2) + adj) / 2
    adj = adj + torch.eye(n, dtype=feat_syn.dtype).view(1, n, n)
    deg = adj.sum(2)
    dinv = deg.pow(-0.5)
    dinv = torch.where(torch.isinf(dinv), torch.zeros_like(dinv), dinv)
    return adj * dinv.view(batch_size, -1, 1) * dinv.view(batch_size, 1, -1)


def sgc_forward_batched(x, adj, params, nlayers=2):
    """models/sgc.py:37-57 with a [B, n', n'] adjacency: `adj @ x` broadcasts, the result is flattened to [B n', C]."""
    nl = len(params) // 2
    for i in range(nl):
        x = x @ params[2 * i] + params[2 * i + 1]
        if i != nl - 1:
            x = F.relu(x)
    for _ in range(nlayers):
        x = adj @ x
    x = x.reshape(-1, x.shape[-1])
    return F.log_softmax(x, dim=1)


class MSGCState:
    """X~, the edge MLP, their Adams, and the ONE model optimiser that lives across epochs (msgc.py:82-90: optimizer_basic_model
    is built once; basic_model.initialize() re-draws the weights in place, the Adam moments and step count carry over)."""

    def __init__(self, feat_syn, params, y_syn, n_each, nclass, rows, cols, batch, batch_size, lr_feat, lr_adj, lr_model,
                 model_params, nlayers=2):
        self.feat_syn = feat_syn.clone().requires_grad_(True)
        self.params = [p.clone().requires_grad_(True) for p in params]
        self.y_syn, self.n_each, self.nclass = y_syn, n_each, nclass
        self.rows, self.cols, self.batch, self.B = rows, cols, batch, batch_size
        self.y_rep = y_syn.repeat(batch_size)
        self.nlayers = nlayers
        self.opt_feat = O.Adam([self.feat_syn], lr_feat)
        self.opt_adj = O.Adam(self.params, lr_adj)
        self.model = [p.clone().requires_grad_(True) for p in model_params]
        self.opt_model = O.Adam(self.model, lr_model)

    def reinit_model(self, model_params):
        """basic_model.initialize() (msgc.py:93): new values in the SAME parameter tensors."""
        with torch.no_grad():
            for p, q in zip(self.model, model_params):
                p.copy_(q)

    def adj(self, detach=False):
        if detach:
            with torch.no_grad():
                return msgc_adj(self.feat_syn.detach(), [p.detach() for p in self.params], self.rows, self.cols, self.batch, self.B)
        return msgc_adj(self.feat_syn, self.params, self.rows, self.cols, self.batch, self.B)

    def match(self, batches, features, labels, dis_metric="ours"):
        """train_class (gcond_base.py:156-241) on the batched structure.  Returns the loss with graph."""
        adj = self.adj()
        n = self.feat_syn.shape[0]
        loss = torch.zeros((), dtype=self.feat_syn.dtype)
        for c in range(self.nclass):
            gw_real, _ = O.class_real_grads("SGC", self.model, features, labels, batches[c], self.nlayers)
            out = sgc_forward_batched(self.feat_syn, adj, self.model, self.nlayers)
            m = self.y_rep == c
            gw_syn = torch.autograd.grad(F.nll_loss(out[m], self.y_rep[m]), self.model, create_graph=True)
            loss = loss + (float(self.n_each[c]) / n) * O.match_loss(gw_syn, gw_real, dis_metric)
        return loss

    def outer_iteration(self, epoch, batches, features, labels, inner_loop, dis_metric="ours"):
        """msgc.py:95-119.  Returns (loss, d loss / d X~, d loss / d params, detached structure used by the inner steps)."""
        loss = self.match(batches, features, labels, dis_metric)
        grads = torch.autograd.grad(loss, [self.feat_syn] + self.params)
        if epoch % 50 < 10:
            self.opt_adj.step(list(grads[1:]))
        else:
            self.opt_feat.step([grads[0]])
        adj_in = self.adj(detach=True)
        x_in = self.feat_syn.detach()
        for _ in range(inner_loop):
            out = sgc_forward_batched(x_in, adj_in, self.model, self.nlayers)
            gm = torch.autograd.grad(F.nll_loss(out, self.y_rep), self.model)
            self.opt_model.step(list(gm))
        return loss.detach(), grads[0].detach(), [g.detach() for g in grads[1:]], adj_in


def reduce_msgc(adj_full, x, y, idx_train, nclass, *, reduction_rate, hidden, ntrans, epochs, outer_loop, inner_loop, batch_adj,
                lr_feat, lr_adj, lr_model, seed, batch_fn, edge_hidden=256, window=20, dis_metric="ours"):
    """MSGC.__init__ + reduce without checkpoints (msgc.py:18-39,66-130) in the reference's RNG draw order:
    seed_everything -> edge MLP constructor -> model constructor -> reset_adj_batch (randperm tie breaks) -> Random init ->
    model.initialize() per epoch.  `batch_fn(c, call_index)` supplies class c's sampled batch.  Returns per-iteration losses,
    final X~, edge-MLP parameters, the last detached structure and the window averages the checkpoints would save."""
    O.seed_everything(seed)
    feat_full = torch.from_numpy(x).float()
    labels_full = torch.from_numpy(y).long()
    labels_train = y[idx_train]
    n_syn = int(len(idx_train) * reduction_rate)
    y_syn, n_each = msgc_labels(labels_train, n_syn, nclass)
    d = x.shape[1]
    params = init_edge_mlp(d, edge_hidden)
    model0 = O.init_sgc(d, hidden, nclass, ntrans)          # constructor draw (msgc.py:79)
    rows, cols, batch = msgc_edges(y_syn, nclass, batch_adj)
    ncd = {c: int(k) for c, k in enumerate(n_each)}
    sel = O.random_init_indices(idx_train, labels_train, ncd)
    st = MSGCState(feat_full[sel], params, y_syn, n_each, nclass, rows, cols, batch, batch_adj, lr_feat, lr_adj, lr_model, model0)
    calls, losses = 0, []
    xs, adjs = [], []
    for it in range(epochs):
        st.reinit_model(O.init_sgc(d, hidden, nclass, ntrans))
        adj_in = None
        for _ in range(outer_loop):
            batches = []
            for c in range(nclass):
                batches.append(batch_fn(c, calls))
                calls += 1
            l, _, _, adj_in = st.outer_iteration(it, batches, feat_full, labels_full, inner_loop, dis_metric)
            losses.append(float(l))
        xs = (xs + [st.feat_syn.detach().clone()])[-window:]            # FixLenList (msgc.py:239-247)
        adjs = (adjs + [adj_in.clone()])[-window:]
    return dict(losses=losses, feat_syn=st.feat_syn.detach(), params=[p.detach() for p in st.params], adj=adj_in,
                best_x=sum(xs) / len(xs), best_adj=sum(adjs) / len(adjs), y_syn=y_syn, n_each=n_each, rows=rows, cols=cols,
                batch=batch, state=st)
