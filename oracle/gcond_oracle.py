"""CPU oracle for the GCond gradient-matching hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (PyTorch-CPU autograd + numpy/scipy, fp32 or fp64) of the
algorithm on the reference's GCond outer-loop path.  It is *not* the product: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it, and only as the
checker / reported baseline.  The product path (graphslim_amd + libgslim_hip.so) never imports it.

Parity pinning: the reference's own tests hold no vectors for this path, so this
oracle is pinned against fixtures produced by importing the reference itself
(`tests/golden/make_golden.py` -> `tests/golden/*.npz`, checked by `tests/test_oracle_golden.py`).
The third-party native ops the reference calls (torch_sparse.spmm_sum / sample_adj, torch_sparse
0.6.18; torch_geometric NeighborSampler 2.5-2.7) are absent from /root/reference; they are restated
from their published semantics (row-wise weighted sum; per-row sampling without replacement with
targets-first relabelling) and are "parity unpinned" at that boundary.

Each function cites the reference file:line (relative to /root/reference) it follows.
"""
from __future__ import annotations

import math
from collections import Counter

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# normalisation  (graphslim/utils.py:403-458)
# --------------------------------------------------------------------------------------------


def sparse_gcn_norm(adj: sp.spmatrix) -> sp.csr_matrix:
    """D^-1/2 (A+I) D^-1/2 in scipy float64 — graphslim/utils.py:451-458."""
    adj = sp.csr_matrix(adj, dtype=np.float64) + sp.eye(adj.shape[0], dtype=np.float64)
    rowsum = np.asarray(adj.sum(1)).flatten()
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -0.5)
    r_inv[np.isinf(r_inv)] = 0.0
    d = sp.diags(r_inv)
    return sp.csr_matrix(d.dot(adj).dot(d))


def normalize_adj_sparse(adj: sp.spmatrix):
    """graphslim/utils.py:403-413: scipy f64 normalisation, cast to fp32, coalesce, transpose.

    Returns CSR (rowptr int64, col int64, val fp32) of the *transposed* normalised matrix (the
    reference builds SparseTensor(...).t(); for a symmetric graph it equals the matrix itself).
    """
    mx = sparse_gcn_norm(adj).astype(np.float32).T.tocsr()
    mx.sum_duplicates()
    mx.sort_indices()
    return mx.indptr.astype(np.int64), mx.indices.astype(np.int64), mx.data.astype(np.float32)


def dense_gcn_norm(adj: torch.Tensor) -> torch.Tensor:
    """graphslim/utils.py:429-439 (row-sum degrees only, differentiable)."""
    mx = adj + torch.eye(adj.shape[0], dtype=adj.dtype)
    rowsum = mx.sum(1)
    r_inv = rowsum.pow(-0.5)
    r_inv = torch.where(torch.isinf(r_inv), torch.zeros_like(r_inv), r_inv)
    return r_inv[:, None] * mx * r_inv[None, :]


# --------------------------------------------------------------------------------------------
# label allocation  (graphslim/condensation/gcond_base.py:79-115)
# --------------------------------------------------------------------------------------------


def generate_labels_syn(labels_train, reduction_rate: float):
    """Returns (labels_syn int64 [n'], num_class_dict {class: count} in visiting order,
    syn_class_indices {class: [lo, hi]}).  Classes are visited by ascending train count, ties by
    first appearance (Counter insertion order + stable sort); the largest class takes the
    remainder — gcond_base.py:93-111."""
    labels_train = np.asarray(labels_train).tolist()
    counter = Counter(labels_train)
    n = len(labels_train)
    ordered = sorted(counter.items(), key=lambda kv: kv[1])
    num_class_dict, idx, labels_syn, sum_ = {}, {}, [], 0
    for ix, (c, num) in enumerate(ordered):
        if ix == len(ordered) - 1:
            cnt = max(int(n * reduction_rate) - sum_, 1)
        else:
            cnt = max(int(num * reduction_rate), 1)
            sum_ += cnt
        num_class_dict[c] = cnt
        idx[c] = [len(labels_syn), len(labels_syn) + cnt]
        labels_syn += [c] * cnt
    return np.array(labels_syn, dtype=np.int64), num_class_dict, idx


def random_init_indices(idx_train, labels_train, num_class_dict, rng=np.random):
    """sparsification/random.py:9-17: per class (num_class_dict order) a np.random.permutation of
    that class's train ids, first `cnt` kept."""
    out = []
    idx_train = np.asarray(idx_train)
    labels_train = np.asarray(labels_train)
    for c, cnt in num_class_dict.items():
        sel = rng.permutation(idx_train[labels_train == c])
        out.append(sel[:cnt])
    return np.hstack(out)


# --------------------------------------------------------------------------------------------
# models  (models/layers.py:354-381 MyLinear, :17-51 GraphConvolution, models/sgc.py, gcn.py)
# --------------------------------------------------------------------------------------------


def init_linear(n_in: int, n_out: int, dtype=torch.float32):
    """MyLinear/GraphConvolution.reset_parameters (layers.py:369-373 / :29-33):
    W[in,out] ~ U(+-1/sqrt(in)) drawn first, then b ~ U(+-1/sqrt(in))."""
    stdv = 1.0 / math.sqrt(n_in)
    w = torch.empty(n_in, n_out, dtype=torch.float32).uniform_(-stdv, stdv)
    b = torch.zeros(n_out, dtype=torch.float32).uniform_(-stdv, stdv)
    return w.to(dtype), b.to(dtype)


def init_sgc(nfeat, nhid, nclass, ntrans, dtype=torch.float32):
    """models/sgc.py:16-35 layer shapes; base.py:44-49 initialisation order."""
    if ntrans == 1:
        dims = [(nfeat, nclass)]
    else:
        dims = [(nfeat, nhid)] + [(nhid, nhid)] * (ntrans - 2) + [(nhid, nclass)]
    params = []
    for a, b in dims:
        w, bb = init_linear(a, b, dtype)
        params += [w, bb]
    return params


def init_gcn(nfeat, nhid, nclass, nlayers, dtype=torch.float32):
    """models/gcn.py:8-23."""
    if nlayers == 1:
        dims = [(nfeat, nclass)]
    else:
        dims = [(nfeat, nhid)] + [(nhid, nhid)] * (nlayers - 2) + [(nhid, nclass)]
    params = []
    for a, b in dims:
        w, bb = init_linear(a, b, dtype)
        params += [w, bb]
    return params


def _prop(adj, x):
    if adj.layout != torch.strided:
        return torch.sparse.mm(adj, x)
    return adj @ x


# Dropout of the condensation model (models/sgc.py:42: F.dropout(x, self.dropout, training=self.training) behind every transform but
# the last; the model is in training mode inside train_class and the inner steps, gcond.py:45, so each forward call draws a fresh mask:
# one per class on the real batch, one per class on the synthetic graph, one per inner step).  DROPOUT = None: dropout 0 (every shipped
# configuration).  Otherwise a callable DROPOUT(where, x) -> multiplier of x's shape (kept entries 1 / (1 - p), dropped entries 0), with
# `where` = (kind, index, rows): kind "real" | "syn" | "inner", index = the class (real, syn) or the inner step, rows = the batch's node
# ids n_id (real) or None (every synthetic row).  The golden test replays the reference's recorded masks through it (G26); the GPU
# tests hand it the engine's counter-based masks (graphslim_amd.engine.dropout_keep).
DROPOUT = None


def sgc_forward(x, adj, params, nlayers=2, where=None):
    """models/sgc.py:37-57: T x (x@W+b [relu, dropout]) then K propagations, log_softmax.
    `adj` is a dense tensor, one sparse tensor, or a list (outermost hop first)."""
    nl = len(params) // 2
    for i in range(nl):
        x = x @ params[2 * i] + params[2 * i + 1]
        if i != nl - 1:
            x = F.relu(x)
            if DROPOUT is not None:
                x = x * DROPOUT(where, x)
    for i in range(nlayers):
        x = _prop(adj[i] if isinstance(adj, (list, tuple)) else adj, x)
    return F.log_softmax(x, dim=1)


BN_EPS = 1e-5


def sgc_bn_forward(x, adj, params, nlayers=2):
    """models/sgc.py:27-33,37-57 with with_bn and ntrans = 2, as GCond runs it: check_bn (gcond_base.py:261-285) puts the
    BatchNorm1d between the two transforms into eval mode before every train_class and inner step, and nothing ever runs it in
    training mode in between, so its running statistics stay (0, 1) and it is the per-column affine map
    y = x / sqrt(1 + eps) * weight + bias with TRAINABLE weight / bias.  params = [W1, b1, W2, b2, bn.weight, bn.bias]
    (nn.Module.parameters() order: layers first, then bns)."""
    w1, b1, w2, b2, g, be = params
    x = x @ w1 + b1
    x = x / torch.sqrt(torch.tensor(1.0 + BN_EPS, dtype=x.dtype)) * g + be
    x = F.relu(x)
    x = x @ w2 + b2
    for i in range(nlayers):
        x = _prop(adj[i] if isinstance(adj, (list, tuple)) else adj, x)
    return F.log_softmax(x, dim=1)


def gcn_forward(x, adj, params, where=None):
    """models/base.py:51-78 + layers.py:36-51: per layer adj @ (x@W) + b, then relu and dropout behind every layer but the last.
    Dropout: the hook gets `where` = (kind, index, rows, layer) with rows = the node ids of THIS layer's output rows (a sampled block's
    targets are the first rows of its sources, loader.py:205-210) and layer = 0, 1, .. (the hidden layer the mask multiplies)."""
    nl = len(params) // 2
    for i in range(nl):
        a = adj[i] if isinstance(adj, (list, tuple)) else adj
        x = _prop(a, x @ params[2 * i]) + params[2 * i + 1]
        if i != nl - 1:
            x = F.relu(x)
            if DROPOUT is not None:
                kind, index, rows = where
                x = x * DROPOUT((kind, index, None if rows is None else rows[:x.shape[0]], i), x)
    return F.log_softmax(x, dim=1)


# --------------------------------------------------------------------------------------------
# match loss  (condensation/utils.py:12-106)
# --------------------------------------------------------------------------------------------


def distance_wb(gwr, gws):
    """condensation/utils.py:67-106: 2-D grads transposed (rows = output units), bias -> 0,
    eps 1e-6 on the norm *product*."""
    if gwr.dim() == 1:
        return 0
    if gwr.dim() == 2:
        gwr, gws = gwr.T, gws.T
    else:
        gwr = gwr.reshape(gwr.shape[0], -1)
        gws = gws.reshape(gws.shape[0], -1)
    return torch.sum(1 - torch.sum(gwr * gws, dim=-1) /
                     (torch.norm(gwr, dim=-1) * torch.norm(gws, dim=-1) + 0.000001))


def match_loss(gw_syn, gw_real, dis_metric="ours"):
    """condensation/utils.py:12-65."""
    if dis_metric == "ours":
        dis = torch.zeros((), dtype=gw_syn[0].dtype)
        for r, s in zip(gw_real, gw_syn):
            dis = dis + distance_wb(r, s)
        return dis
    r = torch.cat([g.reshape(-1) for g in gw_real])
    s = torch.cat([g.reshape(-1) for g in gw_syn])
    if dis_metric == "mse":
        return torch.sum((s - r) ** 2)
    if dis_metric == "cos":
        return 1 - torch.sum(r * s, dim=-1) / (torch.norm(r, dim=-1) * torch.norm(s, dim=-1) + 0.000001)
    raise SystemExit("DC error: unknown distance function")


# --------------------------------------------------------------------------------------------
# PGE  (models/parametrized_adj.py)
# --------------------------------------------------------------------------------------------


def pge_hidden(dataset: str, reduction_rate: float) -> int:
    """parametrized_adj.py:9-17."""
    if dataset in ("ogbn-arxiv", "arxiv", "flickr"):
        return 256
    if dataset == "reddit":
        return 128 if reduction_rate == 0.01 else 256
    return 128


def init_pge(nfeat, nhid, nlayers=3, dtype=torch.float32):
    """nn.Linear / BatchNorm1d default reset_parameters (parametrized_adj.py:20-27,79-86):
    per Linear: kaiming_uniform(a=sqrt5) == U(+-1/sqrt(fan_in)) for W[out,in] then
    b ~ U(+-1/sqrt(fan_in)); BN: weight=1, bias=0.  Returns dict with lists."""
    dims = [(2 * nfeat, nhid)] + [(nhid, nhid)] * (nlayers - 2) + [(nhid, 1)]
    lin = []
    for fi, fo in dims:
        bound = 1.0 / math.sqrt(fi)
        w = torch.empty(fo, fi, dtype=torch.float32).uniform_(-bound, bound)
        b = torch.empty(fo, dtype=torch.float32).uniform_(-bound, bound)
        lin += [w.to(dtype), b.to(dtype)]
    bns = []
    for _ in range(nlayers - 1):
        bns += [torch.ones(nhid, dtype=dtype), torch.zeros(nhid, dtype=dtype)]
    return {"lin": lin, "bn": bns}


# Test hook (tests/test_gpu_fullsize.py): when set, called as RELU_HOOK(layer index, pre-activation) in place of F.relu inside the PGE's
# MLP — the full-size tests use it to find ReLU inputs within rounding distance of zero (a tie the fp32 forward of another summation
# order resolves the other way) and to replay the oracle with such a tie resolved either way.  None: F.relu.
RELU_HOOK = None


def pge_forward(x, pge, eps=1e-5, n_chunks=1):
    """parametrized_adj.py:40-71.  Pair k = r*n + c -> [x_c || x_r] (meshgrid order :30-32);
    BatchNorm always with batch statistics (biased variance); (A+A^T)/2, sigmoid, zero diag.
    n_chunks = 5 is the reddit / r >= 0.01 branch (:41-55): the flat pair list goes through the layers in
    np.array_split chunks, each normalised by its own batch statistics."""
    n = x.shape[0]
    lin, bn = pge["lin"], pge["bn"]
    idx_c = torch.arange(n).repeat(n)            # k % n
    idx_r = torch.arange(n).repeat_interleave(n)  # k // n
    nl = len(lin) // 2

    def mlp(e):
        for i in range(nl):
            e = e @ lin[2 * i].T + lin[2 * i + 1]
            if i != nl - 1:
                mu = e.mean(0)
                var = e.var(0, unbiased=False)
                e = (e - mu) / torch.sqrt(var + eps) * bn[2 * i] + bn[2 * i + 1]
                e = F.relu(e) if RELU_HOOK is None else RELU_HOOK(i, e)
        return e

    if n_chunks > 1:
        parts = [mlp(torch.cat([x[idx_c[torch.from_numpy(ix)]], x[idx_r[torch.from_numpy(ix)]]], dim=1))
                 for ix in np.array_split(np.arange(n * n), n_chunks)]
        e = torch.cat(parts)
    else:
        e = mlp(torch.cat([x[idx_c], x[idx_r]], dim=1))
    adj = e.reshape(n, n)
    adj = (adj + adj.T) / 2
    adj = torch.sigmoid(adj)
    adj = adj - torch.diag(torch.diag(adj, 0))
    return adj


# --------------------------------------------------------------------------------------------
# neighbour sampling  (dataset/loader.py:187-224 -> PyG NeighborSampler -> torch_sparse.sample_adj)
# --------------------------------------------------------------------------------------------


def sample_blocks(rowptr, col, val, seeds, sizes, rng):
    """Restatement of NeighborSampler.sample (return_e_id=False) over the normalised CSR.

    Per hop, for each frontier node keep all neighbours if deg <= f else f without replacement;
    new nodes are appended to n_id after the current targets (targets first); edge values are the
    full-graph normalised weights (no re-normalisation).  Returns (batch_size, n_id, adjs) with
    adjs outermost hop first, each a (rowptr, col, val, shape) CSR block [n_dst x n_src].
    """
    n_id = np.asarray(seeds, dtype=np.int64)
    adjs = []
    for f in sizes:
        local = {int(v): i for i, v in enumerate(n_id)}
        order = list(n_id)
        brp, bc, bv = [0], [], []
        for v in n_id:
            lo, hi = int(rowptr[v]), int(rowptr[v + 1])
            deg = hi - lo
            if f < 0 or deg <= f:
                pick = np.arange(lo, hi)
            else:
                pick = lo + rng.choice(deg, size=f, replace=False)
            for e in pick:
                u = int(col[e])
                if u not in local:
                    local[u] = len(order)
                    order.append(u)
                bc.append(local[u])
                bv.append(val[e])
            brp.append(len(bc))
        n_dst = len(n_id)
        n_id = np.asarray(order, dtype=np.int64)
        adjs.append((np.asarray(brp, np.int64), np.asarray(bc, np.int64),
                     np.asarray(bv, np.float32), (n_dst, len(n_id))))
    return len(seeds), n_id, adjs[::-1]


def block_to_torch(block, dtype=torch.float32):
    rp, c, v, shape = block
    return torch.sparse_csr_tensor(torch.as_tensor(rp), torch.as_tensor(c),
                                   torch.as_tensor(v).to(dtype), size=shape).to_sparse_coo().coalesce()


# --------------------------------------------------------------------------------------------
# train_class  (condensation/gcond_base.py:156-241)
# --------------------------------------------------------------------------------------------


def model_forward(kind, x, adj, params, nlayers, where=None):
    if kind == "SGC":
        return sgc_forward(x, adj, params, nlayers, where)
    if kind == "SGC_BN":
        if DROPOUT is not None:
            raise NotImplementedError("oracle: dropout is restated for SGC and GCN without BatchNorm only")
        return sgc_bn_forward(x, adj, params, nlayers)
    return gcn_forward(x, adj, params, where)


def class_real_grads(kind, params, features, labels, batch, nlayers=2, cls=None):
    """gcond_base.py:212-224: gradient of the real-batch NLL wrt model parameters (detached)."""
    batch_size, n_id, adjs = batch
    n_id = torch.as_tensor(n_id)
    dt = params[0].dtype
    adjs_t = [block_to_torch(a, dt) for a in adjs]
    out = model_forward(kind, features[n_id], adjs_t, params, nlayers, ("real", cls, n_id))
    loss = F.nll_loss(out, labels[n_id[:batch_size]])
    gw = torch.autograd.grad(loss, params)
    return [g.detach().clone() for g in gw], loss.detach()


def train_class(kind, params, batches, features, labels, feat_syn, adj_syn_norm, labels_syn,
                num_class_dict, nclass, nlayers=2, dis_metric="ours", return_parts=False):
    """gcond_base.py:156-241.  `params` must require grad.  `batches[c]` = (batch_size, n_id, adjs).
    Returns the scalar matching loss with graph to feat_syn / adj_syn_norm."""
    n_syn = labels_syn.shape[0]
    loss = torch.zeros((), dtype=feat_syn.dtype)
    parts = []
    for c in range(nclass):
        gw_real, _ = class_real_grads(kind, params, features, labels, batches[c], nlayers, c)
        out_syn = model_forward(kind, feat_syn, adj_syn_norm, params, nlayers, ("syn", c, None))
        m = labels_syn == c
        loss_syn = F.nll_loss(out_syn[m], labels_syn[m])
        gw_syn = torch.autograd.grad(loss_syn, params, create_graph=True)
        coeff = num_class_dict[c] / n_syn
        ml = match_loss(gw_syn, gw_real, dis_metric)
        loss = loss + coeff * ml
        if return_parts:
            parts.append((gw_real, [g.detach() for g in gw_syn], float(ml)))
    return (loss, parts) if return_parts else loss


# --------------------------------------------------------------------------------------------
# Adam (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8) — gcond_base.py:68-69, gcond.py:44
# --------------------------------------------------------------------------------------------


class Adam:
    def __init__(self, params, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.params, self.lr, self.wd, self.b1, self.b2, self.eps = params, lr, weight_decay, *betas, eps
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.t = 0

    @torch.no_grad()
    def step(self, grads):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            if self.wd != 0:
                g = g + self.wd * p
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-self.lr / bc1)


# --------------------------------------------------------------------------------------------
# one outer-loop iteration  (condensation/gcond.py:47-72)
# --------------------------------------------------------------------------------------------


class GCondState:
    """Holds X~, PGE params, their Adams, and the per-epoch model — gcond_base.py:62-69, gcond.py:38-45."""

    def __init__(self, feat_syn, pge, labels_syn, num_class_dict, nclass, lr_feat, lr_adj, lr_model,
                 kind="SGC", nlayers=2):
        self.feat_syn = feat_syn.clone().requires_grad_(True)
        self.pge = {"lin": [p.clone().requires_grad_(True) for p in pge["lin"]],
                    "bn": [p.clone().requires_grad_(True) for p in pge["bn"]]}
        self.labels_syn = labels_syn
        self.num_class_dict, self.nclass = num_class_dict, nclass
        self.kind, self.nlayers = kind, nlayers
        self.lr_model = lr_model
        self.opt_feat = Adam([self.feat_syn], lr_feat)
        self.opt_pge = Adam(self.pge_param_list(), lr_adj)
        self.model = None
        self.opt_model = None

    def pge_param_list(self):
        """nn.Module.parameters() order: layers.{0,1,2}.{weight,bias} then bns.{0,1}.{weight,bias}."""
        return self.pge["lin"] + self.pge["bn"]

    def new_epoch(self, model_params):
        """gcond.py:42-45: fresh theta and fresh Adam state every epoch."""
        self.model = [p.clone().requires_grad_(True) for p in model_params]
        self.opt_model = Adam(self.model, self.lr_model)

    def outer_iteration_ex(self, batches, features, labels, inner_loop, *, step_pge, step_feat, run_inference, run_inner,
                           identity=False, dis_metric="ours"):
        """The sibling loops: DosCond (doscond.py:40-50: both optimisers, no inner steps), GCondX (gcondx.py:30,44-66:
        adj_syn = I, feature step unless ol % 5 < 1, inner steps on I), DosCondX (doscondx.py:30,42-49)."""
        n = self.feat_syn.shape[0]
        if identity:
            adj_norm = torch.eye(n, dtype=self.feat_syn.dtype)
        else:
            adj_norm = dense_gcn_norm(pge_forward(self.feat_syn, self.pge))
        loss = train_class(self.kind, self.model, batches, features, labels, self.feat_syn, adj_norm, self.labels_syn,
                           self.num_class_dict, self.nclass, self.nlayers, dis_metric)
        plist = self.pge_param_list()
        wrt = [self.feat_syn] + ([] if identity else plist)
        grads = torch.autograd.grad(loss, wrt, allow_unused=True)
        if step_pge and not identity:
            self.opt_pge.step([g if g is not None else torch.zeros_like(p) for g, p in zip(grads[1:], plist)])
        if step_feat:
            self.opt_feat.step([grads[0]])
        adj_inner = None
        with torch.no_grad():
            if identity:
                adj_inner, adj_inner_norm = torch.eye(n), torch.eye(n)
            elif run_inference:
                adj_inner = pge_forward(self.feat_syn.detach(), {"lin": [p.detach() for p in self.pge["lin"]],
                                                                 "bn": [p.detach() for p in self.pge["bn"]]})
                adj_inner_norm = dense_gcn_norm(adj_inner)
        if run_inner:
            for i in range(inner_loop):
                out = model_forward(self.kind, self.feat_syn.detach(), adj_inner_norm, self.model, self.nlayers, ("inner", i, None))
                gm = torch.autograd.grad(F.nll_loss(out, self.labels_syn), self.model)
                self.opt_model.step(list(gm))
        return loss.detach(), grads[0].detach(), adj_inner

    def outer_iteration(self, epoch, batches, features, labels, inner_loop, dis_metric="ours"):
        """gcond.py:47-72.  Returns (loss value, grads wrt feat_syn, grads wrt pge params)."""
        adj_syn = pge_forward(self.feat_syn, self.pge)
        adj_norm = dense_gcn_norm(adj_syn)
        loss = train_class(self.kind, self.model, batches, features, labels, self.feat_syn, adj_norm,
                           self.labels_syn, self.num_class_dict, self.nclass, self.nlayers, dis_metric)
        plist = self.pge_param_list()
        grads = torch.autograd.grad(loss, [self.feat_syn] + plist, allow_unused=True)
        g_feat = grads[0]
        g_pge = [g if g is not None else torch.zeros_like(p) for g, p in zip(grads[1:], plist)]
        if epoch % 50 < 10:
            self.opt_pge.step(g_pge)
        else:
            self.opt_feat.step([g_feat])
        adj_inner = self.inference_and_inner(inner_loop)
        return loss.detach(), g_feat.detach(), [g.detach() for g in g_pge], adj_inner

    def inference_and_inner(self, inner_loop):
        """gcond.py:63-72: PGE.inference on the just-updated (X~, theta_PGE), dense_gcn_norm, `inner_loop` model steps.
        Returns the raw inference adjacency."""
        with torch.no_grad():
            feat_inner = self.feat_syn.detach()
            adj_inner = pge_forward(feat_inner, {"lin": [p.detach() for p in self.pge["lin"]],
                                                 "bn": [p.detach() for p in self.pge["bn"]]})
            adj_inner_norm = dense_gcn_norm(adj_inner)
        for i in range(inner_loop):
            out = model_forward(self.kind, feat_inner, adj_inner_norm, self.model, self.nlayers, ("inner", i, None))
            l_in = F.nll_loss(out, self.labels_syn)
            gm = torch.autograd.grad(l_in, self.model)
            self.opt_model.step(list(gm))
        return adj_inner


# --------------------------------------------------------------------------------------------
# evaluation GCN  (models/base.py:80-225, evaluation/eval_agent.py:159-291, dataset/utils.py:8-66)
# --------------------------------------------------------------------------------------------


def sparsify(adj_syn, threshold):
    """dataset/utils.py:47-48,60-61 (method gcond, non-GAT/MLP eval model)."""
    adj_syn = adj_syn.clone()
    if threshold > 0:
        adj_syn[adj_syn < threshold] = 0
    return adj_syn


def accuracy(output, labels):
    """graphslim/utils.py:516-538."""
    preds = output.max(1)[1].type_as(labels)
    return preds.eq(labels).double().sum() / len(labels)


def csr_to_torch(rowptr, col, val, n, dtype=torch.float32):
    return torch.sparse_csr_tensor(torch.as_tensor(rowptr), torch.as_tensor(col),
                                   torch.as_tensor(val).to(dtype), size=(n, n)).to_sparse_coo().coalesce()


def fit_with_val(params, adj_syn, feat_syn, labels_syn, feat_full, adj_full_norm, idx_val, labels_val,
                 train_iters, lr=0.01, weight_decay=5e-4, trace=None):
    """models/base.py:80-187 for GCN(mode='eval') on a reduced graph, transductive validation.
    `params` = freshly initialised [W1,b1,W2,b2]; returns (best_val_acc, best_params)."""
    adj = dense_gcn_norm(adj_syn)
    params = [p.clone().requires_grad_(True) for p in params]
    opt = Adam(params, lr, weight_decay)
    best, best_w = 0.0, None
    for i in range(train_iters):
        if i == train_iters // 2 and lr > 0.001:
            opt = Adam(params, lr * 0.1, weight_decay)
        out = gcn_forward(feat_syn, adj, params)
        loss = F.nll_loss(out, labels_syn)
        g = torch.autograd.grad(loss, params)
        opt.step(list(g))
        with torch.no_grad():
            o = gcn_forward(feat_full, adj_full_norm, params)
            acc = accuracy(o[idx_val], labels_val)
            if trace is not None:
                trace.append((float(loss), float(acc)))
            if acc > best:
                best = acc
                best_w = [p.detach().clone() for p in params]
    if best_w is None:
        best_w = [p.detach().clone() for p in params]
    return float(best), best_w


def fit_with_val_full(params, feat_train, adj_train_norm, idx_loss, labels_loss, feat_val, adj_val_norm, idx_val, labels_val,
                      train_iters, lr=0.01, weight_decay=5e-4, trace=None):
    """models/base.py:80-187 with reduced=False: the GCN is trained on the ORIGINAL graph (trans: full graph, loss on
    output[idx_train]; ind: train sub-graph, loss on every row) and validated every epoch like the reduced case.
    idx_loss = rows whose loss is taken (None = all rows)."""
    params = [p.clone().requires_grad_(True) for p in params]
    opt = Adam(params, lr, weight_decay)
    best, best_w = 0.0, None
    for i in range(train_iters):
        if i == train_iters // 2 and lr > 0.001:
            opt = Adam(params, lr * 0.1, weight_decay)
        out = gcn_forward(feat_train, adj_train_norm, params)
        loss = F.nll_loss(out if idx_loss is None else out[idx_loss], labels_loss)
        g = torch.autograd.grad(loss, params)
        opt.step(list(g))
        with torch.no_grad():
            o = gcn_forward(feat_val, adj_val_norm, params)
            acc = accuracy(o if idx_val is None else o[idx_val], labels_val)
            if trace is not None:
                trace.append((float(loss), float(acc)))
            if acc > best:
                best = acc
                best_w = [p.detach().clone() for p in params]
    if best_w is None:
        best_w = [p.detach().clone() for p in params]
    return float(best), best_w


def eval_test(params, feat_full, adj_full_norm, idx_test, labels_test):
    """models/base.py:190-225 (transductive)."""
    with torch.no_grad():
        o = gcn_forward(feat_full, adj_full_norm, params)
        return float(accuracy(o[idx_test], labels_test))


# --------------------------------------------------------------------------------------------
# whole-run restatement: GCond.__init__/reduce (gcond_base.py:30-70, gcond.py:18-81),
# intermediate_evaluation (gcond_base.py:287-358), Evaluator.evaluate (eval_agent.py:234-291).
# RNG draw ORDER follows the reference so that a seeded run reproduces its trajectory.
# --------------------------------------------------------------------------------------------


def _kaiming_bound(fan_in):
    """torch.nn.init.kaiming_uniform_(a=sqrt(5)) bound, computed the way torch computes it."""
    gain = math.sqrt(2.0 / (1 + math.sqrt(5) ** 2))
    return math.sqrt(3.0) * (gain / math.sqrt(fan_in))


def draw_nn_linear(fan_in, fan_out):
    w = torch.empty(fan_out, fan_in).uniform_(-_kaiming_bound(fan_in), _kaiming_bound(fan_in))
    b = torch.empty(fan_out).uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
    return w, b


def init_pge_reference_order(nfeat, nhid, nlayers=3):
    """PGE.__init__ (parametrized_adj.py:20-37): every nn.Linear is drawn once by its constructor and
    once more by self.reset_parameters(); the second draw is what remains."""
    dims = [(2 * nfeat, nhid)] + [(nhid, nhid)] * (nlayers - 2) + [(nhid, 1)]
    for fi, fo in dims:
        draw_nn_linear(fi, fo)
    lin = []
    for fi, fo in dims:
        lin += list(draw_nn_linear(fi, fo))
    bn = []
    for _ in range(nlayers - 1):
        bn += [torch.ones(nhid), torch.zeros(nhid)]
    return {"lin": lin, "bn": bn}


def seed_everything(seed):
    """graphslim/utils.py:86-91."""
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def run_eval_gcn(feat_syn, adj_syn, labels_syn, g, hidden, nclass, eval_epochs, lr=0.01, nlayers=2):
    """One `test_with_val` / `Evaluator.test` run: GCN ctor draw + initialize() draw, fit, test.  nlayers: GraphConvolution layers
    (models/gcn.py:8-23; `--nlayers` is shared with the condensation model)."""
    d = feat_syn.shape[1]
    init_gcn(d, hidden, nclass, nlayers)            # constructor draw (discarded by initialize())
    params = init_gcn(d, hidden, nclass, nlayers)   # fit_with_val -> self.initialize()
    best_val, w = fit_with_val(params, adj_syn, feat_syn, labels_syn, g["feat_full"], g["adj_full_norm"],
                               g["idx_val"], g["labels_val"], eval_epochs, lr=lr)
    return best_val, eval_test(w, g["feat_full"], g["adj_full_norm"], g["idx_test"], g["labels_test"])


def reduce_and_evaluate(adj_full, x, y, idx_train, idx_val, idx_test, nclass, *, reduction_rate, hidden, ntrans,
                        pge_hid, epochs, outer_loop, inner_loop, checkpoints, lr_feat, lr_adj, lr_model, sizes,
                        threshold, seed, run_inter_eval, eval_epochs, run_eval, batch_fn):
    """Restates a whole reference run.  `batch_fn(c, call_index)` supplies the sampled batch of class c
    (the replayable stand-in for dataset/loader.py:187-224).  Returns a dict of trajectories."""
    seed_everything(seed)
    feat_full = torch.from_numpy(x).float()
    labels_full = torch.from_numpy(y).long()
    labels_train = y[idx_train]
    labels_syn_np, ncd, _ = generate_labels_syn(labels_train, reduction_rate)
    labels_syn = torch.from_numpy(labels_syn_np)
    d = x.shape[1]
    pge = init_pge_reference_order(d, pge_hid)
    sel = random_init_indices(idx_train, labels_train, ncd)
    feat_init = feat_full[sel]
    rp, col, val = normalize_adj_sparse(adj_full)
    adj_full_norm = csr_to_torch(rp, col, val, adj_full.shape[0])
    g = dict(feat_full=feat_full, adj_full_norm=adj_full_norm, idx_val=torch.as_tensor(idx_val),
             labels_val=labels_full[idx_val], idx_test=torch.as_tensor(idx_test), labels_test=labels_full[idx_test])
    st = GCondState(feat_init, pge, labels_syn, ncd, nclass, lr_feat, lr_adj, lr_model, "SGC", 2)
    init_sgc(d, hidden, nclass, ntrans)   # model constructor draw (gcond.py:38)
    calls, loss_avg, best_val = 0, 0.0, 0.0
    out = dict(loss_avg=[], inter=[], saved=None)
    for it in range(epochs):
        st.new_epoch(init_sgc(d, hidden, nclass, ntrans))
        for _ in range(outer_loop):
            batches = []
            for c in range(nclass):
                batches.append(batch_fn(c, calls))
                calls += 1
            l, _, _, adj_inner = st.outer_iteration(it, batches, feat_full, labels_full, inner_loop)
            loss_avg += float(l)
        loss_avg /= nclass * outer_loop
        if it in checkpoints:
            out["loss_avg"].append(loss_avg)
            adj_s, feat_s = adj_inner.detach().clone(), st.feat_syn.detach().clone()
            res = [run_eval_gcn(feat_s, adj_s, labels_syn, g, hidden, nclass, eval_epochs) for _ in range(run_inter_eval)]
            res = np.array(res).T
            out["inter"].append((res[0].mean(), res[0].std(), res[1].mean(), res[1].std()))
            if res[0].mean() > best_val:
                best_val = res[0].mean()
                out["saved"] = (adj_s, feat_s, labels_syn.clone())
    out["feat_syn"], out["adj_syn"], out["state"] = st.feat_syn.detach(), adj_inner.detach(), st
    adj_s, feat_s, lab_s = out["saved"]
    adj_eval = sparsify(adj_s, threshold)
    accs = []
    for i in range(run_eval):
        seed_everything(seed + i)
        accs.append(run_eval_gcn(feat_s, adj_eval, lab_s, g, hidden, nclass, eval_epochs)[1])
    out["eval_mean"], out["eval_std"] = float(np.mean(accs)), float(np.std(accs))
    return out


# --------------------------------------------------------------------------------------------
# the sibling gradient-matching loops: same __init__ / train_class, different update rule.
# --------------------------------------------------------------------------------------------

SIBLING_RULES = {
    # name: (identity structure, (step_pge, step_feat) as a function of ol, inner steps, inference at checkpoints only)
    "doscond": (False, lambda ol: (True, True), False),            # doscond.py:40-50
    "gcondx": (True, lambda ol: (False, not (ol % 5 < 1)), True),   # gcondx.py:44-66 (optimizer_pge has no gradient: no-op)
    "doscondx": (True, lambda ol: (False, True), False),           # doscondx.py:42-49
}


def reduce_sibling(method, adj_full, x, y, idx_train, nclass, *, reduction_rate, hidden, ntrans, pge_hid, epochs, outer_loop,
                   inner_loop, lr_feat, lr_adj, lr_model, seed, batch_fn, dis_metric="ours"):
    """Restates DosCond / GCondX / DosCondX .reduce without checkpoints (doscond.py:18-67, gcondx.py:18-79,
    doscondx.py:18-64) with the reference's RNG draw order (seed_everything -> PGE ctor -> Random init -> model ctor ->
    model.initialize() per epoch).  `batch_fn(c, call_index)` supplies class c's sampled batch.  Returns the per-iteration
    matching losses, the final X~, the final PGE parameters and the final PGE.inference adjacency (identity for the X
    variants)."""
    identity, rule, run_inner = SIBLING_RULES[method]
    seed_everything(seed)
    feat_full = torch.from_numpy(x).float()
    labels_full = torch.from_numpy(y).long()
    labels_train = y[idx_train]
    labels_syn_np, ncd, _ = generate_labels_syn(labels_train, reduction_rate)
    labels_syn = torch.from_numpy(labels_syn_np)
    d = x.shape[1]
    pge = init_pge_reference_order(d, pge_hid)
    sel = random_init_indices(idx_train, labels_train, ncd)
    st = GCondState(feat_full[sel], pge, labels_syn, ncd, nclass, lr_feat, lr_adj, lr_model, "SGC", 2)
    init_sgc(d, hidden, nclass, ntrans)   # model constructor draw
    calls, losses = 0, []
    for it in range(epochs):
        st.new_epoch(init_sgc(d, hidden, nclass, ntrans))
        for ol in range(outer_loop):
            batches = []
            for c in range(nclass):
                batches.append(batch_fn(c, calls))
                calls += 1
            sp_, sf_ = rule(ol)
            l, _, _ = st.outer_iteration_ex(batches, feat_full, labels_full, inner_loop, step_pge=sp_, step_feat=sf_,
                                            run_inference=False, run_inner=run_inner, identity=identity, dis_metric=dis_metric)
            losses.append(float(l))
    with torch.no_grad():
        if identity:
            adj = torch.eye(len(labels_syn_np))
        else:
            adj = pge_forward(st.feat_syn.detach(), {"lin": [p.detach() for p in st.pge["lin"]],
                                                     "bn": [p.detach() for p in st.pge["bn"]]})
    return dict(losses=losses, feat_syn=st.feat_syn.detach(), adj_syn=adj, state=st, labels_syn=labels_syn_np)
