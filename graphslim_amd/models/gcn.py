"""Evaluation GCN of the reference (graphslim/models/gcn.py + models/base.py + models/layers.py:17-51), with every
tensor operation executed by libgslim_hip.so (gs_gemm_f32 on the matrix cores, gs_spmm_csr_f32 for the full-graph
aggregation, gs_adam_step_f32, ...).  This module is host-side orchestration: same constructor and method names,
argument meaning and return values as the reference's `GCN`.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp
import torch

from .. import _lib, host
from .._lib import check, ptr, stream
from ..engine import adam_step, dense_gcn_norm_fwd, gemm, gemm_tn_splitk, spmm_csr


def _i32(a, device):
    return torch.as_tensor(np.asarray(a), dtype=torch.int32).to(device)


def _pad4(n):
    return (n + 3) // 4 * 4


class DeviceGraph:
    """One evaluation graph resident in HBM, prepared once per (split, device):

      * the normalised adjacency A^ = D^-1/2 (A + I) D^-1/2 (normalize_adj_tensor(sparse=True), utils.py:403-413) as CSR,
        restricted to this rank's row block [lo, hi) when the evaluation is row-sharded;
      * S = A^ X for those rows.  X never changes while a model is evaluated, so layer 1 of the reference's GCN,
        A^ (X W1) + b1, is computed as S W1 + b1: the width-h full-graph SpMM of every epoch becomes one width-d SpMM per
        graph (same value up to fp32 re-association);
      * for every target set (validation / test rows) the CSR of just those rows, so the last aggregation of an epoch
        only produces the rows whose accuracy is read.
    """

    def __init__(self, feat, adj, device, lo=0, hi=None):
        rp, col, val = host.normalize_adj_sparse(adj)
        self.n = len(rp) - 1
        self.lo, self.hi = lo, self.n if hi is None else hi
        self.device = device
        self._norm = sp.csr_matrix((val, col, rp), shape=(self.n, self.n))
        d = feat.shape[1]
        X = torch.zeros(self.n, _pad4(d), device=device)
        X[:, :d] = feat.to(device, torch.float32)
        loc = self._norm[self.lo:self.hi]
        csr = (_i32(loc.indptr, device), _i32(loc.indices, device), torch.as_tensor(loc.data, dtype=torch.float32).to(device))
        self.S = spmm_csr(*csr, X)                    # [hi - lo, pad4(d)]
        self._targets = {None: (csr, self.hi - self.lo)}

    def target_t(self, idx):
        """CSR of (A^[idx, :])^T  [n x n_target]: the backward product of the last aggregation when the loss is taken on the
        rows `idx` (None = all rows) of the training graph.  Cached next to the forward CSR."""
        key = ("T", None if idx is None else id(idx))
        if key not in self._targets:
            ids = np.arange(self.n) if idx is None else np.asarray(idx.cpu() if isinstance(idx, torch.Tensor) else idx)
            sub_t = self._norm[ids].T.tocsr()
            self._targets[key] = ((_i32(sub_t.indptr, self.device), _i32(sub_t.indices, self.device),
                                   torch.as_tensor(sub_t.data, dtype=torch.float32).to(self.device)), len(ids), None, idx)
        return self._targets[key][0]

    def target(self, idx):
        """CSR of the target rows `idx` (global ids; None = all rows) that fall into this rank's block, cached."""
        key = None if idx is None else id(idx)
        if key not in self._targets:
            ids = np.asarray(idx.cpu() if isinstance(idx, torch.Tensor) else idx)
            mine = ids[(ids >= self.lo) & (ids < self.hi)]
            sub = self._norm[mine]
            csr = (_i32(sub.indptr, self.device), _i32(sub.indices, self.device),
                   torch.as_tensor(sub.data, dtype=torch.float32).to(self.device))
            # the idx object is kept in the entry so that its id() cannot be recycled while the cache lives
            self._targets[key] = (csr, len(mine), torch.as_tensor((ids >= self.lo) & (ids < self.hi)), idx)
        return self._targets[key]


_warned_bn = False


def evaluation_args(args):
    """args for the evaluation model.  with_bn reaches it through the shared args upstream, where the evaluation GCN then fails
    on the condensed graph (golden G17): a run with --with_bn could never pass its first checkpoint.  Here the condensation
    model honours with_bn and the evaluation GCN is built WITHOUT BatchNorm, announced once."""
    global _warned_bn
    if not getattr(args, "with_bn", False):
        return args
    if not _warned_bn:
        _warned_bn = True
        msg = ("graphslim_amd: with_bn applies to the condensation model only; the evaluation GCN runs without BatchNorm "
               "(upstream's raises RuntimeError on a condensed graph, models/base.py:65)")
        log = getattr(args, "logger", None)
        (log.warning if log is not None else print)(msg)
    import copy
    out = copy.copy(args)
    out.with_bn = False
    return out


class GCN:
    """GCN of `nlayers` GraphConvolution layers (models/gcn.py:8-23; 2 in every shipped configuration, 3 .. 5 with `--nlayers`): per
    layer adj @ (x @ W) + b, ReLU between, log_softmax (models/base.py:51-78)."""

    def __init__(self, nfeat, nhid, nclass, args, mode="train"):
        self.nlayers = int(getattr(args, "nlayers", 2))
        if self.nlayers < 2 or self.nlayers > 5:
            raise NotImplementedError("graphslim_amd.GCN: nlayers must be 2 .. 5")
        if getattr(args, "with_bn", False):
            # The reference's evaluation GCN cannot run with_bn on a condensed graph: GraphConvolution.forward returns a 3-D
            # tensor [1, n', hidden] for a dense adjacency (layers.py:41-47) and BatchNorm1d(hidden) rejects it — RuntimeError in
            # the first fit_with_val(reduced=True) forward (golden G17 records it).  with_bn is built for the condensation model
            # (graphslim_amd/condensation/gcond.py); the evaluation callers pass evaluation_args(args) (below), which drops it.
            raise NotImplementedError("graphslim_amd.GCN: with_bn has no reference behaviour to match in the evaluation model — "
                                      "upstream raises RuntimeError ('running_mean should contain n_syn elements not hidden', "
                                      "models/base.py:65 after layers.py:41-47) on the condensed graph")
        if not str(args.device).startswith("cuda"):
            raise RuntimeError("graphslim_amd has no CPU path: run with a GPU (-G 0)")
        self.args = args
        self.device = torch.device(args.device)
        self.nfeat, self.nhid, self.nclass = nfeat, nhid, nclass
        self.lr = args.lr
        self.weight_decay = args.weight_decay
        self.dropout = args.dropout
        if mode == "eval":                      # models/base.py:28-30
            self.dropout = 0
            self.weight_decay = 5e-4
        if self.dropout != 0:
            raise NotImplementedError("graphslim_amd.GCN: dropout > 0 is outside the built path")
        self.metric = args.metric
        self.shapes = [(nfeat, nhid), (nhid,)] + [(nhid, nhid), (nhid,)] * (self.nlayers - 2) + [(nhid, nclass), (nclass,)]
        self.sizes = [int(np.prod(shp)) for shp in self.shapes]
        self.P = torch.zeros(sum(self.sizes), dtype=torch.float32, device=self.device)
        self.ldc = _pad4(nclass)
        # row-block sharding of the full-graph passes (parallel.RowShard); a single-GPU run owns every row
        from ..parallel import RowShard
        self.shard = getattr(args, "eval_row_shard", None) or RowShard()
        self.initialize()

    # -- parameters -----------------------------------------------------------------------------------
    def initialize(self):
        """models/base.py:44-49 -> GraphConvolution.reset_parameters (layers.py:29-33): torch-CPU draws, W then b."""
        ps = host.init_gcn_params(self.nfeat, self.nhid, self.nclass, self.nlayers)
        self.P.copy_(torch.cat([p.reshape(-1) for p in ps]).to(self.device))

    def _views(self, flat):
        o, out = 0, []
        for n, shp in zip(self.sizes, self.shapes):
            out.append(flat[o:o + n].view(shp))
            o += n
        return out

    def parameters(self):
        return self._views(self.P)

    def state_dict(self):
        return {"flat": self.P.clone()}

    def load_state_dict(self, sd):
        self.P.copy_(sd["flat"])

    def eval(self):
        return self

    def train(self):
        return self

    def to(self, device):
        return self

    # -- forward passes -------------------------------------------------------------------------------
    # -- nlayers >= 3: the same passes with the hidden layers in between (each one full-graph aggregation of width h) -----
    def _hidden_layers(self, G, keep=False):
        """H_1 = relu(S W1 + b1), H_l = relu(A^ (H_(l-1) W_l) + b_l) for l = 2 .. nlayers - 1 on the full graph; returns the
        last one (keep: the list of all of them, for the backward pass)."""
        if self.shard.world > 1:
            raise NotImplementedError("graphslim_amd.GCN: nlayers != 2 is not row-sharded")
        ps = self.parameters()
        d, h, dev = self.nfeat, self.nhid, self.device
        N = G.S.shape[0]
        H = torch.empty(N, h, device=dev)
        gemm(G.S, ps[0], H, N, h, d, (G.S.stride(0), 1), (h, 1), h, bias=ps[1], relu=1)
        Hs = [H]
        csr = G.target(None)[0]
        for l in range(1, self.nlayers - 1):
            T = torch.empty(N, h, device=dev)
            gemm(H, ps[2 * l], T, N, h, h, (h, 1), (h, 1), h)
            H = torch.zeros(N, h, device=dev)
            spmm_csr(*csr, T, F=h, out=H)
            check(_lib.lib().gs_bias_act_f32(ptr(H), h, ptr(ps[2 * l + 1]), N, h, 1, stream()))
            Hs.append(H)
        return Hs if keep else H

    def _train_step_dense_deep(self, X, A, labels_i32, opt):
        """_train_step_dense for nlayers >= 3: H_l = relu(A^ (H_(l-1) W_l) + b_l), Z = A^ (H_(K-1) W_K) + b_K and its backward."""
        ps = self.parameters()
        K, n, Cn, dev = self.nlayers, X.shape[0], self.nclass, self.device
        L = _lib.lib()
        Hs, Hin, win = [], X, self.nfeat
        for l in range(K):
            W, b = ps[2 * l], ps[2 * l + 1]
            wout = W.shape[1]
            T = torch.empty(n, wout, device=dev)
            gemm(Hin, W, T, n, wout, win, (win, 1), (wout, 1), wout)
            H = torch.empty(n, wout, device=dev)
            gemm(A, T, H, n, wout, n, (n, 1), (wout, 1), wout, bias=b, relu=1 if l < K - 1 else 0)
            Hs.append(H)
            Hin, win = H, wout
        Z = Hs[-1]
        G = torch.empty(n, Cn, device=dev)
        lossr = torch.empty(n, device=dev)
        check(L.gs_softmax_nll_grad_f32(ptr(Z), Cn, ptr(labels_i32), n, Cn, ptr(G), ptr(lossr), stream()))
        g = torch.empty_like(self.P)
        gs = self._views(g)
        dZ = G                                                           # d loss / d pre-activation of layer l (masked below l = K)
        for l in range(K - 1, -1, -1):
            W = ps[2 * l]
            win, wout = W.shape
            check(L.gs_colsum_f32(ptr(dZ), wout, n, wout, ptr(gs[2 * l + 1]), stream()))
            dT = torch.empty(n, wout, device=dev)
            gemm(A, dZ, dT, n, wout, n, (1, n), (wout, 1), wout)          # A^T dZ
            Hprev = X if l == 0 else Hs[l - 1]
            gemm(Hprev, dT, gs[2 * l], win, wout, n, (1, win), (wout, 1), wout)      # H_(l-1)^T dT
            if l > 0:
                dH = torch.empty(n, win, device=dev)
                gemm(dT, W, dH, n, win, wout, (wout, 1), (1, wout), win, mask=Hprev, ldmask=win)   # (dT W^T) * [H_(l-1) > 0]
                dZ = dH
        opt["step"] += 1
        adam_step(self.P, g, opt["m"], opt["v"], opt["lr"], opt["step"], weight_decay=self.weight_decay)
        return lossr

    def _train_step_graph_deep(self, G, idx, labels_i32, opt):
        """_train_step_graph for nlayers >= 3 (S = A^ X cached: layer 1 needs no aggregation; every hidden layer one of width h)."""
        ps = self.parameters()
        K, d, h, Cn, ldc, dev = self.nlayers, self.nfeat, self.nhid, self.nclass, self.ldc, self.device
        L = _lib.lib()
        N = G.S.shape[0]
        Hs = self._hidden_layers(G, keep=True)
        H = Hs[-1]
        T2 = torch.zeros(N, ldc, device=dev)
        gemm(H, ps[-2], T2, N, Cn, h, (h, 1), (Cn, 1), ldc)
        csr, nt = G.target(idx)[0], G.target(idx)[1]
        Z = torch.zeros(nt, ldc, device=dev)
        spmm_csr(*csr, T2, F=ldc, out=Z)
        check(L.gs_bias_act_f32(ptr(Z), ldc, ptr(ps[-1]), nt, Cn, 0, stream()))
        Gz = torch.empty(nt, Cn, device=dev)
        lossr = torch.empty(nt, device=dev)
        check(L.gs_softmax_nll_grad_f32(ptr(Z), ldc, ptr(labels_i32), nt, Cn, ptr(Gz), ptr(lossr), stream()))
        g = torch.empty_like(self.P)
        gs = self._views(g)
        ones = self._ones(max(N, nt))
        gemm_tn_splitk(ones, 1, Gz, Cn, gs[-1], 1, Cn, nt)
        dT = spmm_csr(*G.target_t(idx), Gz, F=Cn)                         # (A^[idx, :])^T G   [N, C]
        gemm_tn_splitk(H, h, dT, Cn, gs[-2], h, Cn, N)
        dH = torch.empty(N, h, device=dev)
        gemm(dT, ps[-2], dH, N, h, Cn, (Cn, 1), (1, Cn), h, mask=H, ldmask=h)
        full_t = G.target_t(None)
        for l in range(K - 2, 0, -1):                                     # hidden layers K - 1 .. 2 (0-based l: parameters 2 l, 2 l + 1)
            gemm_tn_splitk(ones, 1, dH, h, gs[2 * l + 1], 1, h, N)
            dTl = spmm_csr(*full_t, dH, F=h)                              # A^^T dZ_l
            gemm_tn_splitk(Hs[l - 1], h, dTl, h, gs[2 * l], h, h, N)
            dHn = torch.empty(N, h, device=dev)
            gemm(dTl, ps[2 * l], dHn, N, h, h, (h, 1), (1, h), h, mask=Hs[l - 1], ldmask=h)
            dH = dHn
        gemm_tn_splitk(ones, 1, dH, h, gs[1], 1, h, N)
        gemm_tn_splitk(G.S, G.S.stride(0), dH, h, gs[0], d, h, N)
        opt["step"] += 1
        adam_step(self.P, g, opt["m"], opt["v"], opt["lr"], opt["step"], weight_decay=self.weight_decay)
        return lossr

    def _forward_graph(self, G, idx=None):
        """Full graph: Z = A^ (relu(S W1 + b1) W2) + b2 on the target rows `idx` (None = every row); S = A^ X is cached
        in `G`.  Returns logits [n_target, ldc] (row stride ldc >= nclass)."""
        if self.nlayers != 2:
            ps = self.parameters()
            h, Cn, ldc, dev = self.nhid, self.nclass, self.ldc, self.device
            H = self._hidden_layers(G)
            T2 = torch.zeros(H.shape[0], ldc, device=dev)
            gemm(H, ps[-2], T2, H.shape[0], Cn, h, (h, 1), (Cn, 1), ldc)
            tgt = G.target(idx)
            Z = torch.zeros(tgt[1], ldc, device=dev)
            if tgt[1] > 0:
                spmm_csr(*tgt[0], T2, F=ldc, out=Z)
                check(_lib.lib().gs_bias_act_f32(ptr(Z), ldc, ptr(ps[-1]), tgt[1], Cn, 0, stream()))
            return Z
        W1, b1, W2, b2 = self.parameters()
        d, h, Cn, ldc = self.nfeat, self.nhid, self.nclass, self.ldc
        dev = self.device
        nl = G.S.shape[0]
        H = torch.empty(nl, h, device=dev)
        gemm(G.S, W1, H, nl, h, d, (G.S.stride(0), 1), (h, 1), h, bias=b1, relu=1)
        T2 = torch.zeros(nl, ldc, device=dev)
        gemm(H, W2, T2, nl, Cn, h, (h, 1), (Cn, 1), ldc)
        T2 = self.shard.all_gather_rows(T2, G.n, G.blocks)          # identity on one GPU
        tgt = G.target(idx)
        Z = torch.zeros(tgt[1], ldc, device=dev)
        if tgt[1] > 0:
            spmm_csr(*tgt[0], T2, F=ldc, out=Z)
            check(_lib.lib().gs_bias_act_f32(ptr(Z), ldc, ptr(b2), tgt[1], Cn, 0, stream()))
        return Z

    def _train_step_dense(self, X, A, labels_i32, opt):
        """One optimisation step on the condensed graph (models/base.py:156-161) with dense A^ [n, n]."""
        W1, b1, W2, b2 = self.parameters()
        n, d, h, Cn = X.shape[0], self.nfeat, self.nhid, self.nclass
        dev = self.device
        L = _lib.lib()
        T1 = torch.empty(n, h, device=dev)
        gemm(X, W1, T1, n, h, d, (d, 1), (h, 1), h)
        H = torch.empty(n, h, device=dev)
        gemm(A, T1, H, n, h, n, (n, 1), (h, 1), h, bias=b1, relu=1)
        T2 = torch.empty(n, Cn, device=dev)
        gemm(H, W2, T2, n, Cn, h, (h, 1), (Cn, 1), Cn)
        Z = torch.empty(n, Cn, device=dev)
        gemm(A, T2, Z, n, Cn, n, (n, 1), (Cn, 1), Cn, bias=b2)
        G = torch.empty(n, Cn, device=dev)
        lossr = torch.empty(n, device=dev)
        check(L.gs_softmax_nll_grad_f32(ptr(Z), Cn, ptr(labels_i32), n, Cn, ptr(G), ptr(lossr), stream()))
        g = torch.empty_like(self.P)
        gW1, gb1, gW2, gb2 = self._views(g)
        check(L.gs_colsum_f32(ptr(G), Cn, n, Cn, ptr(gb2), stream()))
        dT2 = torch.empty(n, Cn, device=dev)
        gemm(A, G, dT2, n, Cn, n, (1, n), (Cn, 1), Cn)                     # A^T G
        gemm(H, dT2, gW2, h, Cn, n, (1, h), (Cn, 1), Cn)                   # H^T dT2
        dH = torch.empty(n, h, device=dev)
        gemm(dT2, W2, dH, n, h, Cn, (Cn, 1), (1, Cn), h, mask=H, ldmask=h)  # (dT2 W2^T) * [H > 0]
        check(L.gs_colsum_f32(ptr(dH), h, n, h, ptr(gb1), stream()))
        dT1 = torch.empty(n, h, device=dev)
        gemm(A, dH, dT1, n, h, n, (1, n), (h, 1), h)                       # A^T dH
        gemm(X, dT1, gW1, d, h, n, (1, d), (h, 1), h)                      # X^T dT1
        opt["step"] += 1
        adam_step(self.P, g, opt["m"], opt["v"], opt["lr"], opt["step"], weight_decay=self.weight_decay)
        return lossr

    def _train_step_graph(self, G, idx, labels_i32, opt):
        """One optimisation step on an ORIGINAL graph (models/base.py:156-161 with reduced=False): loss on the rows `idx`
        of Z = A^ (relu(S W1 + b1) W2) + b2, S = A^ X cached in `G` — so neither pass needs a width-d or width-h SpMM."""
        W1, b1, W2, b2 = self.parameters()
        d, h, Cn, ldc = self.nfeat, self.nhid, self.nclass, self.ldc
        dev = self.device
        L = _lib.lib()
        N = G.S.shape[0]
        H = torch.empty(N, h, device=dev)
        gemm(G.S, W1, H, N, h, d, (G.S.stride(0), 1), (h, 1), h, bias=b1, relu=1)
        T2 = torch.zeros(N, ldc, device=dev)
        gemm(H, W2, T2, N, Cn, h, (h, 1), (Cn, 1), ldc)
        csr, nt = G.target(idx)[0], G.target(idx)[1]
        Z = torch.zeros(nt, ldc, device=dev)
        spmm_csr(*csr, T2, F=ldc, out=Z)
        check(L.gs_bias_act_f32(ptr(Z), ldc, ptr(b2), nt, Cn, 0, stream()))
        Gz = torch.empty(nt, Cn, device=dev)
        lossr = torch.empty(nt, device=dev)
        check(L.gs_softmax_nll_grad_f32(ptr(Z), ldc, ptr(labels_i32), nt, Cn, ptr(Gz), ptr(lossr), stream()))
        g = torch.empty_like(self.P)
        gW1, gb1, gW2, gb2 = self._views(g)
        ones = self._ones(max(N, nt))
        gemm_tn_splitk(ones, 1, Gz, Cn, gb2, 1, Cn, nt)                       # 1^T G
        dT2 = spmm_csr(*G.target_t(idx), Gz, F=Cn)                          # (A^[idx, :])^T G   [N, C]
        gemm_tn_splitk(H, h, dT2, Cn, gW2, h, Cn, N)                          # H^T dT2   (K = N nodes: split-K slabs)
        dH = torch.empty(N, h, device=dev)
        gemm(dT2, W2, dH, N, h, Cn, (Cn, 1), (1, Cn), h, mask=H, ldmask=h)   # (dT2 W2^T) * [H > 0]
        gemm_tn_splitk(ones, 1, dH, h, gb1, 1, h, N)                          # 1^T dH
        gemm_tn_splitk(G.S, G.S.stride(0), dH, h, gW1, d, h, N)              # S^T dH
        opt["step"] += 1
        adam_step(self.P, g, opt["m"], opt["v"], opt["lr"], opt["step"], weight_decay=self.weight_decay)
        return lossr

    def _ones(self, n):
        if getattr(self, "_ones_buf", None) is None or self._ones_buf.numel() < n:
            self._ones_buf = torch.ones(n, device=self.device)
        return self._ones_buf

    def _new_opt(self, lr):
        return {"m": torch.zeros_like(self.P), "v": torch.zeros_like(self.P), "step": 0, "lr": lr}

    def _accuracy(self, Z, labels_i32, n_total):
        """Correct predictions among this rank's target rows (rows of Z), summed over ranks, over the global count."""
        cnt = torch.zeros(1, dtype=torch.int32, device=self.device)
        n = labels_i32.numel()
        if n > 0:
            check(_lib.lib().gs_count_correct(ptr(Z), Z.stride(0), self.nclass, None, ptr(labels_i32), n, ptr(cnt), stream()))
        return int(self.shard.all_reduce_sum(cnt).item()) / n_total

    # -- reference API ----------------------------------------------------------------------------------
    def _graph_for(self, data, which):
        """DeviceGraph of a split (cached on the data object per device and row block)."""
        cache = data.__dict__.setdefault("_gs_dev", {})
        feat, adj = getattr(data, f"feat_{which}"), getattr(data, f"adj_{which}")
        lo, hi = self.shard.row_block(adj)
        key = (which, str(self.device), lo, hi)
        if key not in cache:
            cache[key] = DeviceGraph(feat, adj, self.device, lo, hi)
            cache[key].blocks = self.shard.blocks(adj) if self.shard.world > 1 else [(0, adj.shape[0])]
        return cache[key]

    def _targets(self, G, idx, labels):
        """(idx, int32 labels of the target rows this rank owns, global target count)."""
        labels = torch.as_tensor(labels)
        n_total = labels.numel()
        if self.shard.world > 1:
            if idx is None:
                labels = labels[G.lo:G.hi]
            else:
                labels = labels[G.target(idx)[2]]
        return idx, labels.to(self.device).to(torch.int32), n_total

    def fit_with_val(self, data, train_iters=600, verbose=False, normadj=True, setting="trans", reduced=False,
                     final_output=False, best_val=None, **kwargs):
        """models/base.py:80-187 for the reduced-graph case: train on (adj_syn, feat_syn, labels_syn), validate on the
        full graph (trans) / validation graph (ind) every epoch, keep the best-validation weights."""
        dev = self.device
        self.initialize()
        if reduced:
            adj = torch.as_tensor(data.adj_syn).to(dev, torch.float32).contiguous()
            feat = torch.as_tensor(data.feat_syn).to(dev, torch.float32).contiguous()
            labels = torch.as_tensor(data.labels_syn).to(dev).to(torch.int32)
            if adj.dim() == 3:
                # MSGC hands over B structures over the same nodes (msgc.py:126-130): GraphConvolution broadcasts the batch and
                # flattens it (layers.py:41-47), i.e. one block-diagonal graph of B n' nodes with the features repeated;
                # normalize_adj_tensor normalises structure by structure (utils.py:416-422) = the blocks of that graph
                nb = adj.shape[0]
                adj = torch.block_diag(*adj).contiguous()
                feat = feat.repeat(nb, 1).contiguous()
            adj, _ = dense_gcn_norm_fwd(adj)          # base.py:112 normalises the (dense) synthetic adjacency unconditionally
        else:
            # base.py:91-96: the original graph — transductive: full graph, loss on output[idx_train]; inductive: train graph
            if self.shard.world > 1:
                raise NotImplementedError("graphslim_amd.GCN.fit_with_val(reduced=False) is not row-sharded")
            Gt = self._graph_for(data, "full" if setting == "trans" else "train")
            idx_loss = data.idx_train if setting == "trans" else None
            labels = torch.as_tensor(data.labels_train).to(dev).to(torch.int32)
        if setting == "ind":
            G = self._graph_for(data, "val")
            idx_val, labels_val, n_val = self._targets(G, None, data.labels_val)
        else:
            G = self._graph_for(data, "full")
            idx_val, labels_val, n_val = self._targets(G, data.idx_val, data.labels_val)
        best_acc = 0 if best_val is None else best_val
        weights = None
        opt = self._new_opt(self.lr)
        for i in range(train_iters):
            if i == train_iters // 2 and self.lr > 0.001:           # base.py:153-154
                opt = self._new_opt(self.lr * 0.1)
            if self.nlayers != 2:
                lossr = (self._train_step_dense_deep(feat, adj, labels, opt) if reduced
                         else self._train_step_graph_deep(Gt, idx_loss, labels, opt))
            else:
                lossr = self._train_step_dense(feat, adj, labels, opt) if reduced else self._train_step_graph(Gt, idx_loss, labels, opt)
            if verbose and i % 100 == 0:
                print("Epoch {}, training loss: {}".format(i, float(lossr.mean())))
            acc_val = self._accuracy(self._forward_graph(G, idx_val), labels_val, n_val)
            if acc_val > best_acc:
                best_acc = acc_val
                weights = self.state_dict()
        if final_output:
            return
        if weights is not None:
            self.load_state_dict(weights)
        return float(best_acc)

    @torch.no_grad()
    def test(self, data, setting="trans", verbose=False):
        """models/base.py:190-215."""
        if setting == "ind":
            G = self._graph_for(data, "test")
            idx, labels_test, n = self._targets(G, None, data.labels_test)
        else:
            G = self._graph_for(data, "full")
            idx, labels_test, n = self._targets(G, data.idx_test, data.labels_test)
        acc = self._accuracy(self._forward_graph(G, idx), labels_test, n)
        if verbose:
            print("Test set results:", "accuracy= {:.4f}".format(acc))
        return acc

    @torch.no_grad()
    def predict(self, features=None, adj=None, normadj=True, output_layer_features=False):
        """models/base.py:218-225: log-probabilities [N, C] for a (features, scipy adjacency) pair (this rank's row block
        when the evaluation is row-sharded)."""
        if not sp.issparse(adj):
            raise NotImplementedError("graphslim_amd.GCN.predict expects a scipy sparse adjacency")
        lo, hi = self.shard.row_block(adj)
        G = DeviceGraph(torch.as_tensor(features), adj, self.device, lo, hi)
        G.blocks = self.shard.blocks(adj) if self.shard.world > 1 else [(0, adj.shape[0])]
        Z = self._forward_graph(G)
        out = torch.empty(Z.shape[0], self.nclass, device=self.device)
        check(_lib.lib().gs_log_softmax_f32(ptr(Z), Z.stride(0), Z.shape[0], self.nclass, ptr(out), self.nclass, stream()))
        return out
