"""Edge-list structure generator of MSGC (graphslim/condensation/msgc.py:31-39 `self.pge`, :178-222 reset_adj_batch,
:224-236 get_adj_t_syn), forward and backward, composed from libgslim_hip.so calls: gs_gemm_f32 for the MLP's products and the
row kernels of csrc/gs_structgen.hip around them.  No autograd and no torch arithmetic: torch owns the device buffers only.

The `batch_adj` structures over the same n' nodes are laid out as ONE block-diagonal graph of B n' nodes whose rows are grouped
by class (class c: structure 0's nodes of c, structure 1's, ...), which is the layout the matching engine wants
(gs_gcond_io.labels_syn "grouped by class"); a permutation of rows leaves train_class unchanged.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream
from .engine import adam_step, gemm, gemm_tn_splitk

BN_EPS = 1e-5


class BatchLayout:
    """Positions of (structure b, node i) in the class-grouped block-diagonal graph."""

    def __init__(self, y_syn, nclass, batch_size):
        y = np.asarray(y_syn).astype(np.int64)
        n = len(y)
        assert np.all(np.diff(y) >= 0), "y_syn must be class-major (msgc.py:55-61)"
        counts = np.bincount(y, minlength=nclass)
        starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
        self.n, self.B, self.nclass = n, batch_size, nclass
        self.n_big = n * batch_size
        pos = np.empty((batch_size, n), dtype=np.int64)
        for c in range(nclass):
            for b in range(batch_size):
                i = np.arange(starts[c], starts[c] + counts[c])
                pos[b, i] = batch_size * starts[c] + b * counts[c] + (i - starts[c])
        self.pos = pos                                                   # [B, n] -> row of the big graph
        self.node_of_pos = np.empty(self.n_big, dtype=np.int64)
        self.node_of_pos[pos.reshape(-1)] = np.tile(np.arange(n), batch_size)
        self.labels_big = y[self.node_of_pos]
        self.ranges = [(int(batch_size * starts[c]), int(batch_size * (starts[c] + counts[c]))) for c in range(nclass)]
        self.counts = counts


def _segments(keys, n):
    """(ptr [n+1], perm) such that perm[ptr[i]:ptr[i+1]] lists the positions with key i in ascending position order."""
    perm = np.argsort(keys, kind="stable")
    ptr_ = np.concatenate([[0], np.cumsum(np.bincount(keys, minlength=n))])
    return ptr_.astype(np.int32), perm.astype(np.int32)


class EdgeStructureGenerator:
    """theta = [W0 (h, 2d), b0, bn1.weight, bn1.bias, W1 (h, h), b1, bn2.weight, bn2.bias, w2 (1, h), b2] — nn.Sequential's
    parameters() order — kept as one flat device tensor with its gradient and Adam moments."""

    def __init__(self, params, d, rows, cols, batch, layout: BatchLayout, lr, device, hidden=256):
        self.L = _lib.lib()
        self.dev = torch.device(device)
        self.d, self.h, self.lay = d, hidden, layout
        h = hidden
        self.shapes = [(h, 2 * d), (h,), (h,), (h,), (h, h), (h,), (h,), (h,), (1, h), (1,)]
        self.theta = torch.cat([p.reshape(-1).float() for p in params]).to(self.dev).contiguous()
        assert self.theta.numel() == sum(int(np.prod(s)) for s in self.shapes)
        self.grad = torch.zeros_like(self.theta)
        self.m, self.v = torch.zeros_like(self.theta), torch.zeros_like(self.theta)
        self.lr, self.step_count = lr, 0
        rows, cols, batch = (np.asarray(a).astype(np.int64) for a in (rows, cols, batch))
        self.E = E = len(rows)
        i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int32).to(self.dev)
        self.rows, self.cols = i32(rows), i32(cols)
        self.pr, self.pc = i32(layout.pos[batch, rows]), i32(layout.pos[batch, cols])
        self.seg_r = tuple(i32(a) for a in _segments(rows, layout.n))
        self.seg_c = tuple(i32(a) for a in _segments(cols, layout.n))
        self.seg_node = tuple(i32(a) for a in _segments(layout.node_of_pos, layout.n))
        self.node_of_pos = i32(layout.node_of_pos)
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.dev)
        n, nb = layout.n, layout.n_big
        self.U, self.V, self.dU, self.dV = f(n, h), f(n, h), f(n, h), f(n, h)
        self.E0, self.A1, self.E2, self.A2, self.dA = f(E, h), f(E, h), f(E, h), f(E, h), f(E, h)
        self.mu1, self.is1, self.mu2, self.is2 = f(h), f(h), f(h), f(h)
        self.s, self.dlogit = f(E), f(E)
        self.M, self.adj = f(nb, nb), f(nb, nb)
        self.dX = f(n, d)
        self.scratch = torch.empty(self.L.gs_bn_scratch_bytes(h), dtype=torch.uint8, device=self.dev)

    def views(self, flat=None):
        flat = self.theta if flat is None else flat
        out, o = [], 0
        for shp in self.shapes:
            k = int(np.prod(shp))
            out.append(flat[o:o + k].view(shp))
            o += k
        return out

    def forward(self, X):
        """X [n', d] (contiguous) -> raw symmetric structure [B n', B n'] without self loops (msgc.py:225-228)."""
        L, h, d, E, n = self.L, self.h, self.d, self.E, self.lay.n
        w0, b0, g1, be1, w1, b1, g2, be2, w2, b2 = self.views()
        self.X = X
        gemm(X, w0, self.U, n, h, d, (d, 1), (1, 2 * d), h)                               # U = X W0[:, :d]^T
        gemm(X, w0[:, d:], self.V, n, h, d, (d, 1), (1, 2 * d), h)                        # V = X W0[:, d:]^T
        check(L.gs_edge_pair_sum_f32(ptr(self.U), ptr(self.V), h, ptr(b0), ptr(self.rows), ptr(self.cols), E, h, ptr(self.E0), h,
                                     stream()))
        check(L.gs_bn_relu_fwd_f32(ptr(self.E0), h, E, h, ptr(g1), ptr(be1), BN_EPS, ptr(self.A1), h, ptr(self.mu1), ptr(self.is1),
                                   ptr(self.scratch), stream()))
        gemm(self.A1, w1, self.E2, E, h, h, (h, 1), (1, h), h, bias=b1)                   # E2 = A1 W1^T + b1
        check(L.gs_bn_relu_fwd_f32(ptr(self.E2), h, E, h, ptr(g2), ptr(be2), BN_EPS, ptr(self.A2), h, ptr(self.mu2), ptr(self.is2),
                                   ptr(self.scratch), stream()))
        check(L.gs_edge_score_f32(ptr(self.A2), h, ptr(w2), ptr(b2), E, h, ptr(self.s), stream()))
        check(L.gs_edge_scatter_sym_f32(ptr(self.s), ptr(self.pr), ptr(self.pc), E, self.lay.n_big, ptr(self.M), ptr(self.adj),
                                        stream()))
        return self.adj

    def backward(self, d_adj, accumulate_dx=False, param_grads=True):
        """d loss / d adj [B n', B n'] of the LAST forward -> self.grad (flat, parameter order) and self.dX [n', d]
        (accumulate_dx: added to what self.dX already holds — the direct feature gradient folded over the B copies).
        param_grads = False (feature epochs: only X~ is stepped, msgc.py:105-108, and zero_grad() discards the rest unread): the
        weight / bias gradient products are skipped; the BatchNorm sums the input gradient needs are still formed."""
        L, h, d, E, n = self.L, self.h, self.d, self.E, self.lay.n
        w0, b0, g1, be1, w1, b1, g2, be2, w2, b2 = self.views()
        gw0, gb0, gg1, gbe1, gw1, gb1, gg2, gbe2, gw2, gb2 = self.views(self.grad)
        X = self.X
        check(L.gs_edge_score_bwd_f32(ptr(d_adj), ptr(self.pr), ptr(self.pc), ptr(self.s), ptr(w2), E, self.lay.n_big, h,
                                      ptr(self.dlogit), ptr(self.dA), h, stream()))
        if param_grads:
            gemm_tn_splitk(self.A2, h, self.dlogit, 1, gw2, h, 1, E)                      # dw2 = A2^T dlogit
            check(L.gs_colsum_tall_f32(ptr(self.dlogit), 1, E, 1, ptr(gb2), ptr(self.scratch), stream()))
        check(L.gs_bn_relu_bwd_f32(ptr(self.dA), h, ptr(self.A2), h, ptr(self.E2), h, E, h, ptr(self.mu2), ptr(self.is2), ptr(g2),
                                   ptr(self.dA), h, ptr(gg2), ptr(gbe2), ptr(self.scratch), stream()))         # dA := dE2
        if param_grads:
            gemm_tn_splitk(self.dA, h, self.A1, h, gw1, h, h, E)                          # dW1 = dE2^T A1
            check(L.gs_colsum_tall_f32(ptr(self.dA), h, E, h, ptr(gb1), ptr(self.scratch), stream()))
        w1t = w1.t().contiguous()                                                        # (a 256 x 256 copy: B with contiguous k -> row-panel kernel)
        gemm(self.dA, w1t, self.E2, E, h, h, (h, 1), (1, h), h)                           # dA1 = dE2 W1   (E2 is free now)
        check(L.gs_bn_relu_bwd_f32(ptr(self.E2), h, ptr(self.A1), h, ptr(self.E0), h, E, h, ptr(self.mu1), ptr(self.is1), ptr(g1),
                                   ptr(self.E2), h, ptr(gg1), ptr(gbe1), ptr(self.scratch), stream()))         # E2 := dE0
        if param_grads:
            check(L.gs_colsum_tall_f32(ptr(self.E2), h, E, h, ptr(gb0), ptr(self.scratch), stream()))
        for seg, out in ((self.seg_r, self.dU), (self.seg_c, self.dV)):
            check(L.gs_segment_sum_rows_f32(ptr(self.E2), h, ptr(seg[0]), ptr(seg[1]), n, h, ptr(out), h, stream()))
        if param_grads:
            gemm(self.dU, X, gw0, h, d, n, (1, h), (d, 1), 2 * d)                         # dW0[:, :d] = dU^T X
            gemm(self.dV, X, gw0[:, d:], h, d, n, (1, h), (d, 1), 2 * d)                  # dW0[:, d:] = dV^T X
        gemm(self.dU, w0, self.dX, n, d, h, (h, 1), (2 * d, 1), d, beta=1.0 if accumulate_dx else 0.0)   # dX (+)= dU W0[:, :d] + dV W0[:, d:]
        gemm(self.dV, w0[:, d:], self.dX, n, d, h, (h, 1), (2 * d, 1), d, beta=1.0)
        return self.grad, self.dX

    def adam_step(self):
        self.step_count += 1
        adam_step(self.theta, self.grad, self.m, self.v, self.lr, self.step_count)

    # ---- moving between the n' nodes and the B n' rows of the block-diagonal graph ------------------
    def tile(self, X, out):
        """out[p] = X[node(p)]."""
        check(self.L.gs_gather_rows_f32(ptr(X), X.stride(0), ptr(self.node_of_pos), self.lay.n_big, self.d, ptr(out), out.stride(0),
                                        stream()))
        return out

    def fold(self, d_big, out):
        """out[i] = sum over the B rows of node i (the gradient of tile)."""
        check(self.L.gs_segment_sum_rows_f32(ptr(d_big), d_big.stride(0), ptr(self.seg_node[0]), ptr(self.seg_node[1]), self.lay.n,
                                             self.d, ptr(out), out.stride(0), stream()))
        return out

    def batched(self, big):
        """[B n', B n'] block-diagonal (class-grouped rows) -> [B, n', n'] in node order (index shuffling only)."""
        pos = torch.as_tensor(self.lay.pos, device=big.device)
        return big[pos[:, :, None], pos[:, None, :]].contiguous()


# ==================================================================================================================================
# IGNR graphon learner of SGDD (graphslim/models/ignr.py:95-213)
# ==================================================================================================================================

IGNR_ORDER = (["P"] + [f"c{l}_{k}" for l in range(2) for k in ("w", "b", "g", "be")] + ["c2_w", "c2_b"]
              + [f"x{l}_{k}" for l in range(2) for k in ("w", "b", "g", "be")] + ["x2_w", "x2_b", "out_w", "out_b"])
BN_MOMENTUM = 0.1


def pair_grid(n):
    """ignr.py:46-71 get_mgrid(n): coordinates of the ordered pair (i, j), i slow, in [-1, 1]^2 (float32 arithmetic as upstream)."""
    g = np.stack(np.mgrid[:n, :n], axis=-1)[None, ...].astype(np.float32)
    g[0, :, :, 0] = g[0, :, :, 0] / (n - 1)
    g[0, :, :, 1] = g[0, :, :, 1] / (n - 1)
    g -= 0.5
    g *= 2.0
    return torch.from_numpy(g).view(-1, 2).to(torch.float32)


class IGNRGenerator:
    """theta in nn.Module.parameters() order (IGNR_ORDER: the transport plan P, tower net0 = "c" on cat(x_i, x_j), tower net1 = "x" on
    the pair grid, the unused `output` head) as one flat device tensor with gradient and Adam moments; running BatchNorm statistics
    for the eval-mode (inference) pass.  forward / backward are chains of libgslim_hip.so calls."""

    def __init__(self, params, d, n, ep_ratio, lr, device, nfeat=128):
        self.L = _lib.lib()
        self.dev = torch.device(device)
        self.d, self.n, self.h, self.ep, self.lr = d, n, nfeat, float(ep_ratio), lr
        self.E = E = n * n
        h = nfeat
        self.mx = params["P"].shape[0]
        shp = {"P": (self.mx, n), "out_w": (1, h), "out_b": (1,), "c2_w": (1, h), "c2_b": (1,), "x2_w": (1, h), "x2_b": (1,),
               "c0_w": (h, 2 * d), "x0_w": (h, 2), "c1_w": (h, h), "x1_w": (h, h)}
        for t in "cx":
            for l in range(2):
                for k in ("b", "g", "be"):
                    shp[f"{t}{l}_{k}"] = (h,)
        self.shapes = [shp[k] for k in IGNR_ORDER]
        self.theta = torch.cat([params[k].reshape(-1).float() for k in IGNR_ORDER]).to(self.dev).contiguous()
        assert self.theta.numel() == sum(int(np.prod(s)) for s in self.shapes)
        self.grad = torch.zeros_like(self.theta)
        self.m, self.v, self.step_count = torch.zeros_like(self.theta), torch.zeros_like(self.theta), 0
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.dev)
        i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int32).to(self.dev)
        ii, jj = np.repeat(np.arange(n), n), np.tile(np.arange(n), n)
        self.rows, self.cols = i32(ii), i32(jj)
        self.seg_r, self.seg_c = tuple(i32(a) for a in _segments(ii, n)), tuple(i32(a) for a in _segments(jj, n))
        self.grid = pair_grid(n).to(self.dev).contiguous()
        self.U, self.V, self.dU, self.dV = f(n, h), f(n, h), f(n, h), f(n, h)
        names = ["C0p", "C0", "X0p", "X0", "Z0", "C1p", "C1", "X1p", "X1", "Z1", "G1", "G2", "G3"]
        for k in names:
            setattr(self, k, f(E, h))
        self.stat = {k: (f(h), f(h)) for k in ("c0", "c1", "x0", "x1")}           # (mu, istd) of the last pass
        self.run = {k: (torch.zeros(h, device=self.dev), torch.ones(h, device=self.dev)) for k in ("c0", "c1", "x0", "x1")}
        self.e3, self.de3, self.sx, self.sc = f(E), f(E), f(E), f(E)
        self.adj = f(n, n)
        self.dX = f(n, d)
        self.scratch = torch.empty(self.L.gs_bn_scratch_bytes(h), dtype=torch.uint8, device=self.dev)

    def views(self, flat=None):
        flat = self.theta if flat is None else flat
        out, o = {}, 0
        for k, shp in zip(IGNR_ORDER, self.shapes):
            cnt = int(np.prod(shp))
            out[k] = flat[o:o + cnt].view(shp)
            o += cnt
        return out

    def _bn(self, key, Xp, Y, p, train):
        L, h, E = self.L, self.h, self.E
        mu, istd = self.stat[key]
        if train:
            check(L.gs_bn_relu_fwd_f32(ptr(Xp), h, E, h, ptr(p[key + "_g"]), ptr(p[key + "_be"]), BN_EPS, ptr(Y), h, ptr(mu), ptr(istd),
                                       ptr(self.scratch), stream()))
            check(L.gs_bn_running_update_f32(ptr(mu), ptr(istd), E, h, BN_EPS, BN_MOMENTUM, ptr(self.run[key][0]), ptr(self.run[key][1]),
                                             stream()))
        else:
            check(L.gs_bn_relu_eval_f32(ptr(Xp), h, E, h, ptr(p[key + "_g"]), ptr(p[key + "_be"]), BN_EPS, ptr(self.run[key][0]),
                                        ptr(self.run[key][1]), ptr(Y), h, ptr(mu), ptr(istd), stream()))

    def forward(self, X, train=True):
        """X [n', d] -> raw structure [n', n'] (ignr.py:151-177).  train: batch statistics (and the running ones move); else the
        running statistics (PGE.inference's counterpart, ignr.py:211-213)."""
        L, h, d, E, n = self.L, self.h, self.d, self.E, self.n
        p = self.views()
        self.X, self.trained = X, train
        gemm(X, p["c0_w"], self.U, n, h, d, (d, 1), (1, 2 * d), h)
        gemm(X, p["c0_w"][:, d:], self.V, n, h, d, (d, 1), (1, 2 * d), h)
        check(L.gs_edge_pair_sum_f32(ptr(self.U), ptr(self.V), h, ptr(p["c0_b"]), ptr(self.rows), ptr(self.cols), E, h, ptr(self.C0p), h,
                                     stream()))
        self._bn("c0", self.C0p, self.C0, p, train)
        gemm(self.grid, p["x0_w"], self.X0p, E, h, 2, (2, 1), (1, 2), h, bias=p["x0_b"])
        self._bn("x0", self.X0p, self.X0, p, train)
        check(L.gs_mul_f32(ptr(self.X0), ptr(self.C0), E * h, ptr(self.Z0), stream()))
        gemm(self.C0, p["c1_w"], self.C1p, E, h, h, (h, 1), (1, h), h, bias=p["c1_b"])
        self._bn("c1", self.C1p, self.C1, p, train)
        gemm(self.Z0, p["x1_w"], self.X1p, E, h, h, (h, 1), (1, h), h, bias=p["x1_b"])
        self._bn("x1", self.X1p, self.X1, p, train)
        check(L.gs_mul_f32(ptr(self.X1), ptr(self.C1), E * h, ptr(self.Z1), stream()))
        check(L.gs_mix_heads_f32(ptr(self.Z1), ptr(p["x2_w"]), ptr(p["x2_b"]), 1.0 - self.ep, ptr(self.C1), ptr(p["c2_w"]), ptr(p["c2_b"]),
                                 self.ep, E, h, ptr(self.e3), stream()))
        check(L.gs_sym_sigmoid_f32(ptr(self.e3), n, ptr(self.adj), stream()))
        return self.adj

    def _bn_bwd(self, key, dY, Y, Xp, p, g, dX):
        mu, istd = self.stat[key]
        check(self.L.gs_bn_relu_bwd_f32(ptr(dY), self.h, ptr(Y), self.h, ptr(Xp), self.h, self.E, self.h, ptr(mu), ptr(istd),
                                        ptr(p[key + "_g"]), ptr(dX), self.h, ptr(g[key + "_g"]), ptr(g[key + "_be"]), ptr(self.scratch),
                                        stream()))

    def backward(self, d_adj, accumulate_dx=False, param_grads=True):
        """d loss / d adj [n', n'] of the LAST training-mode forward -> self.grad (P's slot is left alone: the transport plan only
        enters the regulariser) and self.dX.  param_grads = False (feature epochs, sgdd.py:80-83): the weight / bias gradient
        products are skipped."""
        assert self.trained, "backward needs a training-mode forward"
        L, h, d, E, n = self.L, self.h, self.d, self.E, self.n
        p, g = self.views(), self.views(self.grad)
        X, S = self.X, self.scratch
        tall = lambda M, cols, out: check(L.gs_colsum_tall_f32(ptr(M), cols, E, cols, ptr(out), ptr(S), stream()))
        check(L.gs_sym_sigmoid_bwd_f32(ptr(d_adj), ptr(self.adj), n, ptr(self.de3), stream()))
        # heads: dZ1 = (1 - ep) de3 (x) w_x2 -> G1 ; dC1 (direct) = ep de3 (x) w_c2 -> G2
        check(L.gs_mix_heads_bwd_f32(ptr(self.de3), ptr(p["x2_w"]), 1.0 - self.ep, ptr(p["c2_w"]), self.ep, E, h, ptr(self.G1), ptr(self.G2),
                                     ptr(self.sx), ptr(self.sc), stream()))
        if param_grads:
            gemm_tn_splitk(self.Z1, h, self.sx, 1, g["x2_w"], h, 1, E)                       # d w_x2 = Z1^T ((1 - ep) de3)
            gemm_tn_splitk(self.C1, h, self.sc, 1, g["c2_w"], h, 1, E)                       # d w_c2 = C1^T (ep de3)
            tall(self.sx, 1, g["x2_b"])
            tall(self.sc, 1, g["c2_b"])
        # coupling 1: Z1 = X1 * C1
        check(L.gs_mul_bwd_f32(ptr(self.G1), ptr(self.X1), ptr(self.C1), E * h, ptr(self.G1), ptr(self.G2), 1, stream()))   # G1 = dX1, G2 = dC1
        # x tower, layer 1
        self._bn_bwd("x1", self.G1, self.X1, self.X1p, p, g, self.G1)                        # G1 = dX1p
        if param_grads:
            gemm_tn_splitk(self.G1, h, self.Z0, h, g["x1_w"], h, h, E)
            tall(self.G1, h, g["x1_b"])
        w = p["x1_w"].t().contiguous()
        gemm(self.G1, w, self.G3, E, h, h, (h, 1), (1, h), h)                                # G3 = dZ0 = dX1p W_x1
        # c tower, layer 1
        self._bn_bwd("c1", self.G2, self.C1, self.C1p, p, g, self.G2)                        # G2 = dC1p
        if param_grads:
            gemm_tn_splitk(self.G2, h, self.C0, h, g["c1_w"], h, h, E)
            tall(self.G2, h, g["c1_b"])
        w = p["c1_w"].t().contiguous()
        gemm(self.G2, w, self.G1, E, h, h, (h, 1), (1, h), h)                                # G1 = dC0 (through layer 1)
        # coupling 0: Z0 = X0 * C0
        check(L.gs_mul_bwd_f32(ptr(self.G3), ptr(self.X0), ptr(self.C0), E * h, ptr(self.G3), ptr(self.G1), 1, stream()))   # G3 = dX0, G1 = dC0
        # x tower, layer 0 (input: the constant grid)
        if param_grads:        # (the grid is a constant input: nothing flows further down this tower)
            self._bn_bwd("x0", self.G3, self.X0, self.X0p, p, g, self.G3)
            gemm_tn_splitk(self.G3, h, self.grid, 2, g["x0_w"], h, 2, E)
            tall(self.G3, h, g["x0_b"])
        # c tower, layer 0: pre-activation = U[i] + V[j] + b
        self._bn_bwd("c0", self.G1, self.C0, self.C0p, p, g, self.G1)
        if param_grads:
            tall(self.G1, h, g["c0_b"])
        for seg, out in ((self.seg_r, self.dU), (self.seg_c, self.dV)):
            check(L.gs_segment_sum_rows_f32(ptr(self.G1), h, ptr(seg[0]), ptr(seg[1]), n, h, ptr(out), h, stream()))
        if param_grads:
            gemm(self.dU, X, g["c0_w"], h, d, n, (1, h), (d, 1), 2 * d)
            gemm(self.dV, X, g["c0_w"][:, d:], h, d, n, (1, h), (d, 1), 2 * d)
        gemm(self.dU, p["c0_w"], self.dX, n, d, h, (h, 1), (2 * d, 1), d, beta=1.0 if accumulate_dx else 0.0)
        gemm(self.dV, p["c0_w"][:, d:], self.dX, n, d, h, (h, 1), (2 * d, 1), d, beta=1.0)
        return self.grad, self.dX

    def adam_step(self):
        self.step_count += 1
        adam_step(self.theta, self.grad, self.m, self.v, self.lr, self.step_count)


# ==================================================================================================================================
# optimal-transport regulariser of SGDD (graphslim/models/ignr.py:13-41,184-209)
# ==================================================================================================================================


class _SymEig:
    """Symmetric eigen-decomposition on the GPU: rocSOLVER's ssyevd, the library call behind torch.linalg.eigh (which upstream calls,
    ignr.py:201) — bound directly with ctypes from the copies PyTorch-ROCm has already loaded."""

    _inst = None

    def __init__(self):
        libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
        self.blas = C.CDLL(os.path.join(libdir, "librocblas.so"))
        self.solver = C.CDLL(os.path.join(libdir, "librocsolver.so"))
        self.handle = C.c_void_p()
        if self.blas.rocblas_create_handle(C.byref(self.handle)) != 0:
            raise RuntimeError("rocblas_create_handle failed")
        self.solver.rocsolver_ssyevd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                 C.c_void_p]
        self.blas.rocblas_set_stream.argtypes = [C.c_void_p, C.c_void_p]

    @classmethod
    def get(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    def eigh(self, A, lam, work, info):
        """In place: A [n, n] symmetric (row-major; its lower triangle is read, like torch.linalg.eigh's default UPLO = 'L') ->
        A[k] = k-th eigenvector (ROWS of the row-major buffer = columns of the column-major matrix the library fills),
        lam ascending."""
        n = A.shape[0]
        if self.blas.rocblas_set_stream(self.handle, stream()) != 0:
            raise RuntimeError("rocblas_set_stream failed")
        EVECT_ORIGINAL, FILL_UPPER = 211, 121          # column-major upper triangle == row-major lower triangle
        rc = self.solver.rocsolver_ssyevd(self.handle, EVECT_ORIGINAL, FILL_UPPER, n, ptr(A), n, ptr(lam), ptr(work), ptr(info))
        if rc != 0:
            raise RuntimeError(f"rocsolver_ssyevd failed (status {rc})")


class TransportRegulariser:
    """opt_loss(adj) = | tr(Ly_inv) n' - 2 sum_k sqrt(eig_k(Ly_inv_rt P^T Lx_inv P Ly_inv_rt)) |  (ignr.py:184-209) and its gradient
    wrt the structure and the transport plan P, in closed form (no autograd): eigen-decompositions by rocSOLVER, products by
    gs_gemm_f32, spectrum functions by csrc/gs_structgen.hip.  Lx_inv [mx, mx] is the constant pseudo-inverse of the real
    adjacency's corner (ignr.py:13-27, computed once on the host)."""

    SV_EPS = 0.009

    def __init__(self, Lx_inv, n, device):
        self.L = _lib.lib()
        self.dev = torch.device(device)
        self.n, self.mx = n, Lx_inv.shape[0]
        Lx_host = torch.as_tensor(np.asarray(Lx_inv, dtype=np.float32))
        self.Lx_inv = Lx_host.to(self.dev).contiguous()
        # torch.svd's pseudo-inverse of a rank-deficient corner is NOT symmetric in its rounding-level components (U and V span the
        # null space with unrelated bases); eigh reads M's lower triangle and its backward hands a symmetric dM to both triangles,
        # so d/dT1 [T1^T Lx_inv T1] = (Lx_inv + Lx_inv^T) T1 dM.  The sum is formed once on the host.
        self.Lx_sym = (Lx_host + Lx_host.t()).to(self.dev).contiguous()
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.dev)
        n, mx = self.n, self.mx
        self.Qr, self.Wr = f(n, n), f(n, n)                    # eigenvectors as ROWS
        self.lam, self.mu, self.work = f(n), f(n), f(n)
        self.info = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.spec = f(3 * n + 1)
        self.R, self.M, self.T = f(n, n), f(n, n), f(n, n)
        self.T1, self.T2, self.dT1 = f(mx, n), f(mx, n), f(mx, n)
        self.dmu = f(n + 1)
        self.value = f(1)
        self.acc = torch.zeros(1, dtype=torch.float64, device=self.dev)
        self.bad = torch.zeros(2, dtype=torch.int64, device=self.dev)      # [non-converged decompositions, entries the range guard touched]
        self.eig = _SymEig.get()

    def forward_backward(self, adj, P, scale, d_adj, dP):
        """value of opt_loss(adj) (device scalar self.value, also added to self.acc) and, for the term `scale * opt_loss` of the
        total loss: d_adj += its gradient wrt adj (n', n'; d_adj = None: left in self.M for a later add_to), dP = its gradient wrt
        P (mx, n')."""
        L, n, mx = self.L, self.n, self.mx
        s = stream()
        # adj = Q diag(lam) Q^T
        self.Qr.copy_(adj)
        self.eig.eigh(self.Qr, self.lam, self.work, self.info)
        self.bad[0] += (self.info[0] != 0)           # rocSOLVER's `info`: off-diagonals that did not converge (device-side, no sync)
        check(L.gs_ot_spectrum_f32(ptr(self.lam), n, self.SV_EPS, ptr(self.spec), s))
        g, gp, hp, tr = self.spec[:n], self.spec[n:2 * n], self.spec[2 * n:3 * n], self.spec[3 * n:]
        # R = Ly_inv_rt = Q diag(g) Q^T = Qr^T diag(g) Qr:  T = diag(g) Qr (rows scaled) -> via columns of Qr^T ... rows: scale_cols on Qr^T
        check(L.gs_scale_cols_f32(ptr(self.Qr.t().contiguous()), ptr(g), n, n, ptr(self.T), s))        # T[i][k] = Q[i][k] g_k
        gemm(self.T, self.Qr, self.R, n, n, n, (n, 1), (n, 1), n)                                        # R = T Qr  (Qr[k][j] = Q[j][k])
        gemm(P, self.R, self.T1, mx, n, n, (n, 1), (n, 1), n)                                            # T1 = P R
        gemm(self.Lx_inv, self.T1, self.T2, mx, n, mx, (mx, 1), (n, 1), n)                               # T2 = Lx_inv T1
        gemm(self.T1, self.T2, self.M, n, n, mx, (1, n), (n, 1), n)                                      # M = T1^T T2
        self.Wr.copy_(self.M)
        # With a rank-deficient corner (Lx_inv entries up to 1e14 measured) M's entries reach 1e15 .. 1e20: inside the fp32 range, but
        # rocSOLVER's tridiagonalisation squares and sums them.  LAPACK's ssyev (behind upstream's CPU eigh) rescales such a matrix
        # before reducing it; the same here, exactly: M / s with s the power of two below max |M|, mu = s mu'.  (Round 2 clamped the
        # entries at 1e15 instead, which changed the value where upstream's is still finite.)  Only NON-FINITE entries — a product
        # that really left the range, where upstream's run turns NaN — are replaced, and counted (self.bad[1]).
        self.bad[1] += (~torch.isfinite(self.Wr)).sum()
        check(L.gs_clamp_finite_f32(ptr(self.Wr), n * n, 3.0e38, s))
        sc = torch.exp2(torch.floor(torch.log2(self.Wr.abs().max().clamp_min(1e-30))))
        self.Wr.div_(sc)
        self.eig.eigh(self.Wr, self.mu, self.work, self.info)
        self.mu.mul_(sc)
        self.bad[0] += (self.info[0] != 0)
        check(L.gs_ot_value_f32(ptr(self.mu), ptr(tr), n, float(scale), ptr(self.dmu), ptr(self.value), ptr(self.acc), s))
        if scale <= 0:
            return self.value
        # dM = W diag(dmu) W^T
        check(L.gs_scale_cols_f32(ptr(self.Wr.t().contiguous()), ptr(self.dmu), n, n, ptr(self.T), s))  # T[i][k] = W[i][k] dmu_k
        gemm(self.T, self.Wr, self.M, n, n, n, (n, 1), (n, 1), n)                                        # M := dM
        gemm(self.Lx_sym, self.T1, self.T2, mx, n, mx, (mx, 1), (n, 1), n)                               # T2 := (Lx_inv + Lx_inv^T) T1
        gemm(self.T2, self.M, self.dT1, mx, n, n, (n, 1), (n, 1), n)                                     # dT1 = (Lx_inv + Lx_inv^T) T1 dM
        gemm(self.dT1, self.R, dP, mx, n, n, (n, 1), (n, 1), n)                                          # dP = dT1 R   (R symmetric)
        gemm(P, self.dT1, self.M, n, n, mx, (1, n), (n, 1), n)                                           # dR = P^T dT1 -> M
        # K = Q^T dR Q = Qr dR Qr^T
        gemm(self.Qr, self.M, self.T, n, n, n, (n, 1), (n, 1), n)                                        # T = Qr dR
        gemm(self.T, self.Qr, self.M, n, n, n, (n, 1), (1, n), n)                                        # M = T Qr^T = K
        check(L.gs_ot_divdiff_f32(ptr(self.M), ptr(self.lam), ptr(g), ptr(gp), ptr(hp), ptr(self.dmu[n:]), n, s))
        # d adj += Q K' Q^T = Qr^T K' Qr
        gemm(self.Qr, self.M, self.T, n, n, n, (1, n), (n, 1), n)                                        # T = Qr^T K'
        gemm(self.T, self.Qr, self.M, n, n, n, (n, 1), (n, 1), n)                                        # M = T Qr = the term's d / d adj
        self.bad[1] += ((~torch.isfinite(self.M)) | (self.M.abs() > 1e30)).sum() + ((~torch.isfinite(dP)) | (dP.abs() > 1e30)).sum()
        check(L.gs_clamp_finite_f32(ptr(self.M), n * n, 1e30, s))
        check(L.gs_clamp_finite_f32(ptr(dP), dP.numel(), 1e30, s))
        if d_adj is not None:
            self.add_to(d_adj)
        return self.value

    def read_health(self):
        """(eigen-decompositions rocSOLVER reported as not converged, entries the fp32 range guard replaced) since the last call —
        one host sync; the host loop reads it once per epoch and warns."""
        v = [int(x) for x in self.bad.tolist()]
        self.bad.zero_()
        return v[0], v[1]

    def add_to(self, d_adj):
        """d_adj += the term's gradient wrt the structure (left in self.M by forward_backward): one slab, beta = 1."""
        n = self.n
        check(self.L.gs_reduce_slabs_f32(ptr(d_adj), ptr(self.M), n * n, 1, n * n, 1.0, stream()))


def pinv_corner(adj_corner):
    """ignr.py:13-27 mx_inv on the host, once per job (like the scipy normalisation of the real adjacency): pseudo-inverse through
    torch.svd in float32 (the call upstream makes); the singular values EQUAL to the smallest one are zeroed when that one is below
    0.009.  (Only those: a corner with several vanishing singular values gets their rounding-level reciprocals, upstream too.)"""
    a = torch.as_tensor(np.asarray(adj_corner.todense() if hasattr(adj_corner, "todense") else adj_corner, dtype=np.float32))
    U, D, V = torch.svd(a)
    d_min = torch.min(D)
    if d_min < 0.009:
        d1 = torch.zeros_like(D)
        d1[D > d_min] = 1 / D[D > d_min]
    else:
        d1 = 1 / D
    return (U @ d1.diag() @ V.t()).numpy()
