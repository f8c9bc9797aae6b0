"""Host-side handle of the HIP GCond engine (gs_gcond_* in include/gslim.h).

torch is used only to own device memory and the stream; all arithmetic happens in libgslim_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import Blocks, GCondConfig, GCondIO, GemmDesc, check, ptr, stream

METRICS = {"ours": 0, "mse": 1, "cos": 2}

# Operand form of the five large contractions.  The host mirror's default is the EXACT three-way bf16 split
# ("bf16x6": every fp32 operand = h + m + l with 8 + 8 + 8 significant bits, six bf16 MFMA products, fp32 accumulation — fp32-level
# error, measured <= the fp32 MFMA's own against fp64, every parity gate of the fp32 path incl. the headline-shape accuracy gate);
# "f32" selects the fp32 matrix pipe (1/16 of the bf16 rate).  The C library's own default stays 0 = fp32.
_OPERAND_MODE_NAMES = ("f32", "bf16x3", "bf16", "bf16x6")
DEFAULT_OPERAND_MODE = os.environ.get("GS_OPERAND_MODE", "bf16x6")
# the next iteration's training forward beside the inner steps (gs_gcond_set_hoist_forward; library default off, this mirror: on)
DEFAULT_HOIST_FORWARD = int(os.environ.get("GS_HOIST_FWD", "1"))
if DEFAULT_OPERAND_MODE not in _OPERAND_MODE_NAMES:      # (an unknown value used to reach ctypes as a str and fail there, opaquely)
    raise ValueError(f"GS_OPERAND_MODE={DEFAULT_OPERAND_MODE!r}: expected one of {', '.join(_OPERAND_MODE_NAMES)}")


# ------------------------------------------------------------------------------------------------
# thin stateless wrappers (used by the tests and by the evaluation model)
# ------------------------------------------------------------------------------------------------


def gemm_desc(A, B, Cout, M, N, K, sA, sB, ldc, **kw):
    """A filled GemmDesc (element strides sA=(sAm,sAk), sB=(sBk,sBn); kw = other descriptor fields) + the tensors it points to."""
    d = GemmDesc()
    d.A, d.B, d.C = ptr(A), ptr(B), ptr(Cout)
    d.M, d.N, d.K = M, N, K
    d.sAm, d.sAk = sA
    d.sBk, d.sBn = sB
    d.ldc = ldc
    d.nb1 = d.nb2 = 1
    d.alpha = 1.0
    keep = [A, B, Cout]
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            v = ptr(v)
        setattr(d, k, v)
    return d, keep


def gemm(A, B, Cout, M, N, K, sA, sB, ldc, **kw):
    """C = alpha * A*B (+...) with explicit element strides sA=(sAm,sAk), sB=(sBk,sBn); kw = other desc fields."""
    d, _keep = gemm_desc(A, B, Cout, M, N, K, sA, sB, ldc, **kw)
    check(_lib.lib().gs_gemm_f32(C.byref(d), stream()))


def gemm_tn_splitk(A, lda, B, ldb, out, M, N, K, target_workgroups=768):
    """out[M, N] = A[K, M]^T B[K, N] for a tall K (gradients of a full-graph layer: K = number of nodes): K is split into
    slabs that fill the chip, each slab a partial product, summed in slab order (reproducible, no float atomics)."""
    tiles = -(-M // 128) * -(-N // 128)
    nsplit = max(1, min(target_workgroups // max(tiles, 1), -(-K // 256)))
    kchunk = -(-(-(-K // nsplit)) // 16) * 16
    nsplit = -(-K // kchunk)
    if nsplit == 1:
        gemm(A, B, out, M, N, K, (1, lda), (ldb, 1), N)
        return out
    slabs = torch.empty(nsplit, M, N, device=out.device, dtype=torch.float32)
    gemm(A, B, slabs, M, N, K, (1, lda), (ldb, 1), N, nb2=nsplit, kchunk=kchunk, sCb2=M * N)
    check(_lib.lib().gs_reduce_slabs_f32(ptr(out), ptr(slabs), M * N, nsplit, M * N, 0.0, stream()))
    return out


def gemm_group(descs):
    """Mutually independent products [(GemmDesc, keepalive), ...] in one launch when they are all latency-bound."""
    arr = (GemmDesc * len(descs))(*[d for d, _ in descs])
    check(_lib.lib().gs_gemm_group_f32(arr, len(descs), stream()))


def spmm_csr(rowptr, col, val, X, F=None, out=None):
    n = rowptr.numel() - 1
    F = X.shape[1] if F is None else F
    if out is None:
        out = torch.empty(n, F, device=X.device, dtype=torch.float32)
    check(_lib.lib().gs_spmm_csr_f32(ptr(rowptr), ptr(col), ptr(val), n, ptr(X), X.stride(0), F, ptr(out), out.stride(0),
                                     stream()))
    return out


def gather_rows(X, idx, F=None):
    F = X.shape[1] if F is None else F
    out = torch.empty(idx.numel(), F, device=X.device, dtype=torch.float32)
    check(_lib.lib().gs_gather_rows_f32(ptr(X), X.stride(0), ptr(idx), idx.numel(), F, ptr(out), out.stride(0), stream()))
    return out


def dense_gcn_norm_fwd(adj):
    n = adj.shape[0]
    out = torch.empty_like(adj)
    r = torch.empty(2 * n, device=adj.device, dtype=torch.float32)
    check(_lib.lib().gs_dense_gcn_norm_fwd(ptr(adj), n, ptr(out), ptr(r), stream()))
    return out, r


def dense_gcn_norm_bwd(adj, r, d_adj_norm):
    n = adj.shape[0]
    out = torch.empty_like(adj)
    check(_lib.lib().gs_dense_gcn_norm_bwd(ptr(adj), ptr(r), ptr(d_adj_norm), n, ptr(out), stream()))
    return out


def adam_step(p, g, m, v, lr, step, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
    check(_lib.lib().gs_adam_step_f32(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, betas[0], betas[1], eps, weight_decay,
                                      step, stream()))


# ------------------------------------------------------------------------------------------------
# sampled blocks
# ------------------------------------------------------------------------------------------------


class BlockSet:
    """Device arrays of a gs_blocks (class-major, fixed capacity).  fmid: fan-outs of the middle hops (nlayers >= 3)."""

    def __init__(self, nclass, bcap, f1, f2, device, fmid=()):
        self.C, self.bcap, self.f1, self.f2, self.fmid = nclass, bcap, f1, f2, tuple(int(f) for f in fmid)
        self.cap1 = bcap * (1 + f1)
        i32 = dict(device=device, dtype=torch.int32)
        f32 = dict(device=device, dtype=torch.float32)
        self.n_seed = torch.zeros(nclass, **i32)
        self.n1 = torch.zeros(nclass, **i32)
        self.nid1 = torch.zeros(nclass, self.cap1, **i32)
        self.col1 = torch.zeros(nclass, bcap, f1, **i32)
        self.val1 = torch.zeros(nclass, bcap, f1, **f32)
        self.tptr1 = torch.zeros(nclass, self.cap1 + 1, **i32)
        self.tedge1 = torch.zeros(nclass, bcap * f1, **i32)
        self.capm, capp = [], self.cap1
        for m, fm in enumerate(self.fmid):      # middle hop m: frontier m - 1 (capp rows) <- frontier m (capc nodes)
            capc = capp * (1 + fm)
            self.capm.append(capc)
            setattr(self, f"nm{m}", torch.zeros(nclass, **i32))
            setattr(self, f"nidm{m}", torch.zeros(nclass, capc, **i32))
            setattr(self, f"colm{m}", torch.zeros(nclass, capp, fm, **i32))
            setattr(self, f"valm{m}", torch.zeros(nclass, capp, fm, **f32))
            setattr(self, f"tptrm{m}", torch.zeros(nclass, capc + 1, **i32))
            setattr(self, f"tedgem{m}", torch.zeros(nclass, capp * fm, **i32))
            setattr(self, f"scrm{m}", torch.zeros(nclass, capc + 1, **i32))
            capp = capc
        self.cap_last = capp                    # rows of the outermost block per class
        self.col0 = torch.zeros(nclass, capp, f2, **i32)
        self.val0 = torch.zeros(nclass, capp, f2, **f32)

    BASE_FIELDS = ("n_seed", "n1", "nid1", "col1", "val1", "tptr1", "tedge1", "col0", "val0")
    MID_FIELDS = ("nm", "nidm", "colm", "valm", "tptrm", "tedgem")

    @property
    def FIELDS(self):
        return self.BASE_FIELDS + tuple(f"{k}{m}" for m in range(len(self.fmid)) for k in self.MID_FIELDS)

    @property
    def n_last(self):
        """[C] live rows of the outermost block (the last frontier's size)."""
        return getattr(self, f"nm{len(self.fmid) - 1}") if self.fmid else self.n1

    def struct(self):
        b = Blocks()
        b.C, b.bcap, b.f1, b.f2, b.cap1 = self.C, self.bcap, self.f1, self.f2, self.cap1
        for f in self.BASE_FIELDS:
            setattr(b, f, ptr(getattr(self, f)))
        b.nmid = len(self.fmid)
        for m, fm in enumerate(self.fmid):
            b.fm[m], b.capm[m] = fm, self.capm[m]
            for k in self.MID_FIELDS + ("scrm",):
                getattr(b, k)[m] = ptr(getattr(self, f"{k}{m}"))
        return b

    @classmethod
    def from_struct(cls, b, device):
        """Copies of the engine-owned block arrays."""
        out = cls(b.C, b.bcap, b.f1, b.f2, device, fmid=[b.fm[m] for m in range(b.nmid)])
        for f in cls.BASE_FIELDS:
            t = getattr(out, f)
            t.copy_(device_view(getattr(b, f), t.numel(), t.dtype, device).reshape(t.shape))
        for m in range(b.nmid):
            for k in cls.MID_FIELDS:
                t = getattr(out, f"{k}{m}")
                t.copy_(device_view(getattr(b, k)[m], t.numel(), t.dtype, device).reshape(t.shape))
        return out

    @classmethod
    def from_batches(cls, batches, bcap, f1, f2, device, fmid=()):
        """Parity mode: lay out reference-style batches (batch_size, n_id, [outermost block, .., innermost block]) with CSR blocks
        (rowptr, col, val, shape) into the engine's ELL layout.  Host-side data shuffling only."""
        ncls = len(batches)
        out = cls(ncls, bcap, f1, f2, "cpu", fmid=fmid)
        nm = len(out.fmid)

        def csc(rp, col, rows, fan, n_src, tptr, tedge):
            parents = [[] for _ in range(n_src)]
            for r in range(rows):
                for j, e in enumerate(range(int(rp[r]), int(rp[r + 1]))):
                    parents[int(col[e])].append(r * fan + j)
            pos = 0
            for l in range(n_src):
                tptr[l] = pos
                for sl in sorted(parents[l]):
                    tedge[pos] = sl
                    pos += 1
            tptr[n_src:] = pos

        for c, (bs, n_id, blocks) in enumerate(batches):
            assert len(blocks) == 2 + nm
            rp1, c1, v1, (B, n1) = blocks[-1]
            assert B == bs and B <= bcap and n1 <= out.cap1
            out.n_seed[c], out.n1[c] = B, n1
            out.nid1[c, :n1] = torch.from_numpy(np.asarray(n_id[:n1]).astype(np.int32))
            for b in range(B):
                lo, hi = int(rp1[b]), int(rp1[b + 1])
                assert hi - lo <= f1
                for j, e in enumerate(range(lo, hi)):
                    out.col1[c, b, j] = int(c1[e])
                    out.val1[c, b, j] = float(v1[e])
            csc(rp1, c1, B, f1, n1, out.tptr1[c], out.tedge1[c])
            rows = n1
            for m in range(nm):      # middle hop m is blocks[-2 - m]: rows = frontier m - 1, columns = frontier m
                rpm, cm, vm, (r_, ncur) = blocks[-2 - m]
                fm = out.fmid[m]
                assert r_ == rows and ncur <= out.capm[m]
                getattr(out, f"nm{m}")[c] = ncur
                getattr(out, f"nidm{m}")[c, :ncur] = torch.from_numpy(np.asarray(n_id[:ncur]).astype(np.int32))
                colm, valm = getattr(out, f"colm{m}"), getattr(out, f"valm{m}")
                for r in range(rows):
                    lo, hi = int(rpm[r]), int(rpm[r + 1])
                    assert hi - lo <= fm
                    for j, e in enumerate(range(lo, hi)):
                        colm[c, r, j] = int(cm[e])
                        valm[c, r, j] = float(vm[e])
                csc(rpm, cm, rows, fm, ncur, getattr(out, f"tptrm{m}")[c], getattr(out, f"tedgem{m}")[c])
                rows = ncur
            rp0, c0, v0, (r_, n2) = blocks[0]
            assert r_ == rows
            for l in range(rows):
                lo, hi = int(rp0[l]), int(rp0[l + 1])
                assert hi - lo <= f2
                for j, e in enumerate(range(lo, hi)):
                    out.col0[c, l, j] = int(n_id[int(c0[e])])
                    out.val0[c, l, j] = float(v0[e])
        for f in out.FIELDS + tuple(f"scrm{m}" for m in range(nm)):
            setattr(out, f, getattr(out, f).to(device))
        return out


class _DevArray:
    """__cuda_array_interface__ carrier so torch can view library-owned device memory without a copy."""

    def __init__(self, p, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(p), False), "version": 2}


def device_view(p, count, dtype, device):
    if count == 0 or not p:
        return torch.empty(0, dtype=dtype, device=device)
    typestr = {torch.float32: "<f4", torch.int32: "<i4", torch.float64: "<f8", torch.int64: "<i8"}[dtype]
    return torch.as_tensor(_DevArray(p, count, typestr), device=device)


def sample_blocks(rowptr, col, val, n_nodes, cls_ptr, cls_nodes, marker, blocks: BlockSet, seed: int, step: int):
    b = blocks.struct()
    check(_lib.lib().gs_sample_blocks(ptr(rowptr), ptr(col), ptr(val), n_nodes, ptr(cls_ptr), ptr(cls_nodes), ptr(marker),
                                      C.byref(b), seed, step, stream()))


# ------------------------------------------------------------------------------------------------
# the engine
# ------------------------------------------------------------------------------------------------


@dataclass
class EngineShapes:
    n_nodes: int
    d: int
    nclass: int
    n_syn: int
    ntrans: int
    hidden: int
    pge_hidden: int
    bcap: int = 256
    f1: int = 10
    f2: int = 5
    model: str = "SGC"     # condense model: SGC (ntrans transforms) or GCN (two layers)
    pge_chunks: int = 1    # 5 = the chunked PGE of reddit r >= 0.01 (parametrized_adj.py:41-55)
    with_bn: bool = False  # SGC ntrans = 2 with an (always eval-mode) BatchNorm1d between the transforms (models/sgc.py:27-33)
    fmid: tuple = ()       # `--nlayers` >= 3 (SGC): fan-outs of the hops BETWEEN f1 (from the seeds) and f2 (outermost), e.g. (10,) for
                           # loader.py:206's [15, 10, 5] with f1 = 15, f2 = 5; nlayers = 2 + len(fmid) propagations on both sides
    dropout: float = 0.0   # models/sgc.py:42 (SGC, two transforms): masks by dropout_keep below; 0 in every shipped configuration

    @property
    def nlayers(self):
        return 2 + len(self.fmid)


_M64 = (1 << 64) - 1


def _mix64(z):
    """splitmix64's finaliser on uint64 arrays (csrc/gs_kernels.h gs_mix64)."""
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def dropout_keep(seed, iteration, kind, index, rows, hidden, p, layer=0):
    """The engine's dropout mask (csrc/gs_kernels.h gs_drop_mult4), restated: bool [len(rows), hidden], True = kept.  kind "real" (index
    = class, rows = the batch's node ids), "syn" (index = class, rows = 0 .. n'-1) or "inner" (index = inner step, rows = 0 .. n'-1);
    layer = 0, 1, .. : the hidden layer the mask multiplies (the GCN condensation model has nlayers - 1 of them, csrc/gs_gcnk.inc gk_kind;
    SGC with two transforms has one).  The multiplier is keep / (1 - p) in fp32.  Parity tests hand these masks to the oracle
    (oracle.gcond_oracle.DROPOUT)."""
    kind_id = {"real": 0, "syn": 1, "inner": 2}[kind] + 3 * int(layer)
    thr = min(max(int((1.0 - float(p)) * 65536.0 + 0.5), 1), 65535)
    with np.errstate(over="ignore"):
        key0 = _mix64(np.array([int(seed) & _M64], dtype=np.uint64))[0] + np.uint64((int(iteration) << 24) & _M64) + np.uint64(kind_id << 20)
        key = _mix64(np.array([key0 + np.uint64(int(index))], dtype=np.uint64))[0]
        rows = np.asarray(rows, dtype=np.uint64)[:, None]
        col = np.arange(hidden, dtype=np.uint64)[None, :]
        word = _mix64(key + (rows << np.uint64(8)) + (col >> np.uint64(2)))
        u = (word >> (np.uint64(16) * (col & np.uint64(3)))) & np.uint64(0xFFFF)
    return u < np.uint64(thr)


def model_param_count(d, hidden, nclass, ntrans):
    """models/sgc.py:16-35: one transform [d, C]; else [d, h], ntrans - 2 x [h, h], [h, C] (each with its bias)."""
    if ntrans == 1:
        return d * nclass + nclass
    return d * hidden + hidden + (ntrans - 2) * (hidden * hidden + hidden) + hidden * nclass + nclass


def pge_param_count(d, ph):
    return ph * 2 * d + ph + ph * ph + ph + ph + 1 + 4 * ph


class GCondEngine:
    """Owns the device tensors and the gs_gcond handle.  See gs_gcond_* in include/gslim.h."""

    def __init__(self, shapes: EngineShapes, X, csr, cls_ptr, cls_nodes, labels_syn, syn_ranges, *, lr_feat, lr_adj,
                 lr_model, inner_loop, dis_metric="ours", seed=1, external_blocks=False, device="cuda:0"):
        if dis_metric not in METRICS:
            raise SystemExit("DC error: unknown distance function")     # condensation/utils.py:63
        L = _lib.lib()
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("graphslim_amd: the GCond engine needs a GPU device (there is no CPU fallback)")
        if self.dev.index is None:
            self.dev = torch.device("cuda", torch.cuda.current_device())
        # every later call launches on torch's CURRENT stream: that must be a stream of this engine's device
        if torch.cuda.current_device() != self.dev.index:
            raise RuntimeError(f"GCondEngine(device={self.dev}): make it the current device first (torch.cuda.set_device / "
                               f"`with torch.cuda.device(...)`); current device is cuda:{torch.cuda.current_device()}")
        s = self.shapes = shapes
        dev = self.dev
        d = s.d
        self.ldx = (d + 3) // 4 * 4          # 16-byte aligned feature rows for the gather kernels
        Xp = torch.zeros(s.n_nodes, self.ldx, dtype=torch.float32, device=dev)
        Xp[:, :d] = X.to(dev, torch.float32)
        self.X = Xp
        i32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.int32).to(dev)
        if csr is not None:
            rp, col, val = csr
            self.rowptr, self.col = i32(rp), i32(col)
            self.val = torch.as_tensor(np.asarray(val), dtype=torch.float32).to(dev)
            self.cls_ptr, self.cls_nodes = i32(cls_ptr), i32(cls_nodes)
        else:
            self.rowptr = self.col = self.val = self.cls_ptr = self.cls_nodes = None
        self.labels_syn = i32(labels_syn)
        self.syn_lo = i32([r[0] for r in syn_ranges])
        self.syn_hi = i32([r[1] for r in syn_ranges])
        # (GCN: one GraphConvolution per sampled hop, models/gcn.py:8-23 — `nlayers` layers in the SGC layout)
        self.P = model_param_count(d, s.hidden, s.nclass, max(2, s.nlayers) if s.model == "GCN" else s.ntrans) + (2 * s.hidden if s.with_bn else 0)
        self.Ppge = pge_param_count(d, s.pge_hidden)
        self.feat_syn = torch.zeros(s.n_syn, d, dtype=torch.float32, device=dev)
        self.pge_params = torch.zeros(self.Ppge, dtype=torch.float32, device=dev)
        self.model_params = torch.zeros(self.P, dtype=torch.float32, device=dev)
        self.adj_syn = torch.zeros(s.n_syn, s.n_syn, dtype=torch.float32, device=dev)
        cfg = GCondConfig(s.n_nodes, d, self.ldx, s.nclass, s.n_syn, s.ntrans, s.hidden, s.pge_hidden, s.bcap, s.f1, s.f2,
                          METRICS[dis_metric], inner_loop, lr_feat, lr_adj, lr_model, seed, 1 if external_blocks else 0,
                          {"SGC": 0, "GCN": 1}[s.model], int(s.pge_chunks), 1 if s.with_bn else 0, s.nlayers,
                          (C.c_int * 3)(*(list(s.fmid) + [0] * (3 - len(s.fmid)))), float(s.dropout))
        nbytes = C.c_size_t()
        check(L.gs_gcond_workspace_bytes(C.byref(cfg), C.byref(nbytes)))
        self.workspace_bytes = nbytes.value
        self.workspace = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        io = GCondIO(ptr(self.X), ptr(self.rowptr), ptr(self.col), ptr(self.val), ptr(self.cls_ptr), ptr(self.cls_nodes),
                     ptr(self.labels_syn), ptr(self.syn_lo), ptr(self.syn_hi), ptr(self.feat_syn), ptr(self.pge_params),
                     ptr(self.model_params), ptr(self.adj_syn), ptr(self.workspace), nbytes.value)
        h = C.c_void_p()
        with torch.cuda.device(dev):
            check(L.gs_gcond_create(C.byref(cfg), C.byref(io), stream(), C.byref(h)))
        self.h = h
        self.L = L
        self.set_operand_mode(DEFAULT_OPERAND_MODE)
        self.set_hoist_forward(DEFAULT_HOIST_FORWARD)

    def _s(self):
        """torch's current stream — which must belong to this engine's device (an engine built for cuda:1 must
        not be driven from cuda:0's stream)."""
        if torch.cuda.current_device() != self.dev.index:
            raise RuntimeError(f"GCondEngine on {self.dev} called while cuda:{torch.cuda.current_device()} is current")
        return stream()

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            try:
                torch.cuda.synchronize(self.dev)
            except Exception:
                pass
            self.L.gs_gcond_destroy(h)
            self.h = None

    # ---- parameter packing (nn.Module.parameters() order) ------------------------------------------
    def set_model(self, params, keep_optimizer=False):
        """Fresh theta.  keep_optimizer: the model's Adam moments and step count carry over (MSGC builds its model optimiser once,
        msgc.py:90, and re-draws the weights in place every epoch); default: a fresh optimiser per epoch (gcond.py:42-45)."""
        if keep_optimizer and self.shapes.with_bn:
            raise NotImplementedError("keep_optimizer with with_bn")
        self.model_params.copy_(torch.cat([p.reshape(-1).float() for p in params]).to(self.dev))
        if not keep_optimizer:
            check(self.L.gs_gcond_new_epoch(self.h, self._s()))

    def get_model(self):
        s, d = self.shapes, self.shapes.d
        if s.ntrans == 1 and s.model == "SGC":
            shp = [(d, s.nclass), (s.nclass,)]
        else:
            mid = [(s.hidden, s.hidden), (s.hidden,)] * (s.ntrans - 2 if s.model == "SGC" else max(2, s.nlayers) - 2)
            shp = [(d, s.hidden), (s.hidden,)] + mid + [(s.hidden, s.nclass), (s.nclass,)]
        if s.with_bn:
            shp += [(s.hidden,), (s.hidden,)]          # bns.0.weight, bns.0.bias
        return _split(self.model_params, shp)

    def pge_shapes(self):
        ph, d = self.shapes.pge_hidden, self.shapes.d
        return [(ph, 2 * d), (ph,), (ph, ph), (ph,), (1, ph), (1,), (ph,), (ph,), (ph,), (ph,)]

    def set_pge(self, params):
        self.L.gs_gcond_invalidate_forward(self.h)      # (first: waits for a forward the last outer step issued ahead — it reads these)
        self.pge_params.copy_(torch.cat([p.reshape(-1).float() for p in params]).to(self.dev))
        self.L.gs_gcond_invalidate_forward(self.h)

    def get_pge(self):
        return _split(self.pge_params, self.pge_shapes())

    def set_feat(self, x):
        self.L.gs_gcond_invalidate_forward(self.h)      # (first: waits for a forward the last outer step issued ahead — it reads X~)
        self.feat_syn.copy_(x.to(self.dev, torch.float32))
        self.L.gs_gcond_invalidate_forward(self.h)

    def set_forward_reuse(self, on):
        check(self.L.gs_gcond_set_forward_reuse(self.h, int(on)))

    def set_dropout_iter(self, it):
        """The iteration counter inside the dropout masks' keys (gs_gcond_set_dropout_iter): resume, or pin the masks of staged calls."""
        check(self.L.gs_gcond_set_dropout_iter(self.h, int(it)))

    def dropout_iter(self):
        v = C.c_uint64()
        check(self.L.gs_gcond_get_dropout_iter(self.h, C.byref(v)))
        return int(v.value)

    def forward_cached(self):
        return bool(self.L.gs_gcond_forward_cached(self.h))

    def set_hoist_forward(self, on):
        """gs_gcond_set_hoist_forward: the library's default is OFF (a C caller may write X~ / theta_PGE between steps); this host
        mirror opts in for every engine (DEFAULT_HOIST_FORWARD) because set_feat / set_pge bracket their writes."""
        check(self.L.gs_gcond_set_hoist_forward(self.h, int(on)))

    def forward_hoisted(self):
        """The last outer step issued the next iteration's training forward ahead and the next call will take it up."""
        return bool(self.L.gs_gcond_forward_hoisted(self.h))

    def adopt_cached_forward(self):
        check(self.L.gs_gcond_adopt_cached_forward(self.h, self._s()))

    # ---- stages -------------------------------------------------------------------------------------
    def set_blocks(self, blocks: BlockSet):
        b = blocks.struct()
        check(self.L.gs_gcond_set_blocks(self.h, C.byref(b), self._s()))
        self._blocks_keepalive = blocks

    def get_blocks(self) -> BlockSet:
        b = Blocks()
        check(self.L.gs_gcond_get_blocks(self.h, C.byref(b)))
        return BlockSet.from_struct(b, self.dev)

    def stage_sample(self):
        check(self.L.gs_gcond_stage_sample(self.h, self._s()))

    def stage_real_grads(self):
        check(self.L.gs_gcond_stage_real_grads(self.h, self._s()))

    def probe(self, on=True):
        check(self.L.gs_gcond_probe(self.h, int(on)))

    def probe_read(self):
        """(summed duration [ms], launches) of the PGE layer-2 GEMM since probe(True)."""
        t, k = C.c_double(), C.c_long()
        check(self.L.gs_gcond_probe_read(self.h, C.byref(t), C.byref(k)))
        return t.value, k.value

    def probe_read_spans(self):
        """(summed device-side span [ms], launches) of the same launches: first workgroup start .. last workgroup end, i.e. the
        kernel's duration as a kernel trace reports it (the event bracket of probe_read also contains queueing behind the other
        stream's resident workgroups).  Synchronises the device."""
        t, k = C.c_double(), C.c_long()
        check(self.L.gs_gcond_probe_read_spans(self.h, C.byref(t), C.byref(k)))
        return t.value, k.value

    def probe_read_pass(self, inference):
        """One pass of the probed product (False: training forward, True: inference forward): (event bracket [ms], launches, device span
        [ms], launches that recorded a span) since probe(True)."""
        tb, kb, ts, ks = C.c_double(), C.c_long(), C.c_double(), C.c_long()
        check(self.L.gs_gcond_probe_read_pass(self.h, int(bool(inference)), C.byref(tb), C.byref(kb), C.byref(ts), C.byref(ks)))
        return tb.value, kb.value, ts.value, ks.value

    def probe_read_spmm(self):
        """(summed HIP-event bracket [ms], launches) of the hop-2 gather SpMM of the real side since probe(True)."""
        t, k = C.c_double(), C.c_long()
        check(self.L.gs_gcond_probe_read_spmm(self.h, C.byref(t), C.byref(k)))
        return t.value, k.value

    def stage_real_async(self):
        check(self.L.gs_gcond_stage_real_async(self.h, self._s()))

    def stage_pge_forward(self, inference=False):
        check(self.L.gs_gcond_stage_pge_forward(self.h, int(inference), self._s()))

    def stage_syn_match(self):
        check(self.L.gs_gcond_stage_syn_match(self.h, self._s()))

    def stage_pge_backward(self, need_param_grads=True, need_feat_grad=True):
        check(self.L.gs_gcond_stage_pge_backward(self.h, int(need_param_grads), int(need_feat_grad), self._s()))

    def stage_update(self, epoch):
        check(self.L.gs_gcond_stage_update(self.h, epoch, self._s()))

    def stage_inner(self):
        check(self.L.gs_gcond_stage_inner(self.h, self._s()))

    def outer_step(self, epoch):
        check(self.L.gs_gcond_outer_step(self.h, epoch, self._s()))

    def outer_step_ex(self, step_pge, step_feat, run_inference, run_inner):
        check(self.L.gs_gcond_outer_step_ex(self.h, int(step_pge), int(step_feat), int(run_inference), int(run_inner), self._s()))

    def outer_step_sharded(self, epoch, comm):
        """This rank's share of a sharded outer iteration, exchanges included: ONE library call (gs_gcond_outer_step_sharded).
        comm: graphslim_amd.parallel.Comm."""
        check(self.L.gs_gcond_outer_step_sharded(self.h, epoch, comm.h, self._s()))

    def outer_step_sharded_ex(self, step_pge, step_feat, run_inference, run_inner, comm):
        check(self.L.gs_gcond_outer_step_sharded_ex(self.h, int(step_pge), int(step_feat), int(run_inference), int(run_inner), comm.h,
                                                    self._s()))

    def set_identity_structure(self, on=True):
        check(self.L.gs_gcond_set_identity_structure(self.h, int(on), self._s()))

    def set_structure(self, adj, inference=False):
        """Structure supplied by the caller (gs_gcond_set_structure): raw symmetric [n_syn, n_syn] without self loops."""
        assert adj.is_contiguous() and tuple(adj.shape) == (self.shapes.n_syn, self.shapes.n_syn)
        check(self.L.gs_gcond_set_structure(self.h, ptr(adj), int(inference), self._s()))

    def set_block_structure(self, pos=None):
        """The synthetic graph as B disconnected blocks (gs_gcond_set_block_structure): pos [B, n_block] rows of the blocks' nodes
        (numpy / tensor); None returns to dense products.  Index array and scratch are kept alive by the engine object."""
        if pos is None:
            check(self.L.gs_gcond_set_block_structure(self.h, None, 0, 0, None))
            self._blk = None
            return
        pos = torch.as_tensor(np.ascontiguousarray(pos), dtype=torch.int32).to(self.dev)
        B, nb = pos.shape
        scratch = torch.empty(3 * B * nb * nb + 2 * B * nb * max(self.shapes.d, self.shapes.hidden), dtype=torch.float32, device=self.dev)
        self._blk = (pos, scratch)
        check(self.L.gs_gcond_set_block_structure(self.h, ptr(pos), B, nb, ptr(scratch)))

    OPERAND_MODES = {"f32": 0, "bf16x3": 1, "bf16": 2, "bf16x6": 3}

    def set_operand_mode(self, mode):
        """Operand form of the row-panel products (gs_gcond_set_operand_mode): 'f32' (fp32 MFMA), 'bf16x6' (DEFAULT_OPERAND_MODE: exact
        three-way bf16 split, 6 bf16 MFMAs, fp32 accumulate: fp32-level error), 'bf16x3' (two-way split, 3 MFMAs), 'bf16'."""
        if mode not in self.OPERAND_MODES and mode not in self.OPERAND_MODES.values():
            raise ValueError(f"operand mode {mode!r}: expected one of {', '.join(self.OPERAND_MODES)}")
        check(self.L.gs_gcond_set_operand_mode(self.h, self.OPERAND_MODES.get(mode, mode)))
        self.operand_mode = mode

    def set_full_neighbour(self, S_full=None, rs_full=None):
        """Full-neighbour mode for SGC with ntrans = 1 (gs_gcond_set_full_neighbour): S_full [N, ld >= d] = A^ A^ X and
        rs_full [N] = A^ A^ 1 (graphslim_amd/condensation/full_neighbour.py); None returns to neighbour sampling.  The tensors are
        kept alive by the engine object."""
        self._full_nb = (S_full, rs_full)
        if S_full is None:
            check(self.L.gs_gcond_set_full_neighbour(self.h, None, 0, None))
        else:
            assert S_full.is_contiguous() and rs_full.is_contiguous() and S_full.shape[0] == self.shapes.n_nodes
            check(self.L.gs_gcond_set_full_neighbour(self.h, ptr(S_full), S_full.stride(0), ptr(rs_full)))

    def set_overlap(self, on):
        check(self.L.gs_gcond_set_overlap(self.h, int(on)))

    # ---- sharded (multi-GPU) driving: see graphslim_amd/parallel.py ------------------------------------
    def set_shard(self, c_lo, c_hi, r_lo, r_hi):
        check(self.L.gs_gcond_set_shard(self.h, c_lo, c_hi, r_lo, r_hi))

    def pge_forward_part(self, inference, part):
        check(self.L.gs_gcond_stage_pge_forward_part(self.h, int(inference), part, self._s()))

    def pge_backward_part(self, need_param_grads, need_feat_grad, part):
        check(self.L.gs_gcond_stage_pge_backward_part(self.h, int(need_param_grads), int(need_feat_grad), part, self._s()))

    def syn_finish(self):
        check(self.L.gs_gcond_stage_syn_finish(self.h, self._s()))

    def inner_steps(self):
        check(self.L.gs_gcond_stage_inner_steps(self.h, self._s()))

    def view(self, name, dtype=torch.float32):
        """Zero-copy torch view of a named engine buffer (the all-reduce buffers are exchanged in place)."""
        p, cnt = C.c_void_p(), C.c_long()
        check(self.L.gs_gcond_buffer(self.h, name.encode(), C.byref(p), C.byref(cnt)))
        return device_view(p.value, cnt.value, dtype, self.dev)

    def read_loss(self):
        s, k = C.c_double(), C.c_long()
        check(self.L.gs_gcond_read_loss(self.h, C.byref(s), C.byref(k), self._s()))
        return s.value, k.value

    def buffer(self, name, shape=None):
        """Copy of a named engine scratch buffer as a CPU tensor (tests)."""
        p, cnt = C.c_void_p(), C.c_long()
        check(self.L.gs_gcond_buffer(self.h, name.encode(), C.byref(p), C.byref(cnt)))
        out = device_view(p.value, cnt.value, torch.float32, self.dev).cpu()
        return out.reshape(shape) if shape is not None else out


def _split(flat, shapes):
    out, o = [], 0
    for shp in shapes:
        n = int(np.prod(shp))
        out.append(flat[o:o + n].reshape(shp))
        o += n
    return out
