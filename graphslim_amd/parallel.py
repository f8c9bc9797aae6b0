"""Multi-GPU GCond: one process per GPU; the exchanges are issued INSIDE libgslim_hip.so (gs_comm: RCCL over xGMI), one
library call per rank and outer iteration (gs_gcond_outer_step_sharded).  torch.distributed only bootstraps (rendezvous,
broadcast of the RCCL unique id) and carries the test transports (gloo).

What shards: inside one outer iteration theta, X~ and A^ are fixed, so the C class terms of
train_class (gcond_base.py:210-239) are independent summands — classes are partitioned over ranks (sampler, real-side
gradients, per-class synthetic terms) — and the PGE works on the n' x n' pair grid row by row — pair rows are
partitioned over ranks.  What is exchanged: sums of per-rank partials only, five to eight all-reduce(sum) of <= 1.1 MB
per outer iteration (BN-2 column sums, pair scores, [dX~ | d adj | loss_c], BN-2 backward sums, BN-1 marginals, and
dW1/db1 in PGE epochs).  After each exchange every rank holds identical values, applies the identical fused Adam step
and runs the (tiny) inner SGC steps replicated, so the replicas never drift.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np
import torch

from ._lib import ALLREDUCE_FN, check, lib, ptr, stream


def split_even(n, world):
    """[lo, hi) of `n` items for each of `world` ranks, sizes differing by at most one."""
    base, rem = divmod(n, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def split_classes(costs, world):
    """Contiguous class ranges with balanced total cost (cost ~ seeds drawn per class = min(256, pool size))."""
    costs = np.asarray(costs, dtype=np.float64)
    total, out, lo, acc = costs.sum(), [], 0, 0.0
    for r in range(world):
        target = total * (r + 1) / world
        hi = lo
        while hi < len(costs) and (acc + costs[hi] <= target + 1e-9 or hi == lo) and len(costs) - hi > world - r - 1:
            acc += costs[hi]
            hi += 1
        if r == world - 1:
            hi = len(costs)
        out.append((lo, hi))
        lo = hi
    return out


class Collective:
    """all-reduce(sum) in place on engine buffers.  `dist` = torch.distributed (nccl on GPUs, gloo in the CPU tests)."""

    def __init__(self, dist=None, group=None):
        self.dist, self.group = dist, group
        self.timing = False          # bench.py --gpus N: event pairs around every exchange (per-buffer communication time)
        self._events = []

    def allreduce(self, t, name=None):
        if self.dist is not None and self.dist.get_world_size(self.group) > 1:
            if self.timing and t.is_cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
                e1.record()
                self._events.append((name or "?", t.numel() * t.element_size(), e0, e1))
            else:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def timing_report(self):
        """{buffer: {"calls", "bytes", "avg_us"}} of the exchanges recorded since timing was switched on (synchronises)."""
        out = {}
        for name, nbytes, e0, e1 in self._events:
            e1.synchronize()
            r = out.setdefault(name, {"calls": 0, "bytes": nbytes, "total_us": 0.0})
            r["calls"] += 1
            r["total_us"] += e0.elapsed_time(e1) * 1e3
        for r in out.values():
            r["avg_us"] = round(r.pop("total_us") / max(r["calls"], 1), 2)
        self._events = []
        return out


class Comm:
    """gs_comm handle (include/gslim.h): the all-reduce(sum) transport the library itself calls between the parts of a sharded
    iteration.  null() — no exchange (world 1, timing-only virtual ranks); rccl() — ncclAllReduce on a communicator the library
    creates (the unique id travels over torch.distributed); callback() — a Python function performs the exchange (gloo between
    processes sharing one GPU, in-process virtual ranks)."""

    TAGS = ("bn2_sums", "pair_scores", "dfeat_dadj_loss", "bn2_bwd_sums", "marginals_dw1", "tag5", "tag6", "other")

    def __init__(self, h, kind, keep=None):
        self.h, self.kind, self._keep = h, kind, keep

    @classmethod
    def null(cls, rank=0, world=1):
        h = C.c_void_p()
        check(lib().gs_comm_create_null(rank, world, C.byref(h)))
        return cls(h, "null")

    @classmethod
    def rccl(cls, dist, group=None):
        """One RCCL communicator over the ranks of `group`; the current CUDA device is this rank's GPU."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        buf = C.create_string_buffer(128)
        payload = None
        if rank == 0:
            try:      # a failure here (librccl not loadable) must still reach the broadcast: the other ranks are waiting in it
                check(lib().gs_comm_unique_id(buf))
                payload = buf.raw
            except Exception as e:      # noqa: BLE001
                payload = repr(e)
        box = [payload]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if not isinstance(box[0], bytes):
            raise RuntimeError(f"graphslim_amd: rank 0 could not create an RCCL unique id ({box[0]})")
        # ncclCommInitRank below is collective: a rank that cannot even load RCCL must not leave its peers waiting inside it.  Every
        # rank probes its own binding (gs_comm_probe: dlopen + symbols + ncclGetVersion — allocates nothing, starts no bootstrap
        # thread, not collective) and all agree before any enters.
        mine = 1.0
        try:
            check(lib().gs_comm_probe(None))
        except Exception:      # noqa: BLE001
            mine = 0.0
        flag = torch.tensor([mine], device=torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if float(flag) < 1.0:
            raise RuntimeError("graphslim_amd: the library's RCCL binding is not usable on every rank" + ("" if mine else " (not on this one)"))
        idb = C.create_string_buffer(box[0], 128)
        h = C.c_void_p()
        check(lib().gs_comm_create_rccl(idb, rank, world, C.byref(h)))
        return cls(h, "rccl")

    @classmethod
    def ipc(cls, dist, group=None, slot_bytes=2 << 20, fallback=None):
        """The one-shot peer-read transport (gs_comm_create_ipc): every rank allocates its arena on ITS current device, the 64-byte
        hipIpc handles travel over torch.distributed, every rank maps every other rank's arena.  `fallback`: a Comm for exchanges
        beyond slot_bytes (kept alive by the returned object).  Every rank reaches the same collectives whatever fails locally."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        arena, handle, err = C.c_void_p(), C.create_string_buffer(64), None
        try:
            check(lib().gs_comm_ipc_arena(slot_bytes, C.byref(arena), handle))
        except Exception as e:      # noqa: BLE001 — must still reach the gather below
            err = repr(e)
        box = [None] * world
        dist.all_gather_object(box, (handle.raw, err), group=group)
        bad = [(r, b[1]) for r, b in enumerate(box) if b[1]]
        if bad:
            raise RuntimeError(f"graphslim_amd: the ipc transport is not usable on every rank: {bad}")
        handles = C.create_string_buffer(b"".join(b[0] for b in box), 64 * world)
        h, err = C.c_void_p(), None
        try:
            check(lib().gs_comm_create_ipc(arena, handles, slot_bytes, rank, world, fallback.h if fallback is not None else None, C.byref(h)))
        except Exception as e:      # noqa: BLE001
            err = repr(e)
        box = [None] * world
        dist.all_gather_object(box, err, group=group)
        if any(box):
            raise RuntimeError(f"graphslim_amd: the ipc transport could not map every rank's arena: {[(r, b) for r, b in enumerate(box) if b]}")
        return cls(h, "ipc", keep=fallback)

    def ipc_error(self):
        """0, or r + 1 when an exchange of this ipc communicator gave up waiting for rank r (gs_comm_ipc_error)."""
        code = C.c_int()
        check(lib().gs_comm_ipc_error(self.h, C.byref(code)))
        return code.value

    @classmethod
    def rccl_world1(cls):
        """A one-rank RCCL communicator (the 1-GPU test leg of the RCCL binding)."""
        buf = C.create_string_buffer(128)
        check(lib().gs_comm_unique_id(buf))
        h = C.c_void_p()
        check(lib().gs_comm_create_rccl(buf, 0, 1, C.byref(h)))
        return cls(h, "rccl")

    @classmethod
    def callback(cls, fn, rank, world, host=False):
        """fn(tensor) all-reduces the given zero-copy view of the engine buffer in place (any exception fails the library call).
        host: the buffers are host memory (the CPU tests of the transport itself)."""

        def _cb(user, buf, count, dtype, stream):
            try:
                if host:
                    arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double if dtype else C.c_float)), shape=(count,))
                    fn(torch.from_numpy(arr))
                else:
                    from .engine import device_view
                    dev = torch.device("cuda", torch.cuda.current_device())
                    fn(device_view(buf, count, torch.float64 if dtype else torch.float32, dev))
                return 0
            except Exception:      # noqa: BLE001 — must not propagate through the C frames
                import traceback
                traceback.print_exc()
                return 1

        cb = ALLREDUCE_FN(_cb)
        h = C.c_void_p()
        check(lib().gs_comm_create_callback(cb, None, rank, world, C.byref(h)))
        return cls(h, "callback", keep=(cb, fn))

    @classmethod
    def from_dist(cls, dist=None, group=None, slot_bytes=2 << 20):
        """The transport that fits the process group: none -> null, nccl -> RCCL inside the library, anything else (gloo) ->
        callback onto dist.all_reduce.  GS_COMM_TRANSPORT=ipc puts the one-shot peer-read transport in front of that one (which
        stays as the fallback for exchanges beyond slot_bytes)."""
        if dist is None or dist.get_world_size(group) == 1:
            return cls.null()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        if dist.get_backend(group) == "nccl" and os.environ.get("GS_COMM", "rccl") == "rccl":
            base = cls.rccl(dist, group)
        else:
            base = cls.callback(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group), rank, world)
        if os.environ.get("GS_COMM_TRANSPORT") == "ipc":
            # Rank-uniform fallback (VERDICT r05 "next" 7c): a peer mapping that fails on ANY rank, or a first exchange whose flag wait
            # fails on ANY rank, sends EVERY rank to `base` — the failure is agreed on through torch.distributed (all_gather_object inside
            # Comm.ipc / ipc_self_test), so no rank is left alone inside a transport its peers have given up.
            try:
                c = cls.ipc(dist, group, slot_bytes, fallback=base)
            except RuntimeError as e:      # (raised on every rank together: the message comes from an all_gather)
                if rank == 0:
                    print(f"graphslim_amd: ipc transport unavailable ({e}); every rank falls back to {base.kind}", file=sys.stderr)
                return base
            if not c.ipc_self_test(dist, group):
                if rank == 0:
                    print(f"graphslim_amd: the ipc transport's first exchange failed on some rank; every rank falls back to {base.kind}", file=sys.stderr)
                return base
            return c
        return base

    def ipc_self_test(self, dist, group=None):
        """The first exchange of an ipc communicator, checked on every rank: all-reduce [rank + 1, 1] through the transport, wait, compare
        with the closed form and read the error word; the verdicts are gathered so that every rank returns the same bool."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        try:
            ok = self._first_exchange_ok(rank, world)
        except Exception:      # noqa: BLE001 — must still reach the gather below
            ok = False
        box = [None] * world
        dist.all_gather_object(box, bool(ok), group=group)
        return all(box)

    def _first_exchange_ok(self, rank, world):
        t = torch.tensor([rank + 1.0, 1.0], device=torch.device("cuda", torch.cuda.current_device()))
        self.allreduce(t)
        torch.cuda.synchronize()
        return self.ipc_error() == 0 and t.tolist() == [world * (world + 1) / 2.0, float(world)]

    def check(self):
        """GS_OK, or raises: a flag wait of the ipc transport timed out (sticky; gs_comm_check reads a host-visible word, no device
        synchronisation) — polled by ShardedGCond.read_loss once per epoch."""
        check(lib().gs_comm_check(self.h))

    def allreduce(self, t, tag=7):
        check(lib().gs_comm_allreduce(self.h, ptr(t), t.numel(), 1 if t.dtype == torch.float64 else 0, tag, stream()))

    @property
    def world(self):
        return lib().gs_comm_world(self.h)

    def set_timing(self, on):
        check(lib().gs_comm_set_timing(self.h, int(on)))

    def timing_report(self):
        """{exchange point: {"calls", "bytes" (per call), "avg_us"}} since set_timing(True) (synchronises the recorded events)."""
        calls, nbytes, us, timed = (C.c_long * 8)(), (C.c_long * 8)(), (C.c_double * 8)(), (C.c_long * 8)()
        check(lib().gs_comm_read_timing2(self.h, calls, nbytes, us, timed))
        # avg_us over the calls that WERE timed (the library's event pool is finite: "timed_calls" < "calls" says the window was longer)
        return {self.TAGS[t]: {"calls": calls[t], "bytes": nbytes[t] // max(calls[t], 1), "avg_us": round(us[t] / max(timed[t], 1), 2),
                               "timed_calls": timed[t]}
                for t in range(8) if calls[t]}

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            try:
                lib().gs_comm_destroy(h)
            except Exception:      # noqa: BLE001 — interpreter shutdown
                pass
            self.h = None


class ShardedGCond:
    """One rank's engine of a sharded run.  outer_step = ONE library call (gs_gcond_outer_step_sharded: the stage sequence of
    gs_gcond_outer_step with the all-reduces issued by the library through `comm`); `phases` / `outer_step_py` keep the same
    iteration as a host-side phase list over a torch.distributed Collective (the first form; tests compare the two)."""

    PARALLELISM = {"rows": "class-parallel + PGE pair-row sharding, 5-7 all-reduce(sum) per iteration issued inside the library (gs_comm)",
                   "replicated": "class-parallel, PGE replicated on every rank, ONE all-reduce(sum) per iteration issued inside the library (gs_comm)"}

    def __init__(self, engine, rank, world, pool_sizes, collective: Collective = None, comm: Comm = None, pge=None):
        """pge: "rows" — the PGE's n'^2 pair rows are split over the ranks (4 more exchange points per PGE pass pair: BatchNorm-2
        sums, pair scores, BatchNorm-2 backward sums, marginals); "replicated" — every rank runs the whole PGE on the all-reduced
        d_adj (bitwise the same on every rank: same code, same inputs), and the class-partial [d_feat | d_adj | loss_c] of the
        matching stage is the ONLY exchange of the iteration (VERDICT r03 "next" 6: <= 4 exchanges).  Default: GS_PGE_SHARDING or
        "rows"."""
        self.eng, self.rank, self.world = engine, rank, world
        self.pge = pge or os.environ.get("GS_PGE_SHARDING", "rows")
        if self.pge not in self.PARALLELISM:
            raise ValueError(f"ShardedGCond: pge must be one of {sorted(self.PARALLELISM)}, got {self.pge!r}")
        self.parallelism = self.PARALLELISM[self.pge]
        self.col = collective if collective is not None else Collective(None)
        self._comm = comm          # built on first use: the transport that fits the Collective's process group
        self.driver = os.environ.get("GS_SHARDED_DRIVER", "library")      # "python": the host-side phase list (first form)
        s = engine.shapes
        self.class_ranges = split_classes([min(s.bcap, int(p)) for p in pool_sizes], world)
        self.row_ranges = split_even(s.n_syn, world) if self.pge == "rows" else [(0, s.n_syn)] * world
        (c_lo, c_hi), (r_lo, r_hi) = self.class_ranges[rank], self.row_ranges[rank]
        engine.set_shard(c_lo, c_hi, r_lo, r_hi)
        f64 = torch.float64
        self.v = {k: engine.view(k, f64 if k in ("ar_bn2", "ar_l3") else torch.float32)
                  for k in ("ar_bn2", "ar_l3", "e3", "ar_syn", "ar_marg", "ar_w1")}
        # passthroughs so a ShardedGCond can stand in for a GCondEngine in bench.py / GCond.reduce
        for name in ("set_feat", "set_pge", "set_model", "set_forward_reuse", "get_model", "get_pge", "buffer", "shapes", "dev", "feat_syn",
                     "adj_syn", "pge_params", "model_params", "stage_sample", "get_blocks"):
            setattr(self, name, getattr(engine, name))

    def read_loss(self):
        """The engine's loss accumulator (one D2H per epoch) — and the transport's health: a timed-out exchange of the ipc transport left its
        buffer untouched and the replicas have diverged: fatal, raised here at the latest (gs_comm_allreduce refuses earlier
        when the host has already seen the word)."""
        out = self.eng.read_loss()
        if self._comm is not None:
            self._comm.check()
        return out

    def phases(self, epoch):
        """The sharded outer iteration as (stage call, name of the buffer to all-reduce afterwards or None)."""
        e = self.eng
        pge_epoch = epoch % 50 < 10
        np_, nf = pge_epoch, not pge_epoch

        rows = self.pge == "rows"      # (the PGE exchange points exist only when its pair rows are sharded)

        def fwd(inference):
            return [(lambda: e.pge_forward_part(inference, 1), "ar_bn2" if rows else None),
                    (lambda: e.pge_forward_part(inference, 2), "e3" if rows else None),
                    (lambda: e.pge_forward_part(inference, 3), None)]

        ph = [(e.stage_real_async, None)]      # second stream: overlaps the PGE forward and its two exchanges
        ph += [(e.adopt_cached_forward, None)] if e.forward_cached() else fwd(False)
        ph += [(e.stage_syn_match, "ar_syn"), (e.syn_finish, None),
               (lambda: e.pge_backward_part(np_, nf, 1), "ar_l3" if rows else None),
               (lambda: e.pge_backward_part(np_, nf, 2), "ar_marg" if rows else None)]
        if pge_epoch and rows:
            ph += [(lambda: None, "ar_w1")]
        ph += [(lambda: e.pge_backward_part(np_, nf, 3), None), (lambda: e.stage_update(epoch), None)]
        ph += fwd(True) + [(e.inner_steps, None)]
        if getattr(e.shapes, "dropout", 0.0):      # (the one-call iteration advances the masks' iteration counter itself)
            ph += [(lambda: e.set_dropout_iter(e.dropout_iter() + 1), None)]
        return ph

    def outer_step_py(self, epoch):
        for fn, buf in self.phases(epoch):
            fn()
            if buf is not None:
                self.col.allreduce(self.v[buf], buf)

    @property
    def comm(self):
        if self._comm is None:
            s = self.eng.shapes      # the largest join of an iteration: [dX~ | d adj | loss_c], the marginals, dW1 | db1, the pair scores
            n, d, ph, nc = (int(getattr(s, k, 0) or 0) for k in ("n_syn", "d", "pge_hidden", "nclass"))
            largest = 4 * max(n * d + n * n + nc, 2 * n * ph, ph * ph + ph, n * n)
            self._comm = Comm.from_dist(self.col.dist, self.col.group, slot_bytes=max(largest, 1 << 16))
        return self._comm

    def outer_step(self, epoch):
        if self.driver == "python":
            return self.outer_step_py(epoch)
        self.eng.outer_step_sharded(epoch, self.comm)

    def verify_transport(self):
        """Before the first iteration of a multi-rank job: all-reduce a known vector through the library's transport and through
        torch.distributed and compare.  A transport that cannot be created or disagrees (an RCCL build this library's binding does
        not fit) switches EVERY rank to the host-side phase list over torch.distributed, loudly.

        Every rank issues the same sequence of torch.distributed collectives whatever happens locally (a rank whose
        library call raised used to skip the reference all-reduces while its peers were inside them):
          1. [unique-id broadcast inside Comm.rccl — reached by every rank, rank 0 broadcasts its error text if it has one]
          2. flag MIN all-reduce: "my communicator exists" — nobody touches the library transport unless all ranks have one
          3. the library all-reduces (only if 2 agreed), each in its own try
          4. the two reference all-reduces over torch.distributed — ALWAYS
          5. flag MIN all-reduce: "my results agree" — ALWAYS
        GS_SHARDED_DRIVER=python skips the library transport altogether."""
        import sys
        dist, group = self.col.dist, self.col.group
        if dist is None or dist.get_world_size(group) == 1:
            return True
        if self.driver == "python":
            return False
        dev = self.eng.dev
        on_gpu = dist.get_backend(group) == "nccl"

        def agree(ok):
            flag = torch.tensor([1.0 if ok else 0.0], device=dev if on_gpu else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return float(flag) >= 1.0

        def complain(what, e):
            print(f"graphslim_amd.parallel: rank {self.rank}: {what} ({e!r})", file=sys.stderr, flush=True)

        have = True
        try:
            _ = self.comm                      # (collective for RCCL: every rank gets here)
        except Exception as e:                 # noqa: BLE001
            complain("library transport could not be created", e)
            have = False
        all_have = agree(have)                 # 2.
        t = torch.arange(4096, device=dev, dtype=torch.float32) * float(self.rank + 1)
        d = torch.arange(64, device=dev, dtype=torch.float64) * float(self.rank + 1)
        rt, rd = t.clone(), d.clone()
        ok = all_have
        if all_have:                           # 3.
            try:
                self.comm.allreduce(t, 7)
                self.comm.allreduce(d, 7)
            except Exception as e:             # noqa: BLE001
                complain("library all-reduce failed", e)
                ok = False
        dist.all_reduce(rt, op=dist.ReduceOp.SUM, group=group)      # 4.
        dist.all_reduce(rd, op=dist.ReduceOp.SUM, group=group)
        if on_gpu:
            torch.cuda.synchronize(dev)
        if ok and not (torch.equal(t, rt) and torch.equal(d, rd)):
            complain("library all-reduce disagrees with torch.distributed", None)
            ok = False
        if not agree(ok):                      # 5.
            print(f"graphslim_amd.parallel: rank {self.rank}: falling back to the host-side phase list over torch.distributed",
                  file=sys.stderr, flush=True)
            self.driver = "python"
            return False
        return True


class RowShard:
    """Row-block sharding of a full-graph pass (SURVEY.md §8e "mode F": the evaluation GCN's per-epoch full-graph forward,
    models/base.py:168-173).  Rank r owns the contiguous row block [lo, hi) of the normalised adjacency, balanced by
    non-zeros.  Per forward there is ONE exchange: an all-gather of the [N, C] second-layer operand (the first layer is
    local because S = A^ X is precomputed per row block); accuracies are an all-reduce(sum) of correct counts.  Rows are
    computed by the same kernels in the same order as on one GPU, so a sharded forward is bit-identical to the unsharded
    one.  `dist` = torch.distributed (nccl = RCCL on GPUs, gloo in the CPU tests); None = single process."""

    def __init__(self, dist=None, group=None):
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if dist is not None else 1
        self.rank = dist.get_rank(group) if dist is not None else 0
        self._blocks = {}

    @staticmethod
    def split_rows(rowptr, world):
        """Contiguous row blocks with balanced non-zero counts: [(lo, hi)] * world, covering every row exactly once."""
        rowptr = np.asarray(rowptr, dtype=np.int64)
        n, nnz = len(rowptr) - 1, int(rowptr[-1])
        cuts = [0]
        for r in range(1, world):
            t = nnz * r / world
            c = int(np.searchsorted(rowptr, t, side="left"))
            if c > 0 and t - rowptr[c - 1] < rowptr[min(c, n)] - t:
                c -= 1
            cuts.append(min(max(c, cuts[-1]), n))
        cuts.append(n)
        return [(cuts[r], cuts[r + 1]) for r in range(world)]

    def blocks(self, adj):
        key = id(adj)
        if key not in self._blocks:
            a = adj.tocsr()
            self._blocks[key] = (self.split_rows(a.indptr + np.arange(a.shape[0] + 1), self.world), adj)   # + self loops
        return self._blocks[key][0]

    def row_block(self, adj):
        if self.world == 1:
            return 0, adj.shape[0]
        return self.blocks(adj)[self.rank]

    def all_gather_rows(self, local, n, blocks):
        """[hi - lo, ld] row blocks of every rank -> [n, ld] on every rank (blocks padded to the largest, one collective)."""
        if self.world == 1:
            return local
        cap = max(hi - lo for lo, hi in blocks)
        send = local.new_zeros(cap, local.shape[1])
        send[: local.shape[0]] = local
        recv = local.new_empty(self.world * cap, local.shape[1])
        self.dist.all_gather_into_tensor(recv, send, group=self.group)
        out = local.new_empty(n, local.shape[1])
        for r, (lo, hi) in enumerate(blocks):
            out[lo:hi] = recv[r * cap: r * cap + (hi - lo)]
        return out

    def all_reduce_sum(self, t):
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t


class RunShard:
    """The `run_eval` seeded evaluation runs of Evaluator.evaluate (eval_agent.py:266-283) are independent: run i goes
    to rank i % world; the per-run accuracies are combined with one all-reduce(sum) of a zero-filled vector, so every
    rank returns the same mean / std as a single-GPU evaluation (same seeds, same per-run arithmetic)."""

    def __init__(self, dist=None, group=None):
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if dist is not None else 1
        self.rank = dist.get_rank(group) if dist is not None else 0

    def mine(self, n_runs):
        return [i for i in range(n_runs) if i % self.world == self.rank]

    def combine(self, n_runs, results, device="cpu"):
        """results: {run index: value} of this rank's runs -> float64 vector of all runs on every rank."""
        v = torch.zeros(n_runs, dtype=torch.float64, device=device)
        for i, r in results.items():
            v[i] = r
        if self.world > 1:
            self.dist.all_reduce(v, op=self.dist.ReduceOp.SUM, group=self.group)
        return v.cpu().numpy()


def active_dist():
    """torch.distributed if a multi-rank process group is initialised, else None."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


def make_engine(p, dev, rank, world, pge=None):
    """bench.py entry: this rank's sharded engine over torch.distributed (process group already initialised)."""
    import torch.distributed as dist
    from .workloads import make_engine as _make
    eng = _make(p, dev)
    sh = ShardedGCond(eng, rank, world, [len(q) for q in p["pools"]], Collective(dist), pge=pge)
    sh.verify_transport()
    return sh
