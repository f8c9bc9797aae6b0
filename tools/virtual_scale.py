#!/usr/bin/env python3
"""Compute-only estimate of multi-GPU scaling on ONE GPU: for world in (1, 2, 4, 8) every virtual rank's sharded outer
iteration (gs_gcond_outer_step_sharded: ONE library call per iteration, null transport = exchanges are no-ops) is timed in turn; max over ranks is the
per-step time the slowest rank would need, without RCCL latency.  Partial sums are not exchanged, so the values are
meaningless — only the timing is used (no data-dependent control flow in the engine)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from graphslim_amd.parallel import Comm, ShardedGCond  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "ogbn-arxiv"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
worlds = [int(w) for w in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
PGE = sys.argv[4] if len(sys.argv) > 4 else None      # "rows" (default) | "replicated": how the PGE is laid over the ranks
if os.environ.get("GS_H1_ROWS") is not None:      # A/B of the row-list H1 product of a class shard
    from graphslim_amd import _lib
    _lib.lib().gs_gcond_set_h1_rows(int(os.environ["GS_H1_ROWS"]))
p = bench.build_problem(wl, seed=1)
dev = "cuda:0"
eng = bench.make_engine(p, dev)
bench.init_state(p, eng)
pools = [len(q) for q in p["pools"]]


def timed(e):
    bench.run_steps(p, e, 5, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_steps(p, e, steps, steps)
    global host_ms
    host_ms = (time.perf_counter() - t0) / steps * 1e3      # host time to ISSUE the steps (before the sync)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def timed3(e):
    """median of three measurements (one in three 40-step windows on this pool's boxes runs at about half speed, whichever rank it
    hits — host issue rises with it: a box-level disturbance, not a property of a rank)"""
    global host_ms
    r = sorted((timed(e), host_ms) for _ in range(3))
    host_ms = r[1][1]
    return r[1][0]


base = timed3(eng)
print(f"PGE sharding: {PGE or os.environ.get('GS_PGE_SHARDING', 'rows')}")
print(f"fused single-GPU outer step: {base:.3f} ms (host issue {host_ms:.3f} ms)")
for world in worlds:
    ts, hs = [], []
    for r in range(world):
        sh = ShardedGCond(eng, r, world, pools, comm=Comm.null(r, world), pge=PGE)      # the library's one-call sharded iteration, exchanges = no-ops
        bench.init_state(p, eng)
        ts.append(timed3(sh))
        hs.append(round(host_ms, 3))
    print(f"world {world}: per-rank ms {[round(t, 3) for t in ts]}  max {max(ts):.3f}  compute-only speed-up {base / max(ts):.2f}x  host issue ms {hs}", flush=True)
eng.set_shard(0, p["data"].nclass, 0, len(p["labels_syn"]))
