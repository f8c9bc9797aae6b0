#!/usr/bin/env python3
"""One virtual rank of a `world`-way sharded run on one GPU (no exchange; timing only) — meant to be run under
`rocprofv3 --kernel-trace --stats` to see kernel time vs wall time per sharded outer iteration.
usage: virtual_rank.py WORLD RANK [STEPS]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from graphslim_amd.parallel import Comm, ShardedGCond  # noqa: E402

world, rank = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
p = bench.build_problem("ogbn-arxiv", seed=1)
eng = bench.make_engine(p, "cuda:0")
bench.init_state(p, eng)
sh = ShardedGCond(eng, rank, world, [len(q) for q in p["pools"]], comm=Comm.null(rank, world)) if world > 0 else eng
bench.run_steps(p, sh, 5, 5)
torch.cuda.synchronize()
t0 = time.perf_counter()
bench.run_steps(p, sh, steps, steps)
torch.cuda.synchronize()
print(f"VRANK world {world} rank {rank}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step over {steps} steps (+5 warm-up)")
