#!/usr/bin/env python3
"""Host time to ISSUE outer iterations (no back-pressure: few steps into an idle queue) vs their device time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
p = bench.build_problem("ogbn-arxiv", seed=1)
eng = bench.make_engine(p, "cuda:0")
bench.init_state(p, eng)
bench.new_model(p, eng)
from graphslim_amd.parallel import Comm, ShardedGCond
for mode in ("bf16x6", "f32", "sharded-8 (rank 3 of 8, one library call per iteration)"):
    if mode.startswith("sharded"):
        eng.set_operand_mode("bf16x6")
        eng = ShardedGCond(eng, 3, 8, [len(q) for q in p["pools"]], comm=Comm.null(3, 8))
    else:
        eng.set_operand_mode(mode)
    for _ in range(10):
        eng.outer_step(10)
    torch.cuda.synchronize()
    for nstep in (1, 2, 4, 8, 16, 64):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nstep):
            eng.outer_step(10)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{mode}: {nstep:3d} steps: host issue {1e3 * (t1 - t0) / nstep:.3f} ms/step, total {1e3 * (t2 - t0) / nstep:.3f} ms/step")
