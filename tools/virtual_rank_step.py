#!/usr/bin/env python3
"""The fused sharded step of ONE virtual rank (rank r of w, null transport: the exchanges are no-ops) alone on the GPU — what a rank of
a w-GPU run computes per iteration, without its communication.  usage: virtual_rank_step.py [rank] [world] [layout]
(Sharded ranks keep the training forward inside its iteration: profiles/r05/virtual_rank_step_sharded_hoist_not_kept.txt.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from graphslim_amd import _lib  # noqa: E402
from graphslim_amd.parallel import Comm, ShardedGCond  # noqa: E402

rank, world = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3, 8)
layout = sys.argv[3] if len(sys.argv) > 3 else "rows"
if os.environ.get("GS_HOIST_FWD") is not None:
    _lib.lib().gs_gcond_set_hoist_forward_all(int(os.environ["GS_HOIST_FWD"]))
p = bench.build_problem("ogbn-arxiv", seed=1)
eng = bench.make_engine(p, "cuda:0")
bench.init_state(p, eng)
bench.new_model(p, eng)
sh = ShardedGCond(eng, rank, world, [len(q) for q in p["pools"]], comm=Comm.null(rank, world), pge=layout)
for _ in range(20):
    sh.outer_step(10)
torch.cuda.synchronize()
out = []
for g in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(10):
        sh.outer_step(0 if i % 5 == 0 else 10)
    b.record()
    torch.cuda.synchronize()
    out.append(a.elapsed_time(b) * 100.0)
out = sorted(out)
print(f"rank {rank} of {world} ({layout}), GS_HOIST_FWD={os.environ.get('GS_HOIST_FWD', 'default')}: us/step min {out[0]:.0f} median {out[len(out) // 2]:.0f} max {out[-1]:.0f}")
