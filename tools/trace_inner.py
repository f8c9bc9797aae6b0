#!/usr/bin/env python3
"""PGE.inference + the inner SGC steps, repeated, for a rocprofv3 kernel trace (tools/timeline_csv.py splits the trace at the
gs_loss_accum_kernel marker launched between repetitions).  usage: trace_inner.py [workload] [fused 0|1] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from graphslim_amd import _lib  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "ogbn-arxiv"
fused = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
_lib.lib().gs_gcond_set_fused_chains(fused)
p = bench.build_problem(wl, seed=1)
eng = bench.make_engine(p, "cuda:0")
bench.init_state(p, eng)
bench.new_model(p, eng)
eng.outer_step(10)
torch.cuda.synchronize()
for _ in range(reps):
    eng.syn_finish()
    eng.inner_steps()
    torch.cuda.synchronize()
eng.syn_finish()
torch.cuda.synchronize()
