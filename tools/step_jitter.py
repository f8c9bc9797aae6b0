#!/usr/bin/env python3
"""Per-group times of the fused outer step (groups of 10 steps, HIP events): is the iteration time stable from group to group and
from process to process?  usage: step_jitter.py [groups] ; GS_HOIST_FWD=0|1 selects the schedule."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

groups = int(sys.argv[1]) if len(sys.argv) > 1 else 30
if os.environ.get("GS_HOIST_FWD") is not None:
    from graphslim_amd import _lib
    _lib.lib().gs_gcond_set_hoist_forward_all(int(os.environ["GS_HOIST_FWD"]))
if os.environ.get("GS_HOIST_SHARE") is not None:
    from graphslim_amd import _lib
    _lib.lib().gs_panel_set_hoist_share(int(os.environ["GS_HOIST_SHARE"]))
p = bench.build_problem("ogbn-arxiv", seed=1)
eng = bench.make_engine(p, "cuda:0")
bench.init_state(p, eng)
bench.new_model(p, eng)
for _ in range(10):
    eng.outer_step(10)
torch.cuda.synchronize()
out = []
for g in range(groups):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(10):
        eng.outer_step(0 if (g * 10 + i) % 5 == 0 else 10)
    b.record()
    torch.cuda.synchronize()      # (all streams: the hoisted pass of the last step included)
    out.append(a.elapsed_time(b) * 100.0)      # us per step
out = sorted(out)
print(f"hoist {os.environ.get('GS_HOIST_FWD', 'default')}: us/step over {groups} groups of 10: min {out[0]:.0f}  median {out[len(out) // 2]:.0f}  "
      f"p90 {out[int(0.9 * len(out))]:.0f}  max {out[-1]:.0f}")
