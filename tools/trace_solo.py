#!/usr/bin/env python3
"""Feature-epoch outer iterations with the two-stream overlap OFF (every kernel alone on the chip): run under
`rocprofv3 --kernel-trace` and feed the trace to tools/timeline_csv.py for the solo duration of every launch of one iteration.
usage: trace_solo.py [workload] [iterations] [epoch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ogbn-arxiv"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
epoch = int(sys.argv[3]) if len(sys.argv) > 3 else 10
p = bench.build_problem(name, seed=1)
eng = bench.make_engine(p, "cuda:0")
eng.set_overlap(False)
bench.init_state(p, eng)
bench.new_model(p, eng)
for _ in range(iters):
    eng.outer_step(epoch)
torch.cuda.synchronize()
