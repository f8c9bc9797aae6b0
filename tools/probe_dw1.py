#!/usr/bin/env python3
"""PGE-epoch backward at the headline shape, repeated: run under `rocprofv3 --kernel-trace --stats` to read the duration of
gs_dw1_gen_kernel (dW1 | db1 on generated operands) and of the kernels around it.  usage: probe_dw1.py [operand mode] [repeats]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else None
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
p = bench.build_problem("ogbn-arxiv", seed=1)
eng = bench.make_engine(p, "cuda:0")
if mode:
    eng.set_operand_mode(mode)
bench.init_state(p, eng)
bench.new_model(p, eng)
eng.stage_sample()
eng.stage_real_grads()
eng.stage_pge_forward(False)
eng.stage_syn_match()
eng.syn_finish()
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    eng.stage_pge_backward(True, False)
t0.record()
for _ in range(reps):
    eng.stage_pge_backward(True, False)
t1.record()
torch.cuda.synchronize()
print(f"PGE-epoch backward, operand mode {eng.operand_mode}: {1e3 * t0.elapsed_time(t1) / reps:.1f} us per call")
