#!/usr/bin/env python3
"""Per-launch time distribution of the PGE backward's second part (fused marginals) and the forward's first part at a workload's shape,
128-row kernels vs the persistent 256-row kernels: looks for rare long launches (tools/ab_tall.py once averaged 4.3 ms over 20 reps
at the reddit shape).  usage: repro_tall_stall.py [workload] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from graphslim_amd import _lib  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
L = _lib.lib()
p = bench.build_problem(wl, seed=1)
eng = bench.make_engine(p, "cuda:0")
bench.init_state(p, eng)
bench.new_model(p, eng)
eng.outer_step(10)
eng.set_operand_mode("bf16x6")


def dist(fn):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    fn()
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return f"median {t[len(t) // 2]:.1f} us, p99 {t[int(len(t) * 0.99)]:.1f}, max {t[-1]:.1f}, mean {sum(t) / len(t):.1f}"


for rnd in range(3):
    for tall in (0, 1):
        L.gs_gemm_set_panel_tall(tall)
        f = dist(lambda: eng.pge_forward_part(False, 1))
        eng.pge_forward_part(False, 1)
        eng.pge_forward_part(False, 2)
        eng.pge_forward_part(False, 3)
        eng.stage_syn_match()
        eng.pge_backward_part(False, True, 1)
        b = dist(lambda: eng.pge_backward_part(False, True, 2))
        print(f"{wl} round {rnd} tall {tall}: forward part 1 {f} | backward part 2 {b}", flush=True)
L.gs_gemm_set_panel_tall(1)
