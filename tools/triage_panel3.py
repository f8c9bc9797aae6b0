#!/usr/bin/env python3
"""Where the exact-split row-panel loop spends its time: the K loop repeated in-kernel (slope = one pass) with phases compiled in
but switched off at run time (gs_panel_debug triage bits; results are garbage in those runs): 4 = no waits / barrier, 8 = no A-row loads inside
the loop, 32 = no B staging inside the loop, 16 = operand split only in the first step, 1 = no stores."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphslim_amd import _lib  # noqa: E402
from graphslim_amd.engine import gemm_desc  # noqa: E402

dev = "cuda:0"
L = _lib.lib()


def timed(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


M, K = 454 * 454, 256
a = torch.randn(M, K, device=dev)
w = torch.randn(256, K, device=dev) / 16
out = torch.empty(M, 256, device=dev)
scratch = torch.empty(256 * K * 3 // 2 + 16, device=dev)
for tri in (0, 1, 4, 8, 32, 40, 16, 60, 61):
    ts = []
    for R in (1, 2, 3):
        L.gs_panel_debug(R, tri)
        d, keep = gemm_desc(a, w, out, M, 256, K, (K, 1), (1, K), 256, prec=3, scratch=scratch)
        ts.append(min(timed(lambda: _lib.check(L.gs_gemm_f32(C.byref(d), _lib.stream()))) for _ in range(3)))
    L.gs_panel_debug(1, 0)
    print(f"prec=3 triage={tri:2d}: R=1,2,3 -> {ts[0]:.1f} {ts[1]:.1f} {ts[2]:.1f} us; per pass {ts[2] - ts[1]:.1f} us, overhead {ts[0] - (ts[2] - ts[1]):.1f} us")

# the same on the product the bench's roofline names: the PGE forward with the generated A1 operand (A streams from the L2-resident
# U, V instead of HBM), through an ogbn-arxiv-shaped engine
del a, out
import bench  # noqa: E402
p = bench.build_problem("ogbn-arxiv", seed=1)
eng = bench.make_engine(p, dev)
bench.init_state(p, eng)
bench.new_model(p, eng)
eng.outer_step(10)
torch.cuda.synchronize()
for tri in (0, 1, 8, 32, 40, 16, 60):
    ts = []
    for R in (1, 2, 3):
        L.gs_panel_debug(R, tri)
        ts.append(min(timed(lambda: eng.pge_forward_part(False, 1)) for _ in range(3)))
    L.gs_panel_debug(1, 0)
    print(f"fwdgen prec=3 triage={tri:2d}: R=1,2,3 -> {ts[0]:.1f} {ts[1]:.1f} {ts[2]:.1f} us; per pass {(ts[2] - ts[0]) / 2:.1f} us, rest of part 1 {ts[0] - (ts[2] - ts[0]) / 2:.1f} us")
