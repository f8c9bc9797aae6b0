#!/usr/bin/env python3
"""Where the persistent 256-row kernel (csrc/gs_panel2.hip) spends its time: the plain product at the PGE layer-2 shape and at the
H1 shape with the K loop repeated in-kernel (krep = 1, 2, 3 -> time per pass of the loop and the rest) and phases switched off
(gs_panel_tall_debug triage bits; results are garbage then).  usage: triage_tall.py"""
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
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for (M, K) in [(206116, 256), (169343, 128)]:
    torch.manual_seed(0)
    a = torch.randn(M, K, device=dev)
    w = torch.randn(256, K, device=dev) / K ** 0.5
    b = torch.randn(256, device=dev)
    out = torch.empty(M, 256, device=dev)
    scratch = torch.empty(256 * K * 3 // 2 + 16, device=dev)
    d, keep = gemm_desc(a, w, out, M, 256, K, (K, 1), (1, K), 256, prec=3, scratch=scratch, bias=b, relu=True)
    run = lambda: _lib.check(L.gs_gemm_f32(C.byref(d), _lib.stream()))
    L.gs_gemm_set_panel_tall(0)
    print(f"M={M} K={K}: 128-row kernel {timed(run):.1f} us")
    L.gs_gemm_set_panel_tall(1)
    for grid in (0, 224, 192):
        L.gs_panel_tall_debug(grid, 1, 0)
        print(f"  tall grid={grid or 'all'}: {timed(run):.1f} us")
    for tri, what in [(0, "product"), (1, "no epilogue"), (2, "no A staging (vmcnt(0) waits)"), (4, "no B staging (vmcnt(0) waits)"),
                      (6, "no staging"), (14, "no staging, no waits / barriers"), (8, "no waits / barriers"), (16, "no MFMAs"),
                      (32, "no operand split"), (15, "loop = MFMA + LDS reads only, no epilogue"), (31, "nothing but LDS reads + VALU")]:
        ts = []
        for R in (1, 2, 3):
            L.gs_panel_tall_debug(0, R, tri)
            ts.append(timed(run))
        per = (ts[2] - ts[0]) / 2
        print(f"  triage={tri:2d} ({what}): R=1,2,3 -> {ts[0]:.1f} {ts[1]:.1f} {ts[2]:.1f} us; per pass {per:.1f} us, rest {ts[0] - per:.1f} us", flush=True)
    L.gs_panel_tall_debug(0, 1, 0)
