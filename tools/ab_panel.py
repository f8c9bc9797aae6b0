#!/usr/bin/env python3
"""Interleaved A/B timing (one process, HIP events) of the PGE layer-2 / dY1 / H1full products: 128x128 tile vs the row-panel
kernel in its three operand precisions."""
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


def case(name, M, K, **kw):
    a = torch.randn(M, K, device=dev)
    w = torch.randn(256, K, device=dev) / K ** 0.5
    out = torch.empty(M, 256, device=dev)
    scratch = torch.empty(256 * K * 3 // 2 + 16, device=dev)
    extra = {k: (v() if callable(v) else v) for k, v in kw.items()}
    res = {}
    variants = [("tile128", 0, 0, 0), ("panel f32 wide-tail", 1, 0, 0), ("panel f32", 1, 0, 1),
                ("panel bf16x6 (exact split)", 1, 3, 1), ("panel bf16x3 wide-tail", 1, 1, 0), ("panel bf16x3", 1, 1, 1),
                ("panel bf16", 1, 2, 1)]
    for rnd in range(3):
        for label, panel, prec, deph in variants:
            L.gs_gemm_set_panel(panel)
            L.gs_gemm_set_panel_narrow_tail(deph)
            d, keep = gemm_desc(a, w, out, M, 256, K, (K, 1), (1, K), 256, prec=prec, scratch=scratch, **extra)
            t = timed(lambda: _lib.check(L.gs_gemm_f32(C.byref(d), _lib.stream())))
            res.setdefault(label, []).append(t)
    L.gs_gemm_set_panel(1)
    L.gs_gemm_set_panel_narrow_tail(1)
    flops = 2.0 * M * 256 * K
    for label, ts in res.items():
        t = min(ts)
        print(f"{name:28s} {label:26s} min {t:7.1f} us  median {sorted(ts)[1]:7.1f} us   {flops / t / 1e6:7.1f} TFLOP/s   "
              f"{(M * K + M * 256) * 4 / t / 1e6:5.2f} TB/s (A read + C write)")




def looprate():
    """K loop repeated R times inside the kernel: slope = time of one pass of the main loop, intercept = per-tile overhead."""
    M, K = 454 * 454, 256
    a = torch.randn(M, K, device=dev)
    w = torch.randn(256, K, device=dev) / 16
    out = torch.empty(M, 256, device=dev)
    scratch = torch.empty(256 * K * 3 // 2 + 16, device=dev)
    for prec in (0, 3, 1):
        for tri in (0, 1):
            ts = []
            for R in (1, 2, 3):
                L.gs_panel_debug(R, tri)
                d, keep = gemm_desc(a, w, out, M, 256, K, (K, 1), (1, K), 256, prec=prec, scratch=scratch)
                ts.append(min(timed(lambda: _lib.check(L.gs_gemm_f32(C.byref(d), _lib.stream()))) for _ in range(3)))
            L.gs_panel_debug(1, 0)
            print(f"loop-rate prec={prec} nostore={tri}: R=1,2,3 -> {ts[0]:.1f} {ts[1]:.1f} {ts[2]:.1f} us; per pass {ts[2] - ts[1]:.1f} us, "
                  f"overhead {ts[0] - (ts[2] - ts[1]):.1f} us")


looprate()
rows = 454 * 454
case("E2 = A1 W1^T + b1 (+stats)", rows, 256, bias=lambda: torch.randn(256, device=dev),
     colstat=lambda: torch.empty(-(-rows // 128), 2, 256, device=dev))
case("dY1 = (dE2 W1)*[A1>0]", rows, 256, mask=lambda: torch.randn(rows, 256, device=dev), ldmask=256)
case("H1full = relu(X W1 + b1)", 169343, 128, bias=lambda: torch.randn(256, device=dev), relu=1)
