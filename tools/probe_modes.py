#!/usr/bin/env python3
"""Error of the row-panel product (gs_gemm_f32 route 5) against fp64 in every operand mode at N = 128 and N = 256 — mode 3 (the exact
three-way split) must sit at fp32 accumulation noise whatever the shape.  usage: probe_modes.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphslim_amd import _lib  # noqa: E402
from graphslim_amd.engine import gemm_desc  # noqa: E402

dev = "cuda:0"
L = _lib.lib()
for (M, N, K) in [(4900, 128, 128), (4900, 256, 256), (19600, 128, 128), (4900, 128, 64), (300, 128, 128), (4900, 128, 256)]:
    torch.manual_seed(0)
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    ref = a.double() @ w.double().T + b.double()
    scratch = torch.empty(N * K * 3 // 2 + 64, device=dev)
    line = []
    for prec in (0, 1, 2, 3):
        out = torch.full((M, N), float("nan"), device=dev)
        d, keep = gemm_desc(a, w, out, M, N, K, (K, 1), (1, K), N, prec=prec, scratch=scratch, bias=b)
        route = L.gs_gemm_route(C.byref(d))
        _lib.check(L.gs_gemm_f32(C.byref(d), _lib.stream()))
        torch.cuda.synchronize()
        line.append(f"prec {prec}: {float((out.double() - ref).abs().max() / ref.abs().max()):.2e}")
    print(f"M={M} N={N} K={K} route {route}: " + "  ".join(line), flush=True)
