#!/usr/bin/env python3
"""Launches the 128-row kernel and the persistent 256-row kernel (plain product, PGE layer-2 shape and H1 shape, bf16x6) a few
times each — the subject of rocprofv3 --pmc passes (tools/gpu/pmc_tall.sh)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphslim_amd import _lib  # noqa: E402
from graphslim_amd.engine import gemm_desc  # noqa: E402

dev = "cuda:0"
L = _lib.lib()
for (M, K) in [(206116, 256), (169343, 128)]:
    torch.manual_seed(0)
    a = torch.randn(M, K, device=dev)
    w = torch.randn(256, K, device=dev) / K ** 0.5
    b = torch.randn(256, device=dev)
    out = torch.empty(M, 256, device=dev)
    scratch = torch.empty(256 * K * 3 // 2 + 16, device=dev)
    d, keep = gemm_desc(a, w, out, M, 256, K, (K, 1), (1, K), 256, prec=3, scratch=scratch, bias=b, relu=True)
    for tall in (0, 1):
        L.gs_gemm_set_panel_tall(tall)
        for _ in range(6):
            _lib.check(L.gs_gemm_f32(C.byref(d), _lib.stream()))
        torch.cuda.synchronize()
L.gs_gemm_set_panel_tall(1)
print("probe done")
