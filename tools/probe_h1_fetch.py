#!/usr/bin/env python3
"""HBM bytes of the H1-shaped row-panel product (M = 169 343, K = 128, N = 256, exact split) with phases of the kernel switched
off (gs_panel_debug triage bits: 8 = no A-row loads in the loop, 32 = no B staging, 1 = no stores): run under
`rocprofv3 --kernel-trace --pmc FETCH_SIZE` (or WRITE_SIZE); the dispatches appear in the order printed here, 3 per setting.
usage: probe_h1_fetch.py [lda]   (row stride of the A operand in floats, default 128)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphslim_amd import _lib  # noqa: E402
from graphslim_amd.engine import gemm_desc  # noqa: E402

dev = "cuda:0"
L = _lib.lib()
lda = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M, K = 169343, 128
a = torch.randn(M, lda, device=dev)
w = torch.randn(256, K, device=dev) / 16
b = torch.randn(256, device=dev)
out = torch.empty(M, 256, device=dev)
scratch = torch.empty(256 * K * 3 // 2 + 16, device=dev)
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for tri in (0, 8, 32, 1, 41):
    L.gs_panel_debug(1, tri)
    d, keep = gemm_desc(a, w, out, M, 256, K, (lda, 1), (1, K), 256, prec=3, scratch=scratch)
    for _ in range(3):
        big.zero_()          # flush L2 / Infinity Cache between launches
        _lib.check(L.gs_gemm_f32(C.byref(d), _lib.stream()))
    torch.cuda.synchronize()
    print("triage", tri)
L.gs_panel_debug(1, 0)
