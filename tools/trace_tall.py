#!/usr/bin/env python3
"""Per-step timeline of the persistent 256-row kernel (csrc/gs_panel2.hip, plain product): three s_memtime stamps per 16-k step of
waves 0 (runs [loads, prepare, MFMA]) and 4 (runs [MFMA, loads, prepare]) of the first workgroups -> cycles spent waiting for loads,
at the barrier, in the first and in the second part of a step.  usage: trace_tall.py [M K]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphslim_amd import _lib  # noqa: E402
from graphslim_amd.engine import gemm_desc  # noqa: E402

dev = "cuda:0"
L = _lib.lib()
M, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (206116, 256)
torch.manual_seed(0)
a = torch.randn(M, K, device=dev)
w = torch.randn(256, K, device=dev) / K ** 0.5
b = torch.randn(256, device=dev)
out = torch.empty(M, 256, device=dev)
scratch = torch.empty(256 * K * 3 // 2 + 16, device=dev)
d, keep = gemm_desc(a, w, out, M, 256, K, (K, 1), (1, K), 256, prec=3, scratch=scratch, bias=b, relu=True)
L.gs_gemm_set_panel_tall(1)
for _ in range(5):
    _lib.check(L.gs_gemm_f32(C.byref(d), _lib.stream()))
buf = torch.zeros(16 * 2816, dtype=torch.int32, device=dev)
L.gs_panel_tall_trace(C.c_void_p(buf.data_ptr()))
_lib.check(L.gs_gemm_f32(C.byref(d), _lib.stream()))
torch.cuda.synchronize()
L.gs_panel_tall_trace(None)
t = buf.cpu().numpy().astype(np.int64).reshape(16, 2, 1408) & 0xFFFFFFFF
nsb = K // 16
for wg in (0, 1, 9, 15):
    for half, name in ((0, "wave 0 [loads, prepare | MFMA]"), (1, "wave 4 [MFMA | loads, prepare]")):
        s = t[wg, half]
        n = int((s != 0).sum()) // 3
        if n < 2:
            continue
        st = s[:3 * n].reshape(n, 3)          # arrive, release, mid
        arrive, release, mid = st[:, 0], st[:, 1], st[:, 2]
        nxt = np.append(arrive[1:], arrive[-1])
        bar = (release - arrive) % (1 << 32)
        p1 = (mid - release) % (1 << 32)
        p2 = (nxt - mid) % (1 << 32)
        tot = (nxt - arrive) % (1 << 32)
        print(f"wg {wg} {name}: {n} steps, {nsb} per tile; cycles per step (median): total {np.median(tot[:-1]):.0f}, barrier {np.median(bar):.0f}, "
              f"first part {np.median(p1):.0f}, second part + waits {np.median(p2[:-1]):.0f}")
        for i0 in range(0, min(n - 1, 2 * nsb + 4), 1):
            print(f"   step {i0:3d}: barrier {bar[i0]:6d}  part1 {p1[i0]:6d}  part2+wait {p2[i0]:6d}  total {tot[i0]:6d}")
