#!/usr/bin/env python3
"""Per-step timeline of the persistent 256-row kernel (csrc/gs_panel2.hip, plain product): four s_memtime stamps per 16-k step of
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
    for half, name, parts in ((0, "wave 0 [loads | prepare | MFMA]", ("loads", "prepare", "MFMA+wait")),
                              (1, "wave 4 [MFMA | loads | prepare]", ("MFMA", "loads", "prepare+wait"))):
        s = t[wg, half]
        n = int((s != 0).sum()) // 4
        if n < 2:
            continue
        st = s[:4 * n].reshape(n, 4)          # arrive, release, after the first piece, after the second piece
        arrive = st[:, 0]
        nxt = np.append(arrive[1:], arrive[-1])
        bar = (st[:, 1] - st[:, 0]) % (1 << 32)
        a = (st[:, 2] - st[:, 1]) % (1 << 32)
        b = (st[:, 3] - st[:, 2]) % (1 << 32)
        c = (nxt - st[:, 3]) % (1 << 32)
        tot = (nxt - arrive) % (1 << 32)
        m = slice(0, n - 1)
        print(f"wg {wg} {name}: {n} steps, {nsb} per tile; cycles per step (median): total {np.median(tot[m]):.0f}, barrier {np.median(bar[m]):.0f}, "
              f"{parts[0]} {np.median(a[m]):.0f}, {parts[1]} {np.median(b[m]):.0f}, {parts[2]} {np.median(c[m]):.0f}")
        for i0 in range(0, min(n - 1, 2 * nsb + 4), 1):
            print(f"   step {i0:3d}: barrier {bar[i0]:6d}  {parts[0]} {a[i0]:6d}  {parts[1]} {b[i0]:6d}  {parts[2]} {c[i0]:6d}  total {tot[i0]:6d}")
