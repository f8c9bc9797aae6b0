#!/usr/bin/env python3
"""Per-workgroup placement and phase timeline of the row-panel kernel (fp32, PGE layer-2 shape): which workgroups share a CU,
when each one starts, ends its K loop and ends its stores."""
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
M, K = 454 * 454, 256
a = torch.randn(M, K, device=dev)
w = torch.randn(256, K, device=dev) / 16
out = torch.empty(M, 256, device=dev)
nt = -(-M // 32)      # upper bound of the workgroup count (the tail runs 32-row tiles)
for deph in (0, 1):
    trace = torch.zeros(nt * 5, dtype=torch.int64, device=dev)
    L.gs_gemm_set_panel_narrow_tail(deph)
    L.gs_panel_debug(1, 2)
    d, keep = gemm_desc(a, w, out, M, 256, K, (K, 1), (1, K), 256, prec=0, scratch=trace)
    for _ in range(2):
        _lib.check(L.gs_gemm_f32(C.byref(d), _lib.stream()))
    torch.cuda.synchronize()
    L.gs_panel_debug(1, 0)
    t = trace.cpu().numpy().reshape(nt, 5)
    t = t[t[:, 4] != 0]
    hw, xcc = t[:, 0], t[:, 1] & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    key = (xcc << 8) | (se << 5) | (sh << 4) | cu
    t0 = t[:, 2].min()
    st, lp, en = (t[:, 2] - t0) / 100.0, (t[:, 3] - t0) / 100.0, (t[:, 4] - t0) / 100.0     # us
    print(f"--- narrow tail {deph}: {len(np.unique(key))} distinct CUs, kernel span {en.max():.1f} us")
    print("first 24 workgroups: id xcc se sh cu start loop_end end")
    for b in list(range(6)) + list(range(256, 262)) + list(range(len(t) - 6, len(t))):
        print(f"  {b:5d} {xcc[b]:2d} {se[b]:2d} {sh[b]:2d} {cu[b]:3d}   {st[b]:7.1f} {lp[b]:7.1f} {en[b]:7.1f}")
    # which block ids share the CU of block 0 .. 3
    for b in range(4):
        mates = np.nonzero(key == key[b])[0]
        print(f"  CU of block {b}: blocks {mates.tolist()}  starts {[round(float(st[m]), 1) for m in mates]}")
    dur_loop, dur_store = lp - st, en - lp
    print(f"  K-loop duration  mean {dur_loop.mean():.1f} us (min {dur_loop.min():.1f}, max {dur_loop.max():.1f}); "
          f"epilogue mean {dur_store.mean():.1f} us (min {dur_store.min():.1f}, max {dur_store.max():.1f})")
    # concurrency over time: workgroups in their K loop / in their epilogue
    grid = np.arange(0, en.max(), 5.0)
    inloop = [(int(((st <= g) & (lp > g)).sum()), int(((lp <= g) & (en > g)).sum())) for g in grid]
    print("  t[us]: (#in K loop, #in epilogue)  " + "  ".join(f"{int(g)}:{a_}/{b_}" for g, (a_, b_) in zip(grid, inloop)))
L.gs_gemm_set_panel_narrow_tail(1)
