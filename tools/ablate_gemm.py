#!/usr/bin/env python3
"""Ablation of the big fp32-MFMA GEMM at the PGE layer-2 shape (one process, interleaved)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphslim_amd import _lib
from graphslim_amd.engine import gemm
L = _lib.lib()
dev = "cuda:0"
n, ph = 454, 256
a = torch.randn(n * n, ph, device=dev); w = torch.randn(ph, ph, device=dev); out = torch.empty(n * n, ph, device=dev)
def run(): gemm(a, w, out, n * n, ph, ph, (ph, 1), (1, ph), ph)
def t(reps=20):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
names = {0: "full", 1: "no-loads", 2: "no-stores", 3: "no-loads,no-stores", 4: "no-mfma", 6: "no-mfma,no-stores", 5: "no-mfma,no-loads", 7: "nothing"}
res = {}
for rnd in range(3):
    for bits in names:
        L.gs_gemm_set_debug(bits)
        res.setdefault(bits, []).append(t())
L.gs_gemm_set_debug(0)
for bits, nm in names.items():
    print(f"{nm:22s} {np.median(res[bits]):7.1f} us")
