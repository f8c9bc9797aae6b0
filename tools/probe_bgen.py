#!/usr/bin/env python3
"""The real-side edge GEMM (generated B operand, gathered K) at the ogbn-arxiv bench shape, with the gathers neutralised one
at a time: which of its three random-row streams (X rows by source id, H1full mask rows by source id, Y1 rows) bounds it."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphslim_amd.engine import gemm
dev = "cuda:0"
N, d, h, C, cap1, f2 = 169343, 128, 256, 40, 2816, 5
n1 = 2300
K = cap1 * f2
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 14
kchunk = (K + nsplit - 1) // nsplit
kchunk = (kchunk + 15) // 16 * 16
g = torch.Generator(device="cpu").manual_seed(0)
X = torch.randn(N, d, device=dev)
H1 = torch.randn(N, h, device=dev)
Y1 = torch.randn(C * cap1, h, device=dev)
w = torch.rand(C * K, device=dev)
slabs = torch.empty(C * nsplit * d * h, device=dev)
kcount = torch.full((C,), n1 * f2, dtype=torch.int32, device=dev)
rand_idx = torch.randint(0, N, (C * K,), generator=g).to(torch.int32).to(dev)
zero_idx = torch.zeros(C * K, dtype=torch.int32, device=dev)
seq_idx = (torch.arange(C * K) % N).to(torch.int32).to(dev)


def run(a_idx, m_idx):
    gemm(X, Y1, slabs, d, h, K, (1, d), (h, 1), h, gatherA_k=a_idx, sGb1=K, nb1=C, nb2=nsplit, kchunk=kchunk, Kb=kcount,
         sBb1=cap1 * h, sCb1=nsplit * d * h, sCb2=d * h, bgen_mask=H1, bgen_ldm=h, bgen_idx=m_idx, bgen_w=w, bgen_div=f2)


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


flops = 2.0 * d * h * n1 * f2 * C
for name, a, m in (("random X rows, random mask rows (as in the engine)", rand_idx, rand_idx),
                   ("random X rows, ONE mask row", rand_idx, zero_idx),
                   ("ONE X row, random mask rows", zero_idx, rand_idx),
                   ("sequential X rows, sequential mask rows", seq_idx, seq_idx),
                   ("ONE X row, ONE mask row", zero_idx, zero_idx)):
    us = t(lambda: run(a, m))
    print(f"nsplit {nsplit}: {name}: {us:.1f} us  {flops / us / 1e6:.1f} TFLOP/s")
