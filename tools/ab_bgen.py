#!/usr/bin/env python3
"""A/B of the real-side edge product at the ogbn-arxiv shape: 128x128 tile vs the 128x256 tile (A gathered once), several K-chunk sizes."""
import ctypes as C
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphslim_amd import _lib
from graphslim_amd.engine import gemm
dev = "cuda:0"
L = _lib.lib()
N, d, h, Cc, cap1, f2 = 169343, 128, 256, 40, 2816, 5
n1 = 2300
K = cap1 * f2
g = torch.Generator(device="cpu").manual_seed(0)
X = torch.randn(N, d, device=dev); H1 = torch.randn(N, h, device=dev); Y1 = torch.randn(Cc * cap1, h, device=dev)
w = torch.rand(Cc * K, device=dev)
kcount = torch.full((Cc,), n1 * f2, dtype=torch.int32, device=dev)
idx = torch.randint(0, N, (Cc * K,), generator=g).to(torch.int32).to(dev)


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


flops = 2.0 * d * h * n1 * f2 * Cc
VAR = ["tile128", "edge f32", "edge bf16x3", "edge bf16"]
ref = None
for kchunk in (256, 384, 512, 768, 1024):
    nsplit = -(-K // kchunk)
    slabs = torch.zeros(Cc * nsplit * d * h, device=dev)
    def run():
        gemm(X, Y1, slabs, d, h, K, (1, d), (h, 1), h, gatherA_k=idx, sGb1=K, nb1=Cc, nb2=nsplit, kchunk=kchunk, Kb=kcount,
             sBb1=cap1 * h, sCb1=nsplit * d * h, sCb2=d * h, bgen_mask=H1, bgen_ldm=h, bgen_idx=idx, bgen_w=w, bgen_div=f2)
    def run_p(prec):
        gemm(X, Y1, slabs, d, h, K, (1, d), (h, 1), h, gatherA_k=idx, sGb1=K, nb1=Cc, nb2=nsplit, kchunk=kchunk, Kb=kcount,
             sBb1=cap1 * h, sCb1=nsplit * d * h, sCb2=d * h, bgen_mask=H1, bgen_ldm=h, bgen_idx=idx, bgen_w=w, bgen_div=f2, prec=prec)
    for rnd in range(2):
        for wide in (0, 1, 2, 3):
            L.gs_gemm_set_edge_kernel(1 if wide else 0)
            us = t(run) if wide < 2 else t(lambda: run_p(wide - 1))
            tot = slabs.view(Cc, nsplit, d * h).sum(1)
            if ref is None: ref = tot.clone()
            err = float((tot - ref).abs().max() / ref.abs().max())
            print(f"kchunk {kchunk} ({nsplit} slabs) variant={VAR[wide]}: {us:.1f} us  {flops / us / 1e6:.1f} TFLOP/s  rel diff vs first {err:.1e}")
L.gs_gemm_set_edge_kernel(1)
