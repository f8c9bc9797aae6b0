#!/usr/bin/env python3
"""Launches only the PGE layer-2 GEMM (and dY1) a few times — for rocprofv3 --pmc passes on the dominant kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphslim_amd.engine import gemm  # noqa: E402

dev = "cuda:0"
n, ph = 454, 256
rows = n * n
a = torch.randn(rows, ph, device=dev)
w = torch.randn(ph, ph, device=dev)
out = torch.empty(rows, ph, device=dev)
for _ in range(5):
    gemm(a, w, out, rows, ph, ph, (ph, 1), (1, ph), ph)          # PGE layer 2 (K,K operand modes)
for _ in range(5):
    gemm(a, w, out, rows, ph, ph, (ph, 1), (ph, 1), ph)          # dY1 (K,M)
torch.cuda.synchronize()
print("probe done", float(out[0, 0]))
