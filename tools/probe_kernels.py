#!/usr/bin/env python3
"""Launches the dominant kernels at the bench shapes a few times (for rocprofv3 --pmc passes)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphslim_amd.engine import gemm, spmm_csr  # noqa: E402

dev = "cuda:0"
n, ph = 454, 256
rows = n * n
a = torch.randn(rows, ph, device=dev)
w = torch.randn(ph, ph, device=dev)
out = torch.empty(rows, ph, device=dev)
for _ in range(5):
    gemm(a, w, out, rows, ph, ph, (ph, 1), (1, ph), ph)          # PGE layer 2 (K,K operand modes)
torch.cuda.synchronize()
# full-graph CSR SpMM at the eval-GCN shape (N = 169343, nnz = 2.48 M, F = 256)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
p = bench.build_problem("ogbn-arxiv")
rp, col, val = p["csr"]
i32 = lambda v: torch.as_tensor(np.asarray(v), dtype=torch.int32, device=dev)
X = torch.randn(p["data"].num_nodes, 256, device=dev)
rpd, cold, vald = i32(rp), i32(col), torch.as_tensor(val, device=dev)
for _ in range(5):
    y = spmm_csr(rpd, cold, vald, X)
torch.cuda.synchronize()
print("probe done", float(out[0, 0]), float(y[0, 0]))
