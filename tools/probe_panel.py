#!/usr/bin/env python3
"""Launches the row-panel kernels at the PGE layer-2 shape — for rocprofv3 --pmc passes: the plain product (5 x fp32,
5 x bf16x3, 5 x bf16), then, through an ogbn-arxiv-shaped engine, the forward with the generated A1 operand
(gs_panel_fwdgen_kernel) and the fused-marginals backward with generated dE2 / regenerated mask (gs_panel_marg_kernel),
5 x per operand mode."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphslim_amd import _lib  # noqa: E402
from graphslim_amd.engine import gemm_desc  # noqa: E402

dev = "cuda:0"
L = _lib.lib()
n, ph = 454, 256
rows = n * n
a = torch.randn(rows, ph, device=dev)
w = torch.randn(ph, ph, device=dev) / 16
b = torch.randn(ph, device=dev)
out = torch.empty(rows, ph, device=dev)
scratch = torch.empty(ph * ph * 3 // 2 + 16, device=dev)
stat = torch.empty(-(-rows // 128), 2, ph, device=dev)
for prec in (0, 3, 1, 2):
    d, keep = gemm_desc(a, w, out, rows, ph, ph, (ph, 1), (1, ph), ph, prec=prec, scratch=scratch, bias=b, colstat=stat)
    assert L.gs_gemm_route(C.byref(d)) == 5
    for _ in range(5):
        _lib.check(L.gs_gemm_f32(C.byref(d), _lib.stream()))
torch.cuda.synchronize()
del a, out
import bench  # noqa: E402
p = bench.build_problem("ogbn-arxiv", seed=1)
eng = bench.make_engine(p, dev)
bench.init_state(p, eng)
bench.new_model(p, eng)
eng.outer_step(10)
for mode in ("f32", "bf16x6", "bf16x3", "bf16"):
    eng.set_operand_mode(mode)
    for _ in range(5):
        eng.pge_forward_part(False, 1)
        eng.pge_forward_part(False, 2)
        eng.pge_backward_part(False, True, 1)
        eng.pge_backward_part(False, True, 2)
torch.cuda.synchronize()
print("probe done")
