#!/usr/bin/env python3
"""A/B of the GEMM K-tile depth in ONE process (interleaved rounds, median): PGE layer-2 shape, a few small
synthetic-side shapes, and the whole outer iteration."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from graphslim_amd import _lib  # noqa: E402
from graphslim_amd.engine import gemm  # noqa: E402

dev = "cuda:0"
L = _lib.lib()


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


n, ph = 454, 256
a = torch.randn(n * n, ph, device=dev)
w = torch.randn(ph, ph, device=dev)
out = torch.empty(n * n, ph, device=dev)
A = torch.randn(n, n, device=dev)
Q = torch.empty(n, n, device=dev)
X = torch.randn(n, 128, device=dev)
W1 = torch.randn(128, 256, device=dev)
H = torch.empty(n, 256, device=dev)
shapes = {
    "pge_l2 206116x256x256 (K,K)": lambda: gemm(a, w, out, n * n, ph, ph, (ph, 1), (1, ph), ph),
    "dY1    206116x256x256 (K,M)": lambda: gemm(a, w, out, n * n, ph, ph, (ph, 1), (ph, 1), ph),
    "Q=AA   454x454x454": lambda: gemm(A, A, Q, n, n, n, (n, 1), (n, 1), n),
    "XW1    454x256x128": lambda: gemm(X, W1, H, n, 256, 128, (128, 1), (256, 1), 256),
    "QH     454x256x454": lambda: gemm(A, H, out, n, 256, n, (n, 1), (256, 1), 256),
}
res = {}
for rnd in range(3):
    for bs, bb in ((16, 16), (32, 16), (16, 32), (32, 32)):
        _lib.check(L.gs_gemm_set_tuning(bs, bb))
        for k, fn in shapes.items():
            res.setdefault((k, bs, bb), []).append(timeit(fn))
for k in shapes:
    print(k, {f"small{bs}/big{bb}": round(float(np.median(res[(k, bs, bb)])), 1) for bs, bb in ((16, 16), (32, 16), (16, 32), (32, 32))})

p = bench.build_problem("ogbn-arxiv")
eng = bench.make_engine(p, dev)
bench.init_state(p, eng)
bench.new_model(p, eng)
steps = {}
for rnd in range(3):
    for bs, bb in ((16, 16), (32, 16), (16, 32), (32, 32)):
        _lib.check(L.gs_gemm_set_tuning(bs, bb))
        for ep in (0, 10):
            steps.setdefault((bs, bb, ep), []).append(timeit(lambda: eng.outer_step(ep), reps=10))
_lib.check(L.gs_gemm_set_tuning(32, 16))
ov = {}
for rnd in range(3):
    for on in (0, 1):
        eng.set_overlap(on)
        for ep in (0, 10):
            ov.setdefault((on, ep), []).append(timeit(lambda: eng.outer_step(ep), reps=10))
print("overlap off/on (us):", {f"ep{ep}/ov{on}": round(float(np.median(v)), 0) for (on, ep), v in ov.items()})
for ep in (0, 10):
    print("outer step us, epoch", ep, {f"small{bs}/big{bb}": round(float(np.median(steps[(bs, bb, ep)])), 0) for bs, bb in ((16, 16), (32, 16), (16, 32), (32, 32))})
