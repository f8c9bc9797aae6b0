#!/usr/bin/env python3
"""A/B of the latency-bound GEMM routing (gs_gemm_set_small) in ONE process (interleaved rounds, median): synthetic-side
shapes, correctness of the routed kernel against the tiled one, and the whole outer iteration."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from graphslim_amd import _lib  # noqa: E402
from graphslim_amd.engine import gemm  # noqa: E402

dev = "cuda:0"
L = _lib.lib()
CFG = {"off": (0, 0.0), "on": (8192, 1.5)}
if len(sys.argv) > 1:
    for a in sys.argv[1:]:
        t, g = a.split(":")
        CFG[f"{t}/{g}"] = (int(t), float(g))


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


n, d, h, C = 454, 128, 256, 40
A = torch.randn(n, n, device=dev)
X = torch.randn(n, d, device=dev)
W1 = torch.randn(d, h, device=dev)
W2 = torch.randn(h, C, device=dev)
H = torch.randn(n, h, device=dev)
G = torch.randn(n, C, device=dev)
o_nn, o_nd, o_nh, o_nc, o_hc, o_dh = (torch.empty(*s, device=dev) for s in ((n, n), (n, d), (n, h), (n, C), (h, C), (d, h)))
shapes = {
    "S1=A X      454x128x454 (K,M)": lambda: gemm(A, X, o_nd, n, d, n, (n, 1), (d, 1), d),
    "H=S1 W1     454x256x128 (K,M)": lambda: gemm(X, W1, o_nh, n, h, d, (d, 1), (h, 1), h),
    "S2=A H      454x256x454 (K,M)": lambda: gemm(A, H, o_nh, n, h, n, (n, 1), (h, 1), h),
    "Z=S2 W2     454x40x256  (K,M)": lambda: gemm(H, W2, o_nc, n, C, h, (h, 1), (C, 1), C),
    "T=G W2^T    454x256x40  (K,K)": lambda: gemm(G, W2, o_nh, n, h, C, (C, 1), (1, C), h),
    "dA=dS H^T   454x454x256 (K,K)": lambda: gemm(H, H, o_nn, n, n, h, (h, 1), (1, h), n),
    "dH=A^T dS   454x256x454 (M,M)": lambda: gemm(A, H, o_nh, n, h, n, (1, n), (h, 1), h),
    "gW2=S^T G   256x40x454  (M,M)": lambda: gemm(H, G, o_hc, h, C, n, (1, h), (C, 1), C),
    "gW1=S1^T dH 128x256x454 (M,M)": lambda: gemm(X, H, o_dh, d, h, n, (1, d), (h, 1), h),
}
# correctness of the routed kernel vs the tiled kernel and vs torch
for k, fn in shapes.items():
    outs = []
    for name in ("off", "on"):
        _lib.check(L.gs_gemm_set_small(*CFG[name]))
        fn()
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (o_nn, o_nd, o_nh, o_nc, o_hc, o_dh)])
    err = max(float((a - b).abs().max() / (a.abs().max() + 1e-30)) for a, b in zip(*outs))
    print(f"{k}: max rel diff small vs tiled {err:.2e}")
res = {}
for rnd in range(3):
    for name, cfg in CFG.items():
        _lib.check(L.gs_gemm_set_small(*cfg))
        for k, fn in shapes.items():
            res.setdefault((k, name), []).append(timeit(fn))
for k in shapes:
    print(k, {name: round(float(np.median(res[(k, name)])), 1) for name in CFG})

p = bench.build_problem("ogbn-arxiv")
eng = bench.make_engine(p, dev)
bench.init_state(p, eng)
bench.new_model(p, eng)
steps = {}
for rnd in range(3):
    for name, cfg in CFG.items():
        _lib.check(L.gs_gemm_set_small(*cfg))
        for ep in (0, 10):
            steps.setdefault((name, ep), []).append(timeit(lambda: eng.outer_step(ep), reps=10))
_lib.check(L.gs_gemm_set_small(*CFG["on"]))
for ep in (0, 10):
    print("outer step us, epoch", ep, {name: round(float(np.median(steps[(name, ep)])), 0) for name in CFG})
