#!/usr/bin/env python3
"""Time SGDD outer iterations (graphslim/condensation/sgdd.py:64-95) at the reference's configured shapes on the synthetic
dataset-shaped graphs of bench.py: IGNR towers over all n'^2 pairs (width 128), the optimal-transport regulariser (two rocSOLVER
eigen-decompositions of n' x n' matrices, corner size mx_size), gradient matching on the engine.
  python tools/sgdd_bench.py [cora|ogbn-arxiv|reddit] [--steps 20]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

# configs/sgdd/*.json: (ntrans, inner_loop, lr_feat, lr_adj, opt_scale, mx_size)
SGDD = {"cora": (1, 15, 1e-4, 1e-4, 1e-11, 200), "ogbn-arxiv": (2, 3, 0.01, 0.01, 1e-12, 1000), "reddit": (1, 3, 0.1, 0.001, 1e-12, 100)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="cora")
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    from graphslim_amd import host as H
    from graphslim_amd.condensation.sgdd import SGDDLoop
    from graphslim_amd.engine import EngineShapes, GCondEngine
    from graphslim_amd.structure import pinv_corner
    dev = "cuda:0"
    torch.cuda.set_device(0)
    p = bench.build_problem(a.workload)
    T, inner, lrf, lra, scale, mx = SGDD[a.workload]
    data = p["data"]
    d, nclass, n = data.feat_full.shape[1], data.nclass, len(p["labels_syn"])
    H.seed_everything(1)
    params = H.init_ignr_params(d, n, mx)
    sh = EngineShapes(data.bench_n, d, nclass, n, T, p["hidden"], 4, 256, p["fan"][0], p["fan"][1])
    cls_ptr = np.concatenate([[0], np.cumsum([len(q) for q in p["pools"]])])
    eng = GCondEngine(sh, data.bench_feat, p["csr"], cls_ptr, np.concatenate(p["pools"]), p["labels_syn"], p["ranges"], lr_feat=lrf,
                      lr_adj=lra, lr_model=0.01, inner_loop=inner, seed=1, device=dev)
    ids = data.idx_train.numpy() if data.bench_n == data.num_nodes else np.arange(data.bench_n)
    eng.set_feat(data.bench_feat[H.random_init_indices(ids, data.labels_train.numpy(), p["ncd"])])
    t0 = time.time()
    Lx_inv = pinv_corner(data.adj_full[:mx, :mx])
    t_pinv = time.time() - t0
    loop = SGDDLoop(eng, params, Lx_inv, 0.5, scale, lrf, lra)
    eng.set_model(H.init_sgc_params(d, p["hidden"], nclass, T))
    print(f"{a.workload}: n' = {n}, pairs = {n * n}, corner {mx} x {mx} (pseudo-inverse on the host once: {t_pinv:.2f} s), opt_scale {scale}")
    out = {}
    for name, epoch in (("structure epoch", 0), ("feature epoch", 10)):
        for _ in range(3):
            loop.step(epoch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loop.step(epoch)
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / a.steps * 1e3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loop.reg.forward_backward(loop.gen.adj, loop.gen.views()["P"], scale, loop.d_adj, loop.gen.views(loop.gen.grad)["P"])
    torch.cuda.synchronize()
    reg_ms = (time.perf_counter() - t0) / a.steps * 1e3
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loop.gen.forward(loop.x, train=True)
        loop.gen.backward(loop.d_adj)
        loop.gen.forward(loop.x, train=False)
    torch.cuda.synchronize()
    gen_ms = (time.perf_counter() - t0) / a.steps * 1e3
    loss, k = eng.read_loss()
    mix = 0.2 * out["structure epoch"] + 0.8 * out["feature epoch"]
    print(f"  ms / outer iteration: structure epochs {out['structure epoch']:.2f}, feature epochs {out['feature epoch']:.2f} (job mix 20 % / 80 %: "
          f"{mix:.2f} ms = {1e3 / mix:.0f} it/s); of which the IGNR (2 forwards + 1 backward) {gen_ms:.2f} ms, the regulariser (2 eigen-"
          f"decompositions + products) {reg_ms:.2f} ms; mean matching loss {loss / max(k, 1):.4f}, regulariser value {float(loop.reg.value):.1f}")


if __name__ == "__main__":
    main()
