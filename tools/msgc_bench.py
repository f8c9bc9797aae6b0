#!/usr/bin/env python3
"""Time MSGC outer iterations (graphslim/condensation/msgc.py:95-118) at the reference's configured shapes on the synthetic
dataset-shaped graphs of bench.py: batch_adj = 16 structures, the edge MLP (hidden 256) over all directed edges, gradient
matching on the block-diagonal graph of 16 n' rows.  Prints ms per iteration for structure epochs and feature epochs and the
split generator / engine.   python tools/msgc_bench.py [cora|ogbn-arxiv|reddit] [--batch 16] [--steps 20]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

# configs/msgc/*.json: (ntrans, inner_loop, lr)
MSGC = {"cora": (1, 15, 1e-4), "ogbn-arxiv": (2, 3, 0.01), "reddit": (2, 1, 0.1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="cora")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    from graphslim_amd import host as H
    from graphslim_amd.condensation.msgc import MSGCLoop
    from graphslim_amd.engine import EngineShapes, GCondEngine
    from graphslim_amd.structure import BatchLayout
    dev = "cuda:0"
    torch.cuda.set_device(0)
    p = bench.build_problem(a.workload)
    T, inner, lr = MSGC[a.workload]
    data = p["data"]
    d, nclass = data.feat_full.shape[1], data.nclass
    n_syn = int(data.feat_train.shape[0] * p["rate"])
    H.seed_everything(1)
    y_syn, each = H.msgc_labels(data.labels_train.numpy(), n_syn, nclass)
    params = H.init_edge_mlp_params(d)
    t0 = time.time()
    rows, cols, batch = H.msgc_edges(y_syn, nclass, a.batch)
    t_edges = time.time() - t0
    lay = BatchLayout(y_syn, nclass, a.batch)
    sh = EngineShapes(data.bench_n, d, nclass, lay.n_big, T, p["hidden"], 4, 256, p["fan"][0], p["fan"][1])
    cls_ptr = np.concatenate([[0], np.cumsum([len(q) for q in p["pools"]])])
    eng = GCondEngine(sh, data.bench_feat, p["csr"], cls_ptr, np.concatenate(p["pools"]), lay.labels_big, lay.ranges, lr_feat=lr,
                      lr_adj=lr, lr_model=0.01, inner_loop=inner, seed=1, device=dev)
    ids = data.idx_train.numpy() if data.bench_n == data.num_nodes else np.arange(data.bench_n)
    sel = H.random_init_indices(ids, data.labels_train.numpy(), {c: int(k) for c, k in enumerate(each)})
    loop = MSGCLoop(eng, params, rows, cols, batch, lay, data.bench_feat[sel], lr, lr)
    eng.set_model(H.init_sgc_params(d, p["hidden"], nclass, T))
    print(f"{a.workload}: n' = {n_syn}, B = {a.batch}, rows of the block-diagonal graph = {lay.n_big}, directed edges = {len(rows)} "
          f"({len(rows) / a.batch / n_syn:.1f} per node and structure), edge lists built on the host in {t_edges:.1f} s, "
          f"engine workspace {eng.workspace_bytes / 2**30:.2f} GiB")
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
    # the generator alone: forward + backward + forward (what one iteration runs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loop.gen.forward(loop.x)
        loop.gen.backward(loop.d_adj)
        loop.gen.forward(loop.x)
    torch.cuda.synchronize()
    gen_ms = (time.perf_counter() - t0) / a.steps * 1e3
    loss, k = eng.read_loss()
    print(f"  ms / outer iteration: structure epochs {out['structure epoch']:.2f}, feature epochs {out['feature epoch']:.2f} "
          f"(job mix 20 % / 80 %: {0.2 * out['structure epoch'] + 0.8 * out['feature epoch']:.2f} ms = "
          f"{1e3 / (0.2 * out['structure epoch'] + 0.8 * out['feature epoch']):.0f} it/s); of which the edge-list generator "
          f"(2 forwards + 1 backward) {gen_ms:.2f} ms; mean matching loss {loss / max(k, 1):.4f}")


if __name__ == "__main__":
    main()
