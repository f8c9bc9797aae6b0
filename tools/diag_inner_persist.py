#!/usr/bin/env python3
"""Persistent inner-step launch against the launch chain AND against an fp64 evaluation of the first step's gradient on the host:
which of the two is closer.  usage (MI355X): diag_inner_persist.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from graphslim_amd import _lib  # noqa: E402
from graphslim_amd.host import init_sgc_params  # noqa: E402

L = _lib.lib()
KNOB = _lib.KNOBS["INNER_PERSIST"]
for workload, scale, inner in (("ogbn-arxiv", 1.0, 1), ("ogbn-arxiv", 0.25, 1), ("arxiv-gate", 0.05, 1), ("flickr", 0.5, 1), ("arxiv-gate-c8", 0.05, 1),
                               ("ogbn-arxiv", 0.25, 3)):
    p = bench.build_problem(workload, seed=1, scale=scale)
    p["inner"] = inner
    d, C = p["data"].feat_full.shape[1], p["data"].nclass
    torch.manual_seed(5)
    theta = init_sgc_params(d, p["hidden"], C, p["T"])
    res = {}
    for on in (0, 1):
        eng = bench.make_engine(p, "cuda:0")
        _lib.check(L.gs_gcond_set_knob(eng.h, KNOB, 2 * on))
        bench.init_state(p, eng)
        eng.set_model(theta)
        eng.stage_pge_forward(True)
        A = eng.buffer("adj_norm_inner", (eng.shapes.n_syn, eng.shapes.n_syn)).double()
        X = eng.feat_syn.cpu().double()
        eng.inner_steps()
        torch.cuda.synchronize()
        res[on] = (eng.model_params.cpu().clone(), eng.buffer("m_model").clone(), eng.buffer("v_model").clone(), eng.buffer("g_model").clone())
        eng.read_loss()
    # fp64 gradient of the first step
    W1, b1, Wl, bl = [t.double().clone().requires_grad_(True) for t in theta]
    Q = A @ A
    y = torch.from_numpy(p["labels_syn"]).long()
    out = Q @ (torch.relu(X @ W1 + b1) @ Wl) + Q.sum(1, keepdim=True) * bl
    loss = torch.nn.functional.nll_loss(torch.log_softmax(out, 1), y)
    loss.backward()
    g64 = torch.cat([t.grad.reshape(-1) for t in (W1, b1, Wl, bl)])
    gmax = float(g64.abs().max())
    line = f"{workload} scale {scale} inner {inner}: n'={eng.shapes.n_syn} d={d} C={C}"
    if inner == 1:
        for on in (0, 1):
            g = res[on][3].double()
            line += f" | {'persist' if on else 'chain'} g vs fp64: max {float((g - g64).abs().max()) / gmax:.2e} of gmax"
    for k, nm in ((1, "m"), (2, "v"), (0, "theta")):
        a, b = res[0][k], res[1][k]
        line += f" | {nm} chain-persist {float((a - b).abs().max()) / float(a.abs().max()):.2e}"
    print(line, flush=True)
