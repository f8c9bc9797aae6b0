#!/usr/bin/env python3
"""Stage-by-stage difference between the row-panel products on and off (gs_gemm_set_panel) for one workload in one operand mode:
where two forms of the same product part ways.  usage: diag_panel_stages.py [workload] [mode] [epoch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from graphslim_amd import _lib  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cora"
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16x6"
epoch = int(sys.argv[3]) if len(sys.argv) > 3 else 0
L = _lib.lib()
if os.environ.get("GS_MATERIALISE"):      # every intermediate of the PGE backward in memory
    L.gs_gemm_set_panel_marg(0)
    L.gs_gcond_set_dw1_generated(0)
    L.gs_gemm_set_panel_a1gen(0)
p = bench.build_problem(wl, seed=1)
out = {}
for panel in (0, 1):
    L.gs_gemm_set_panel(panel)
    eng = bench.make_engine(p, "cuda:0")
    eng.set_operand_mode(mode)
    bench.init_state(p, eng)
    torch.manual_seed(123)
    bench.new_model(p, eng)
    eng.stage_sample()
    eng.stage_real_grads()
    eng.stage_pge_forward(False)
    torch.cuda.synchronize()
    o = {k: eng.buffer(k).clone() for k in ("E2", "mu2", "is2", "e3", "adj_norm")}
    pg = eng.get_pge()      # [W0, b0, W1, b1, W2, b2, g1, be1, g2, be2]
    n2 = o["E2"].numel() // o["mu2"].numel()
    o["act2"] = (o["E2"].reshape(n2, -1) - o["mu2"]) * o["is2"] * pg[8].to(o["E2"].device) + pg[9].to(o["E2"].device)      # BatchNorm-2 output: ReLU-2 input
    eng.stage_syn_match()
    eng.syn_finish()
    o["d_adj"] = eng.buffer("d_adj").clone()
    eng.stage_pge_backward(epoch % 50 < 10, True)
    torch.cuda.synchronize()
    for k in ("ar_marg", "d_pge", "d_feat", "de3", "W1T") + (("dE2", "A1", "dY1") if os.environ.get("GS_MATERIALISE") else ()):
        o[k] = eng.buffer(k).clone()
    o.update({f"d_pge[{i}]": t.clone() for i, t in enumerate(eng.buffer("d_pge").split([int(torch.tensor(sh).prod()) for sh in eng.pge_shapes()]))})
    out[panel] = o
    del eng
L.gs_gemm_set_panel(1)
for k in out[0]:
    a, b = out[0][k].double(), out[1][k].double()
    print(f"{k:10s} max |panel - general| / max |general| = {float((a - b).abs().max() / a.abs().max()):.3e}   (max {float(a.abs().max()):.3e})")
flips = (out[0]["act2"] > 0) != (out[1]["act2"] > 0)
print(f"ReLU-2 inputs whose SIGN differs between the two forwards: {int(flips.sum())} of {flips.numel()} "
      f"(their |value|: {[float(f'{v:.2e}') for v in out[0]['act2'][flips].abs().tolist()[:8]]})")
if os.environ.get("GS_MATERIALISE"):
    d = (out[0]["dE2"] - out[1]["dE2"]).abs().reshape(flips.shape)
    print(f"dE2: max |diff| on the flipped elements {float(d[flips].max()) if flips.any() else 0.0:.3e}, elsewhere {float(d[~flips].max()):.3e}")
