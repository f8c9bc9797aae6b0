#!/usr/bin/env python3
"""Same-device A/B of the PGE operand forms of round 2 (interleaved, two rounds), ogbn-arxiv shape, per operand mode:
  fused marginals level 0 / 1 / 2   (gs_gemm_set_panel_marg: dY1 materialised / sums in the product's epilogue / + generated dE2)
  generated A1 off / on             (gs_gemm_set_panel_a1gen)
  folded reduction stages off / on  (gs_gcond_set_folded_reductions)
Prints ms per feature-epoch and per PGE-epoch outer iteration."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from graphslim_amd import _lib  # noqa: E402

L = _lib.lib()
p = bench.build_problem("ogbn-arxiv", seed=1)
eng = bench.make_engine(p, "cuda:0")
bench.init_state(p, eng)
bench.new_model(p, eng)


def ms(epoch, reps=40):
    for _ in range(5):
        eng.outer_step(epoch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.outer_step(epoch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def reset():
    L.gs_gemm_set_panel_marg(2)
    L.gs_gemm_set_panel_a1gen(1)
    L.gs_gcond_set_folded_reductions(1)


for mode in ("f32", "bf16x6", "bf16x3", "bf16"):
    eng.set_operand_mode(mode)
    for rnd in range(2):
        for label, setup in (("round-2 first form: dY1 materialised, A1 materialised", lambda: (L.gs_gemm_set_panel_marg(0), L.gs_gemm_set_panel_a1gen(0))),
                             ("+ marginals in the product's epilogue", lambda: (L.gs_gemm_set_panel_marg(1), L.gs_gemm_set_panel_a1gen(0))),
                             ("+ dE2 generated from E2 (feature epochs)", lambda: (L.gs_gemm_set_panel_marg(2), L.gs_gemm_set_panel_a1gen(0))),
                             ("+ A1 generated from U, V (forward operand, backward mask)", lambda: None),
                             ("  (same, reduction stages as separate launches)", lambda: L.gs_gcond_set_folded_reductions(0))):
            reset()
            setup()
            print(f"{mode:7s} {label:62s} feature epoch {ms(10):.3f} ms   PGE epoch {ms(0):.3f} ms", flush=True)
reset()
