#!/usr/bin/env python3
"""Solo duration of every stage of one outer iteration (each stage run back to back on an otherwise idle chip, no profiler):
what the iteration would cost with no overlap, where the launch-latency floor sits (launches per stage are listed), and how
much the two-stream overlap hides.  usage: stage_times.py [workload] [operand mode]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ogbn-arxiv"
mode = sys.argv[2] if len(sys.argv) > 2 else None      # None: the engine default (bf16x6)
if os.environ.get("GS_SMALL"):      # routing sweep: "max_tiles,max_gflop" of the latency-oriented 32x32 kernel
    import ctypes as C
    from graphslim_amd import _lib
    t, g = os.environ["GS_SMALL"].split(",")
    _lib.lib().gs_gemm_set_small(C.c_long(int(t)), C.c_double(float(g)))
for env, fn in (("GS_H1_BITS", "gs_gcond_set_h1_bits"), ("GS_EDGE_KERNEL", "gs_gemm_set_edge_kernel"),
                ("GS_FUSED_CHAINS", "gs_gcond_set_fused_chains"), ("GS_H1_ROWS", "gs_gcond_set_h1_rows"), ("GS_AUX_PRIORITY", "gs_gcond_set_aux_priority"),
                ("GS_DW1_GEN", "gs_gcond_set_dw1_generated"), ("GS_EDGE_BITS", "gs_gcond_set_edge_bits"),
                ("GS_EDGE_PLAN", "gs_gcond_set_edge_plan"), ("GS_ELL_XCD", "gs_gcond_set_ell_xcd"),
                ("GS_HOIST_FWD", "gs_gcond_set_hoist_forward_all"), ("GS_PAD_ZERO", "gs_gcond_set_pad_zero"),
                ("GS_HOIST_SHARE", "gs_panel_set_hoist_share"), ("GS_INNER_PERSIST", "gs_gcond_set_inner_persist"),
                ("GS_INNER_GRID", "gs_inner_persist_set_grid"), ("GS_EDGE_SRC", "gs_gcond_set_edge_src"), ("GS_DEDUP_FORM", "gs_dedup_set_form"), ("GS_PLAIN_SHARE", "gs_panel_set_plain_share")):   # A/B knobs of the library
    if os.environ.get(env):
        from graphslim_amd import _lib
        getattr(_lib.lib(), fn)(int(os.environ[env]))
        print(f"{fn}({os.environ[env]})")
if os.environ.get("GS_FWD_PAD") is not None:      # dynamic LDS pad of the training pass's 128-row forward launch (one workgroup per CU above 0)
    from graphslim_amd import _lib
    _lib.lib().gs_panel_set_fwd_lds_pad(int(os.environ["GS_FWD_PAD"]))
    print(f"gs_panel_set_fwd_lds_pad({os.environ['GS_FWD_PAD']})")
if os.environ.get("GS_TALL") is not None:      # 0: the 128-row exact-split kernels of gs_panel.hip; "1,GRID": persistent grid size
    from graphslim_amd import _lib
    v = os.environ["GS_TALL"].split(",")
    _lib.lib().gs_gemm_set_panel_tall(int(v[0]))      # 0 off, 1 default, 16 + mask: sites (1 plain, 2 training forward, 4 inference forward, 8 marginals)
    if len(v) > 1:
        _lib.lib().gs_panel_tall_debug(int(v[1]), 1, int(v[2]) if len(v) > 2 else 0)
    print(f"gs_gemm_set_panel_tall({os.environ['GS_TALL']})")
p = bench.build_problem(name, seed=1)
eng = bench.make_engine(p, "cuda:0")
if mode:
    eng.set_operand_mode(mode)
if os.environ.get("GS_FORK_LATE") is not None:      # 1: the real-side branch is forked AFTER the training forward's long product
    from graphslim_amd import _lib
    _lib.check(_lib.lib().gs_gcond_set_knob(eng.h, 7, int(os.environ["GS_FORK_LATE"])))
    print(f"knob FORK_LATE = {os.environ['GS_FORK_LATE']}")
if os.environ.get("GS_CHAINED") is not None:      # 0: the batched per-class products of the match; 1: the two chained launches
    from graphslim_amd import _lib
    _lib.check(_lib.lib().gs_gcond_set_knob(eng.h, 6, int(os.environ["GS_CHAINED"])))
    print(f"knob CHAINED_PRODUCTS = {os.environ['GS_CHAINED']}")
print(f"workload {name}, operand mode {eng.operand_mode}")
bench.init_state(p, eng)
bench.new_model(p, eng)
step = eng.outer_step
if os.environ.get("GS_SHARD"):      # "rank,world[,layout]": the stages of ONE virtual rank of a sharded run (no exchange): the Amdahl table
    from graphslim_amd.parallel import Comm, ShardedGCond
    v = os.environ["GS_SHARD"].split(",")
    sh = ShardedGCond(eng, int(v[0]), int(v[1]), [len(q) for q in p["pools"]], comm=Comm.null(int(v[0]), int(v[1])), pge=v[2] if len(v) > 2 else None)
    step = sh.outer_step
    print(f"virtual rank {v[0]} of {v[1]} ({sh.pge} layout): classes {sh.class_ranges[int(v[0])]}, pair rows {sh.row_ranges[int(v[0])]}")
for _ in range(3):
    step(10)
torch.cuda.synchronize()


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def fwd(inference):      # (a pair-row shard runs the forward / backward in three parts, with an exchange between them in a real run)
    if os.environ.get("GS_SHARD"):
        return lambda: [eng.pge_forward_part(inference, k) for k in (1, 2, 3)]
    return lambda: eng.stage_pge_forward(inference)


def bwd(param_grads):
    if os.environ.get("GS_SHARD"):
        return lambda: [eng.pge_backward_part(param_grads, True, k) for k in (1, 2, 3)]
    return lambda: eng.stage_pge_backward(param_grads, True)


stages = [
    ("sample (next blocks)", eng.stage_sample),
    ("real-side gradients", eng.stage_real_grads),
    ("PGE forward (training)", fwd(False)),
    ("synthetic match (fwd + class terms + 2nd-order bwd)", eng.stage_syn_match),
    ("loss bookkeeping", eng.syn_finish),
    ("PGE backward (feature epoch)", bwd(False)),
    ("PGE backward (PGE epoch)", bwd(True)),
    ("Adam (feature epoch)", lambda: eng.stage_update(10)),
    ("PGE forward (inference)", fwd(True)),
    ("inner SGC steps", eng.inner_steps),
]
tot = 0.0
for label, fn in stages:
    us = timed(fn)
    if "PGE epoch" not in label and "sample" not in label:
        tot += us
    print(f"{label:52s} {us:8.1f} us")
print(f"sum of the feature-epoch stages (no overlap, sampling excluded): {tot:.1f} us")
it = timed(lambda: step(10), 40)
print(f"outer_step (feature epoch, two streams + prefetched sampling): {it:.1f} us")
it0 = timed(lambda: step(0), 40)
print(f"outer_step (PGE epoch, two streams + prefetched sampling): {it0:.1f} us")
eng.set_forward_reuse(True)      # what GCond.reduce runs: the training forward adopted from the previous iteration's inference forward
itr = timed(lambda: step(10), 40)
itr0 = timed(lambda: step(0), 40)
eng.set_forward_reuse(False)
print(f"outer_step with forward reuse (feature epoch / PGE epoch): {itr:.1f} / {itr0:.1f} us")
