#!/usr/bin/env python3
"""Whole jobs at the product configurations (examples/train_gcond.py's five calls, configs/gcond/*.json) on the
dataset-shaped synthetic graphs: wall time of reduce (all epochs, checkpoints with intermediate evaluation) and of the final
evaluation, and the accuracy.  usage: full_runs.py dataset:rate[:full][:bf16x6|:bf16x3|:bf16][:msgc][:seedN] [...]   (":full" =
args.neighbour_mode = "full", the exact 2-hop neighbourhoods of SURVEY 8e mode F, one-transform configs only; ":bf16x6" etc. =
args.operand_mode; ":msgc" / ":sgdd" = the MSGC / SGDD method with configs/<method>/*.json instead of GCond)"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphslim  # noqa: F401
from graphslim.config import cli
from graphslim.condensation import GCond, MSGC, SGDD
from graphslim.dataset import get_dataset
from graphslim.evaluation.eval_agent import Evaluator
from graphslim_amd.host import seed_everything

for env, fn in (("GS_DW1_GEN", "gs_gcond_set_dw1_generated"), ("GS_EDGE_KERNEL", "gs_gemm_set_edge_kernel")):      # A/B knobs of the library
    if os.environ.get(env):
        from graphslim_amd import _lib
        getattr(_lib.lib(), fn)(int(os.environ[env]))
        print(f"{fn}({os.environ[env]})", flush=True)
for spec in sys.argv[1:]:
    ds, rate, *opt = spec.split(":")
    method = "msgc" if "msgc" in opt else ("sgdd" if "sgdd" in opt else "gcond")
    args = cli(["-D", ds, "-M", method, "-R", rate, "-G", "0", "--save_path", tempfile.mkdtemp(prefix="gs_full_")], standalone_mode=False)
    if "full" in opt:
        args.neighbour_mode = "full"
    for m in ("f32", "bf16x6", "bf16x3", "bf16"):
        if m in opt:
            args.operand_mode = m
    for o in opt:
        if o.startswith("seed"):
            args.seed = int(o[4:])
        if o.startswith("optscale"):          # SGDD: override configs/sgdd/<dataset>.json's opt_scale (":optscale0" = no regulariser)
            args.opt_scale = float(o[8:])
    seed_everything(args.seed)
    t0 = time.time(); graph = get_dataset(args.dataset, args); t1 = time.time()
    agent = {"msgc": MSGC, "sgdd": SGDD, "gcond": GCond}[method](setting=args.setting, data=graph, args=args)
    reduced = agent.reduce(graph, verbose=False); torch.cuda.synchronize(); t2 = time.time()
    mean, std = Evaluator(args).evaluate(reduced, model_type="GCN", verbose=False); t3 = time.time()
    t4 = time.time()
    keep = args.run_eval
    args.run_eval = 3
    wmean, wstd = Evaluator(args).evaluate(graph, model_type="GCN", verbose=False, reduced=False)   # whole-dataset baseline
    args.run_eval = keep
    t5 = time.time()
    print(f"FULL {method} seed={args.seed} {ds}{' [full-neighbour mode]' if 'full' in opt else ''}{' [operand mode ' + (getattr(args, 'operand_mode', None) or 'default bf16x6') + ']'} r={rate} setting={args.setting} N={graph.num_nodes} n'={agent.nnodes_syn} epochs={args.epochs} outer={args.outer_loop} "
          f"inner={args.inner_loop}: data {t1 - t0:.1f}s  reduce {t2 - t1:.1f}s ({args.epochs * args.outer_loop} outer iterations, "
          f"{(args.epochs * args.outer_loop) / (t2 - t1):.0f} it/s incl. checkpoints)  final eval ({args.run_eval} runs x {args.eval_epochs} epochs) "
          f"{t3 - t2:.1f}s  acc {100 * mean:.2f} +/- {100 * std:.2f}  | GCN trained on the ORIGINAL graph (3 runs, {t5 - t4:.1f}s): "
          f"{100 * wmean:.2f} +/- {100 * wstd:.2f}", flush=True)
