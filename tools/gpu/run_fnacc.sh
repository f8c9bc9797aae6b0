#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/fnacc
GRAPHSLIM_AMD_SYNTHETIC=1 python3 - <<'PY' > gpurun_out/fnacc/out.txt 2>&1
import os, sys, tempfile, logging
sys.path.insert(0, os.getcwd())
import torch
import graphslim
from graphslim.config import cli
from graphslim.condensation import GCond
from graphslim.dataset import get_dataset
from graphslim.evaluation.eval_agent import Evaluator
from graphslim_amd.host import seed_everything
for mode in ("sampled", "full"):
    for lr in (None, 0.01):
        args = cli(["-D", "reddit", "-M", "gcond", "-R", "0.001", "-G", "0", "--save_path", tempfile.mkdtemp()], standalone_mode=False)
        args.neighbour_mode = mode
        args.epochs, args.checkpoints = 400, [99, 199, 299, 399]
        if lr is not None:
            args.lr_feat = args.lr_adj = lr
        seed_everything(args.seed)
        graph = get_dataset(args.dataset, args)
        lines = []
        class L:
            def info(self, m, *a): lines.append(str(m) % a if a else str(m))
            warning = info
        args.logger = L()
        agent = GCond(setting=args.setting, data=graph, args=args)
        reduced = agent.reduce(graph, verbose=False)
        mean, std = Evaluator(args).evaluate(reduced, model_type="GCN", verbose=False)
        print(f"== {mode} lr_feat={args.lr_feat} lr_adj={args.lr_adj}: final acc {100*mean:.2f} +/- {100*std:.2f}")
        print("   ", [l.strip() for l in lines if "Val" in l or "Test" in l])
        x = reduced.feat_syn; a = reduced.adj_syn
        print(f"    |X~| mean {float(x.abs().mean()):.4f} max {float(x.abs().max()):.3f}; adj mean {float(a.mean()):.4f} >0.5: {float((a>0.5).float().mean()):.3f}")
PY
grep -v "^/opt\|Warning\|warn" gpurun_out/fnacc/out.txt | tail -20
