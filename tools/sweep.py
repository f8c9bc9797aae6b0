#!/usr/bin/env python3
"""Multi-dataset plumbing sweep (BASELINE.json configs[0..2]): cli -> get_dataset -> GCond.reduce -> Evaluator.evaluate
on dataset-shaped synthetic graphs with a reduced epoch budget.  Prints one line per dataset."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import graphslim  # noqa: E402,F401
from graphslim.config import cli  # noqa: E402
from graphslim.condensation import GCond  # noqa: E402
from graphslim.dataset import get_dataset  # noqa: E402
from graphslim.evaluation.eval_agent import Evaluator  # noqa: E402

runs = [("cora", "0.5"), ("cora", "0.25"), ("citeseer", "0.5"), ("citeseer", "0.005"), ("pubmed", "0.5"), ("pubmed", "0.005"),
        ("flickr", "0.01")]
if len(sys.argv) > 1:
    runs = [tuple(a.split(":")) for a in sys.argv[1:]]
for ds, rate in runs:
    tmp = tempfile.mkdtemp(prefix="gs_sweep_")
    if ds == "flickr":
        os.environ["GRAPHSLIM_AMD_SCALE"] = "0.2"
    else:
        os.environ.pop("GRAPHSLIM_AMD_SCALE", None)
    args = cli(["-D", ds, "-M", "gcond", "-R", rate, "-G", "0", "--epochs", "40", "--eval_epochs", "100", "--run_eval", "3",
                "--save_path", tmp], standalone_mode=False)
    args.run_inter_eval = 1
    t0 = time.time()
    graph = get_dataset(args.dataset, args)
    t1 = time.time()
    agent = GCond(setting=args.setting, data=graph, args=args)
    reduced = agent.reduce(graph, verbose=False)
    torch.cuda.synchronize()
    t2 = time.time()
    mean, std = Evaluator(args).evaluate(reduced, model_type="GCN", verbose=False)
    t3 = time.time()
    iters = args.epochs * args.outer_loop
    print(f"SWEEP {ds} r={rate} setting={args.setting} N={graph.num_nodes} d={graph.feat_full.shape[1]} C={graph.nclass} n'={agent.nnodes_syn} "
          f"T={args.ntrans} outer={args.outer_loop} inner={args.inner_loop}: data {t1 - t0:.1f}s reduce {t2 - t1:.1f}s "
          f"({iters} outer iters incl. 4 checkpoints) eval {t3 - t2:.1f}s acc {100 * mean:.2f} +/- {100 * std:.2f}", flush=True)
