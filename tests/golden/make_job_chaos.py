#!/usr/bin/env python3
"""How far do two copies of ONE oracle job separate?  The jobs of tests/golden/make_job_accuracy.py are re-run on the same blocks from
X~ (1 + 1e-7 noise) — the perturbation the headline-shape test uses — and the evaluation accuracies / last-epoch losses of the pair
are compared: the yardstick for the engine-vs-oracle differences of tests/test_gpu_reduce.py::test_job_level_accuracy_on_the_oracles_
own_blocks.  Writes tests/golden/job_chaos.json.  usage (build container, CPU): python tests/golden/make_job_chaos.py [first last]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_job_accuracy as J  # noqa: E402
from graphslim_amd import workloads as W  # noqa: E402
from oracle import gcond_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "job_chaos.json")


def main():
    lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 8)
    torch.set_num_threads(os.cpu_count())
    p = W.build_problem(J.JOB["workload"], seed=J.JOB["data_seed"], scale=J.JOB["scale"])
    base = json.load(open(J.OUT))["seeds"]
    blob = json.load(open(OUT)) if os.path.exists(OUT) else {"perturbation": 1e-7, "seeds": {}}
    # the perturbed copy: GCondState is built from x[sel] inside oracle_job — perturb through a wrapped constructor
    orig = O.GCondState

    class Perturbed(orig):
        def __init__(self, feat_syn, *a, **k):
            g = torch.Generator().manual_seed(12345)
            super().__init__(feat_syn * (1 + 1e-7 * torch.randn(feat_syn.shape, generator=g)), *a, **k)

    for seed in range(lo, hi + 1):
        t0 = time.time()
        O.GCondState = Perturbed
        try:
            accs, loss = J.oracle_job(p, seed)
        finally:
            O.GCondState = orig
        ref = base[str(seed)]
        blob["seeds"][str(seed)] = dict(test_acc=accs, last_epoch_loss=loss, ref_test_acc=ref["test_acc"], ref_last_epoch_loss=ref["last_epoch_loss"])
        print(f"seed {seed}: perturbed copy {100 * np.mean(accs):.2f} % vs {100 * np.mean(ref['test_acc']):.2f} %; last-epoch loss {loss:.4f} vs "
              f"{ref['last_epoch_loss']:.4f}; {time.time() - t0:.0f} s", flush=True)
        json.dump(blob, open(OUT, "w"), indent=1)
    d = np.array([np.mean(v["test_acc"]) - np.mean(v["ref_test_acc"]) for v in blob["seeds"].values()])
    r = np.array([abs(v["last_epoch_loss"] - v["ref_last_epoch_loss"]) / v["ref_last_epoch_loss"] for v in blob["seeds"].values()])
    print(f"{len(d)} pairs: accuracy difference mean {100 * d.mean():+.2f} pp, std {100 * d.std(ddof=1):.2f} pp, max |.| {100 * np.abs(d).max():.2f} pp; "
          f"last-epoch loss relative difference max {r.max():.2e}")


if __name__ == "__main__":
    main()
