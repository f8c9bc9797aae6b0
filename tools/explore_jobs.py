#!/usr/bin/env python3
"""Which whole-job configuration is QUIET enough for a +-0.3 pp job-level gate?  (VERDICT r04 "next" 1c.)  Runs whole GCond jobs of
the HIP engine (tests/test_gpu_reduce.py::_engine_job: its own sampler, the HIP evaluation GCN) over a list of job specs and seeds and
prints mean / standard deviation over seeds of the evaluation accuracy, next to the accuracy of the UN-condensed start (epochs = 0:
picked train rows, no structure) — a job only gates something if condensing moves the accuracy.
usage (MI355X): python tools/explore_jobs.py <n_seeds> '<json job spec>' ['<json job spec>' ...]
  spec keys: workload, scale, epochs, outer, inner, lr_feat, lr_adj, threshold, eval_epochs, run_eval, lr_model"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_gpu_reduce as T
    from graphslim_amd import workloads as W
    n_seeds = int(sys.argv[1])
    for spec in sys.argv[2:]:
        job = dict(scale=1.0, data_seed=1, threshold=0.01, eval_epochs=150, run_eval=3, lr_model=0.01)
        job.update(json.loads(spec))
        p = T.job_problem(job)
        for epochs in (0, job["epochs"]):
            j = dict(job, epochs=epochs)
            t0, acc, loss = time.time(), [], []
            for seed in range(1, n_seeds + 1):
                a, l = T._engine_job(p, j, seed)
                acc.append(np.mean(a))
                loss.append(l)
            acc = np.array(acc)
            print(f"{json.dumps(j)}\n   n'={len(p['labels_syn'])} N={p['data'].num_nodes}: accuracy over {n_seeds} seeds {100 * acc.mean():.2f} +- "
                  f"{100 * acc.std(ddof=1):.2f} % (min {100 * acc.min():.2f} max {100 * acc.max():.2f}), last-epoch loss {np.mean(loss):.4f}; "
                  f"{(time.time() - t0) / n_seeds:.1f} s per job", flush=True)


if __name__ == "__main__":
    main()
