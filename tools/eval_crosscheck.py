#!/usr/bin/env python3
"""Is the +2 pp of the job-level comparison (profiles/r04) in the CONDENSATION or in the EVALUATION?  The engine condenses the
arxiv-gate jobs (own sampler); each condensed graph is then evaluated twice with the same seeds: by the HIP evaluation GCN
(graphslim_amd/models/gcn.py, what the product and the GPU tests use) and by the oracle's CPU evaluation (oracle.run_eval_gcn, what
the fixture's accuracies were computed with).  usage (GPU box): eval_crosscheck.py [n_seeds]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_reduce as T  # noqa: E402
from graphslim_amd import workloads as W  # noqa: E402
from graphslim_amd.host import init_pge_params, init_sgc_params, random_init_indices, seed_everything  # noqa: E402
from oracle import gcond_oracle as O  # noqa: E402

blob = json.load(open(os.path.join(ROOT, "tests", "golden", "job_accuracy.json")))
job = blob["job"]
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
torch.set_num_threads(min(os.cpu_count() or 1, 32))
p = W.build_problem(job["workload"], seed=job["data_seed"], scale=job["scale"])
d = p["data"]
rp, col, val = p["csr"]
x, y = d.feat_full, d.labels_full
g = dict(feat_full=x, adj_full_norm=O.csr_to_torch(rp, col, val, d.num_nodes), idx_val=d.idx_val, labels_val=y[d.idx_val],
         idx_test=d.idx_test, labels_test=y[d.idx_test])
diffs = []
for seed in range(1, nseeds + 1):
    eng = W.make_engine(p, T.DEV)
    seed_everything(seed)
    sel = random_init_indices(d.idx_train.numpy(), d.labels_train.numpy(), p["ncd"])
    eng.set_feat(d.feat_full[sel])
    eng.set_pge(init_pge_params(x.shape[1], p["ph"]))
    for epoch in range(job["epochs"]):
        eng.set_model(init_sgc_params(x.shape[1], p["hidden"], d.nclass, p["T"]))
        for _ in range(job["outer"]):
            eng.outer_step(epoch)
    torch.cuda.synchronize()
    feat, adj = eng.feat_syn.clone(), eng.adj_syn.clone()
    del eng
    hip = T._eval_runs_cfg(p, feat, adj, seed, job)
    adj_c = O.sparsify(adj.cpu().clone(), job["threshold"])
    cpu = []
    for i in range(job["run_eval"]):
        seed_everything(seed + i)
        cpu.append(float(O.run_eval_gcn(feat.cpu(), adj_c, torch.from_numpy(p["labels_syn"]), g, p["hidden"], d.nclass, job["eval_epochs"])[1]))
    diffs.append(np.mean(hip) - np.mean(cpu))
    print(f"seed {seed}: same condensed graph — HIP evaluation {100 * np.mean(hip):.2f} % {[round(100 * a, 2) for a in hip]}, "
          f"oracle CPU evaluation {100 * np.mean(cpu):.2f} % {[round(100 * a, 2) for a in cpu]}", flush=True)
diffs = np.array(diffs)
print(f"{nseeds} graphs: HIP - CPU evaluation {100 * diffs.mean():+.2f} pp (std {100 * diffs.std(ddof=1):.2f} pp)")
