#!/bin/bash
# round 4: job-level accuracy gate against the oracle fixture, and whole ogbn-arxiv-shaped jobs with dW1 | db1 on generated operands
# on / off over paired seeds (VERDICT r03 "next" 3) -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-acc4}
mkdir -p $O
if [ -z "$4" ]; then timeout 900 python -m pytest tests/test_gpu_reduce.py -x -q -m gpu -s -k "job_level_accuracy_against" > $O/test_job_level_accuracy.log 2>&1; tail -4 $O/test_job_level_accuracy.log; fi
for s in $(seq ${3:-1} ${2:-15}); do
  for g in 1 0; do
    GS_DW1_GEN=$g GRAPHSLIM_AMD_SYNTHETIC=1 timeout 200 python tools/full_runs.py ogbn-arxiv:0.005:seed$s 2>/dev/null | grep FULL | sed "s/^FULL/FULL dw1_generated=$g/" >> $O/full_runs_dw1_ab.txt
  done
done
python3 - $O/full_runs_dw1_ab.txt <<'PY' | tee -a $O/full_runs_dw1_ab.txt
import re, sys, numpy as np
acc = {0: {}, 1: {}}
for l in open(sys.argv[1]):
    m = re.search(r"dw1_generated=(\d) gcond seed=(\d+) .* acc ([\d.]+) \+/-", l)
    if m:
        acc[int(m.group(1))][int(m.group(2))] = float(m.group(3))
seeds = sorted(set(acc[0]) & set(acc[1]))
a1, a0 = np.array([acc[1][s] for s in seeds]), np.array([acc[0][s] for s in seeds])
d = a1 - a0
print(f"SUMMARY {len(seeds)} paired seeds: generated {a1.mean():.2f} +- {a1.std(ddof=1):.2f}, materialised {a0.mean():.2f} +- {a0.std(ddof=1):.2f}; "
      f"paired difference {d.mean():+.2f} pp, standard error {d.std(ddof=1) / np.sqrt(len(d)):.2f} pp")
PY
