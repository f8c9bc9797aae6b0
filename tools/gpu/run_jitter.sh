#!/bin/bash
# round 5: run-to-run / group-to-group spread of the fused step with and without the hoisted forward -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-jitter}
mkdir -p $O
for i in 1 2 3 4; do
  for v in 1 0; do GS_HOIST_FWD=$v timeout 300 python tools/step_jitter.py 40 2>/dev/null | tee -a $O/step_jitter.txt; done
done
