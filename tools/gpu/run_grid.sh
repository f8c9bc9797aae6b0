#!/bin/bash
# persistent kernels on fewer than all CUs (7/8, 3/4 of them), default sites and all sites
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-grid1}
mkdir -p $O
for cfg in "29 0" "29 224" "31 224" "31 192" "29 192" "31 0" "29 0"; do
  set -- $cfg
  echo "== tall sites=$1 grid=$2" | tee -a $O/grid.txt
  GS_TALL=$1,$2 timeout 300 python tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -E "PGE forward|real-side|PGE backward \(feature|outer_step \(feature" | tee -a $O/grid.txt
done
