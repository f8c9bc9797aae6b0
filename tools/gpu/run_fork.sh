#!/bin/bash
# real-side branch forked before (default) / after the training forward's long product, with the 128-row or the persistent 256-row
# kernel at that site (tall site masks: 16 + 13 = default sites, 16 + 15 = + training forward)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-fork1}
mkdir -p $O
for cfg in "0 29" "1 29" "1 31" "0 31" "0 29" "1 31"; do
  set -- $cfg
  echo "== FORK_LATE=$1 tall sites=$2" | tee -a $O/fork.txt
  GS_FORK_LATE=$1 GS_TALL=$2 timeout 300 python tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -E "PGE forward \(training\)|real-side|outer_step|sum of" | tee -a $O/fork.txt
done
