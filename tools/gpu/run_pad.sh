#!/bin/bash
# training-pass forward at one workgroup per CU (dynamic LDS pad) beside the real-side stream; H1 on the persistent or the 128-row kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-pad1}
mkdir -p $O
for cfg in "0 29" "16384 29" "16384 28" "0 28" "0 29" "16384 29"; do
  set -- $cfg
  echo "== FWD_PAD=$1 tall sites=$2" | tee -a $O/pad.txt
  GS_FWD_PAD=$1 GS_TALL=$2 timeout 300 python tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -E "PGE forward \(training\)|real-side|outer_step|sum of" | tee -a $O/pad.txt
done
