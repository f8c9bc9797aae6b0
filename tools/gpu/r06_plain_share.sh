#!/bin/bash
# round 6: does the match head start earlier when H1 = relu(X W1 + b1) leaves compute units free?  (the persistent product holds every
# CU's register file: the main stream's first launches of an iteration wait ~145 us for it, profiles/r06/timeline_noreuse.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
: > $O/stage_times_plain_share.txt
for sh in 16 15 14 12 16 14; do
  echo "== GS_PLAIN_SHARE=$sh" >> $O/stage_times_plain_share.txt
  GS_PLAIN_SHARE=$sh timeout 300 python3 tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -E "real-side|outer_step" >> $O/stage_times_plain_share.txt
done
cat $O/stage_times_plain_share.txt
