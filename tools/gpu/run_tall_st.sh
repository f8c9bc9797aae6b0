#!/bin/bash
# stage times with the 128-row kernels and the persistent 256-row kernels at subsets of the call sites
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-tall_st}
mkdir -p $O
for cfg in 0 31 24 28 29 30 25; do
  echo "== GS_TALL=$cfg" >> $O/stage_times.txt
  GS_TALL=$cfg timeout 300 python tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -v "^/opt" >> $O/stage_times.txt
done
cat $O/stage_times.txt
