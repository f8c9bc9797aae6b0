#!/bin/bash
# round 6: the entry-form edge product once more, on the final build (H1's grid on 14/16: the main stream idles while the edge product runs)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
: > $O/edge_src_ab3.txt
for src in 0 1 0 1; do
  echo "== GS_EDGE_SRC=$src" >> $O/edge_src_ab3.txt
  GS_EDGE_SRC=$src timeout 300 python3 tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -E "real-side|outer_step" >> $O/edge_src_ab3.txt
done
cat $O/edge_src_ab3.txt
