#!/bin/bash
# round 6: with H1's grid on 14/16 of the chip the co-residency knobs of round 5 are swept again (edge slabs aimed at, hoisted share)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
: > $O/stage_times_retune.txt
for ep in 1 352 384 416 448 480 1; do
  echo "== GS_EDGE_PLAN=$ep" >> $O/stage_times_retune.txt
  GS_EDGE_PLAN=$ep timeout 300 python3 tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -E "real-side|outer_step" >> $O/stage_times_retune.txt
done
for hs in 12 13 14 15; do
  echo "== GS_HOIST_SHARE=$hs" >> $O/stage_times_retune.txt
  GS_HOIST_SHARE=$hs timeout 300 python3 tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -E "inner SGC|outer_step" >> $O/stage_times_retune.txt
done
for ps in 13 14; do
  echo "== GS_PLAIN_SHARE=$ps" >> $O/stage_times_retune.txt
  GS_PLAIN_SHARE=$ps timeout 300 python3 tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -E "real-side|outer_step" >> $O/stage_times_retune.txt
done
grep -E "^==|feature epoch, two" $O/stage_times_retune.txt
