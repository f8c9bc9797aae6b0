#!/bin/bash
# round 5: H1 = relu(X W1 + b1) on the listed source rows also for the full class range (GS_H1_ROWS=2) against all N rows -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-h1rows}
mkdir -p $O
for v in 1 2 1 2; do
  echo "GS_H1_ROWS=$v" >> $O/stage_times_h1_rows_ab.txt
  GS_H1_ROWS=$v timeout 300 python tools/stage_times.py 2>/dev/null | grep -E "real-side|sum of|outer_step|reuse" >> $O/stage_times_h1_rows_ab.txt
done
cat $O/stage_times_h1_rows_ab.txt
