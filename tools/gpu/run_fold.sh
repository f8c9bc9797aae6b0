#!/bin/bash
# launches folded into the PGE passes: bitwise test + stage times, then the tests that touch the PGE passes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-fold1}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "folded_into or fused_chains or chained or match or dist" > $O/test_fold.log 2>&1; tail -15 $O/test_fold.log
for r in 1 2; do
  for f in 0 1; do
    echo "== FUSED_CHAINS=$f" | tee -a $O/st.txt
    GS_FUSED_CHAINS=$f timeout 300 python tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -E "PGE|outer_step|sum of|synthetic" | tee -a $O/st.txt
  done
done
