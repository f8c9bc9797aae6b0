#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/reddit_tall
mkdir -p $O
for t in 0 1; do
  echo "== GS_TALL=$t" | tee -a $O/st.txt
  GS_TALL=$t timeout 300 python tools/stage_times.py reddit 2>/dev/null | grep -E "PGE|outer_step|sum of" | tee -a $O/st.txt
done
timeout 300 python tools/ab_tall.py 20 2>/dev/null | grep -E "reddit" | tee -a $O/st.txt
