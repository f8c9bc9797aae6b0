#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-skip1}
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_engine.py -x -q -m gpu -k "arxiv or flickr or sharded or sibling or fork_order or fused" > $O/tests.log 2>&1; tail -5 $O/tests.log
for f in 0 1; do
  echo "== FUSED_CHAINS=$f" | tee -a $O/st.txt
  GS_FUSED_CHAINS=$f timeout 300 python tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -E "outer_step" | tee -a $O/st.txt
done
