#!/bin/bash
# chained per-class products of the synthetic match (csrc/gs_match.hip): parity against the batched products, stage times both ways
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-match1}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "chained_class" > $O/test_match.log 2>&1; tail -15 $O/test_match.log
for m in 0 1; do
  GS_MATCH_CLASS=$m timeout 300 python tools/stage_times.py ogbn-arxiv > $O/stage_times_match$m.txt 2>&1
  grep -E "synthetic match|outer_step|sum of" $O/stage_times_match$m.txt
done
