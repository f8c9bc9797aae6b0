#!/bin/bash
# round 5: slabs aimed at by the balanced edge product (GS_EDGE_PLAN: 0 off, N slabs) x class -> XCD affinity of the ELL gathers -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-edgesweep}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_kernels.py -x -q -m gpu -k "balanced_edge or every_knob or ell or real_side or sampler" > $O/tests.log 2>&1; tail -3 $O/tests.log
for x in 1 0; do
for v in 0 384 416 448 480 512; do
  echo "== GS_ELL_XCD=$x GS_EDGE_PLAN=$v" >> $O/stage_times_edge_sweep.txt
  GS_ELL_XCD=$x GS_EDGE_PLAN=$v timeout 300 python tools/stage_times.py 2>/dev/null | grep -E "real-side|sum of|outer_step" >> $O/stage_times_edge_sweep.txt
done
done
cat $O/stage_times_edge_sweep.txt
