#!/bin/bash
# round 5: padding rows of the real side's intermediate gathers left unwritten: tests, stage times A/B -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-padzero}
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_kernels.py -x -q -m gpu -k "golden or outer_iterations or sharded_engines_reproduce or edge or sign_bit or sampler or dropout or three_hops or fused_chains" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests_fullsize.log 2>&1; tail -3 $O/tests_fullsize.log
for v in 1 0 1 0; do
  echo "GS_PAD_ZERO=$v" >> $O/stage_times_pad_zero.txt
  GS_PAD_ZERO=$v timeout 300 python tools/stage_times.py 2>/dev/null | grep -E "real-side|outer_step" >> $O/stage_times_pad_zero.txt
done
cat $O/stage_times_pad_zero.txt
