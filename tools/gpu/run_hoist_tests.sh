#!/bin/bash
# round 5: the hoisted-forward tests (engine + full size against the oracle) -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-hoistt}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -q -m gpu -k "hoisted or outer_iterations" > $O/tests_engine.log 2>&1; tail -4 $O/tests_engine.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -k "fused_outer_step" > $O/tests_fullsize_fused.log 2>&1; tail -12 $O/tests_fullsize_fused.log
