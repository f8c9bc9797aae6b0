#!/bin/bash
# round 5: with_bn under mse / cos (golden G23), the edge tests, the Amdahl stage table -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-bn}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "with_bn or balanced_edge or every_knob" > $O/tests_bn.log 2>&1; tail -25 $O/tests_bn.log
bash tools/gpu/run_amdahl.sh $1 > /dev/null 2>&1
grep -E "^==|virtual rank|us$|Error|error" $O/amdahl_stage_times.txt | head -90
