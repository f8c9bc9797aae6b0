#!/bin/bash
# the whole GPU suite on the current build -> gpurun_out/$1/tests_gpu_all.log
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-suite}
mkdir -p $O
timeout ${2:-2400} python -m pytest tests -x -q -m gpu > $O/tests_gpu_all.log 2>&1
tail -8 $O/tests_gpu_all.log
