#!/bin/bash
# round 5: the ipc peer-read transport between processes sharing the GPU, the edge tests, and the Amdahl stage table -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-ipc}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multiprocess.py -x -q -m gpu -k "ipc_peer_read or two_processes" > $O/tests_ipc.log 2>&1; tail -15 $O/tests_ipc.log
timeout 300 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "balanced_edge or every_knob" > $O/tests_edge.log 2>&1; tail -2 $O/tests_edge.log
bash tools/gpu/run_amdahl.sh $1 > /dev/null 2>&1
grep -E "^==|virtual rank|us$" $O/amdahl_stage_times.txt | head -80
