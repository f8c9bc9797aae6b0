#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/k
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/k/tests.log 2>&1
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/k/bench.json 2>/dev/null
tail -3 gpurun_out/k/tests.log; cut -c1-330 gpurun_out/k/bench.json
