#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/m
timeout 300 python tools/ab_panel.py > gpurun_out/m/ab_panel.txt 2>&1
timeout 300 python tools/trace_panel.py > gpurun_out/m/trace_panel.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_panel.py tests/test_gpu_gemm_big.py tests/test_gpu_reduce.py -x -q -m gpu > gpurun_out/m/tests.log 2>&1
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --sustain-s 0 > gpurun_out/m/bench.json 2>/dev/null
grep -v loop-rate gpurun_out/m/ab_panel.txt; grep -E "^---|K-loop" gpurun_out/m/trace_panel.txt; tail -3 gpurun_out/m/tests.log; cut -c1-330 gpurun_out/m/bench.json
