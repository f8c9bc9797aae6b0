#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/g
timeout 300 python tools/ab_panel.py > gpurun_out/g/ab_panel.txt 2>&1
timeout 300 python tools/trace_panel.py > gpurun_out/g/trace_panel.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_panel.py tests/test_gpu_gemm_big.py -x -q -m gpu > gpurun_out/g/tests.log 2>&1
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/g/bench.json 2> gpurun_out/g/bench.err
cat gpurun_out/g/ab_panel.txt; grep -E "^---|K-loop" gpurun_out/g/trace_panel.txt; tail -2 gpurun_out/g/tests.log; cat gpurun_out/g/bench.json
