#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/c
timeout 300 python tools/ab_panel.py > gpurun_out/c/ab_panel.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_operand_modes.py -q -m gpu -s > gpurun_out/c/tests_fullsize.log 2>&1
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/c/bench.json 2> gpurun_out/c/bench.err
cat gpurun_out/c/ab_panel.txt; tail -8 gpurun_out/c/tests_fullsize.log; cat gpurun_out/c/bench.json
