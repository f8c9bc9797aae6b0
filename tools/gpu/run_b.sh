#!/bin/bash
# round-2 GPU pass B: panel kernel parity, full-size parity again, operand modes, timing of the panel kernel per mode, bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/b
timeout 900 python -m pytest tests/test_gpu_panel.py tests/test_gpu_gemm_big.py -x -q -m gpu -s > gpurun_out/b/tests_panel.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_operand_modes.py -q -m gpu -s > gpurun_out/b/tests_fullsize.log 2>&1
timeout 300 python tools/ab_panel.py > gpurun_out/b/ab_panel.txt 2>&1
timeout 600 python bench.py --steps 50 --warmup 5 > gpurun_out/b/bench.json 2> gpurun_out/b/bench.err
tail -3 gpurun_out/b/tests_panel.log; grep -E "^panel|passed|failed" gpurun_out/b/tests_panel.log | tail -40; tail -15 gpurun_out/b/tests_fullsize.log; cat gpurun_out/b/ab_panel.txt; cat gpurun_out/b/bench.json
