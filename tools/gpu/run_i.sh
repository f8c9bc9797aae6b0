#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/i
timeout 300 python tools/ab_bgen.py > gpurun_out/i/ab_bgen.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_gemm_big.py -q -m gpu -k edge_product > gpurun_out/i/tests.log 2>&1
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/i/bench.json 2>/dev/null
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --operand-mode bf16x3 > gpurun_out/i/bench_bf16x3.json 2>/dev/null
cat gpurun_out/i/ab_bgen.txt; tail -3 gpurun_out/i/tests.log; cut -c1-330 gpurun_out/i/bench.json; echo; cut -c1-330 gpurun_out/i/bench_bf16x3.json
