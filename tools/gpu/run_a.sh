#!/bin/bash
# round-2 GPU pass A: new full-size parity tests, bench line, FETCH/WRITE calibration, PMC of the tiled GEMM with the XCD-aware grid
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/a
nproc > gpurun_out/a/host.txt; rocm-smi --showproductname >> gpurun_out/a/host.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_gemm_big.py -x -q -m gpu > gpurun_out/a/tests_gemm_big.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s > gpurun_out/a/tests_fullsize.log 2>&1
timeout 600 python bench.py --steps 50 --warmup 5 > gpurun_out/a/bench.json 2> gpurun_out/a/bench.err
./tools/micro/copy_calib > gpurun_out/a/copy_calib.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/a/calib_$c -- ./tools/micro/copy_calib > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/a/gemm_$c -- python3 tools/probe_gemm.py > /dev/null 2>&1
done
tail -5 gpurun_out/a/tests_gemm_big.log; tail -30 gpurun_out/a/tests_fullsize.log; cat gpurun_out/a/bench.json; cat gpurun_out/a/copy_calib.txt
