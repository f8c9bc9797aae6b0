#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/h
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/h/kt -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/h/bench_prof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/h/kt_bf -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --operand-mode bf16x3 > gpurun_out/h/bench_prof_bf.json 2>/dev/null
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --operand-mode bf16x3 > gpurun_out/h/bench_bf16x3.json 2>/dev/null
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --operand-mode bf16 > gpurun_out/h/bench_bf16.json 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_reduce.py -q -m gpu -k "sibling_runs or whole_reference" > gpurun_out/h/tests.log 2>&1
f=$(ls gpurun_out/h/kt/*/*kernel_trace.csv | head -1); python3 tools/timeline_csv.py $f 3 > gpurun_out/h/timeline_f32.txt
f=$(ls gpurun_out/h/kt_bf/*/*kernel_trace.csv | head -1); python3 tools/timeline_csv.py $f 3 > gpurun_out/h/timeline_bf16x3.txt
rm -rf gpurun_out/h/kt/*/*agent_info.csv
tail -3 gpurun_out/h/tests.log; cat gpurun_out/h/bench_bf16x3.json; cat gpurun_out/h/bench_bf16.json; tail -1 gpurun_out/h/timeline_f32.txt; tail -1 gpurun_out/h/timeline_bf16x3.txt
