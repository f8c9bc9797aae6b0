#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/j
timeout 300 python tools/ab_bgen.py > gpurun_out/j/ab_bgen.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/j/kt -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/j/bench_prof.json 2>/dev/null
f=$(ls gpurun_out/j/kt/*/*kernel_trace.csv | head -1); python3 tools/timeline_csv.py $f 18 > gpurun_out/j/timeline_f32_noreuse.txt; python3 tools/timeline_csv.py $f 3 > gpurun_out/j/timeline_f32_reuse.txt
rm -rf gpurun_out/j/kt
cat gpurun_out/j/ab_bgen.txt; tail -1 gpurun_out/j/timeline_f32_noreuse.txt
