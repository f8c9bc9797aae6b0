#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --sustain-s 0 --eval-epochs 0 --no-modes --no-prerun > $O/bench_under_rocprof2.json 2>/dev/null
f=$(ls /tmp/kt/*/*kernel_trace.csv | head -1); python3 tools/timeline_csv.py $f 70 > $O/timeline_noreuse_plain14.txt 2>/dev/null
cp $(ls /tmp/kt/*/*kernel_stats.csv | head -1) $O/bench_arxiv_kernel_stats_plain14.csv
sed -n 45,85p $O/timeline_noreuse_plain14.txt
