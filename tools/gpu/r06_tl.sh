#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06/tl
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --sustain-s 0 --no-modes --eval-epochs 0 --no-prerun > $O/bench_under_rocprof.json 2>$O/err.txt
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
f=$(ls $O/kt/*/*kernel_trace.csv | head -1)
python3 tools/timeline_csv.py $f 40 > $O/timeline_a.txt 2>&1
rm -rf $O/kt
tail -c 400 $O/bench_under_rocprof.json
grep -n "inner_persist" -B8 -A12 $O/timeline_a.txt | cut -c1-150
tail -2 $O/timeline_a.txt
