#!/bin/bash
# kernel-trace timeline of one feature-epoch and one PGE-epoch outer iteration + per-kernel stats of the bench command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/tl
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --sustain-s 0 > $O/bench_under_rocprof.json 2>$O/err.txt
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
f=$(ls $O/kt/*/*kernel_trace.csv | head -1)
python3 tools/timeline_csv.py $f 190 > $O/timeline_a.txt 2>&1
python3 tools/timeline_csv.py $f 230 > $O/timeline_b.txt 2>&1
rm -rf $O/kt
head -30 $O/kernel_stats.csv | cut -c1-160
tail -3 $O/timeline_a.txt
