#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/span
mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline --sustain-s 0 > $O/bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --no-cpu-baseline --sustain-s 0 > $O/bench_under_rocprof.json 2>/dev/null
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv; rm -rf $O/kt
grep fwdgen $O/kernel_stats.csv | cut -c1-160
python3 -c "
import json
for f in ('bench.json','bench_under_rocprof.json'):
    d=json.load(open('$O/'+f)); r=d['roofline']; print(f, d['value'], d['ms_per_step'], r['avg_launch_us'], r['event_bracket_us'], r['isolated_launch_us'], r['isolated_span_us'], r['frac'], r['traffic'])"
