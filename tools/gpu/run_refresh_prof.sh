#!/bin/bash
# refresh of the bench evidence after the last default change: the two bench lines, kernel-trace stats of the same command, timelines, stage times -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-refprof}
mkdir -p $O
timeout 600 python bench.py > $O/bench_arxiv.json 2> $O/bench_arxiv.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_arxiv_driver_args.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --sustain-s 0 --eval-epochs 0 --no-modes --no-prerun > $O/bench_under_rocprof.json 2>/dev/null
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/bench_arxiv_kernel_stats.csv
f=$(ls $O/kt/*/*kernel_trace.csv | head -1); python3 tools/timeline_csv.py $f 70 > $O/timeline_noreuse.txt 2>/dev/null; python3 tools/timeline_csv.py $f 12 > $O/timeline_reuse.txt 2>/dev/null
rm -rf $O/kt
timeout 300 python tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -v "^/opt" > $O/stage_times.txt
timeout 300 python tools/stage_times.py ogbn-arxiv f32 2>/dev/null | grep -v "^/opt" >> $O/stage_times.txt
python3 -c "
import json
for f in ('bench_arxiv.json','bench_arxiv_driver_args.json','bench_under_rocprof.json'):
    d=json.load(open('$O/'+f)); print(f, d['value'], d['ms_per_step'], d['config']['iters_per_sec_with_forward_reuse'], d['roofline']['avg_launch_us'], d['roofline']['training_pass']['avg_launch_us'])"
grep "gs_tall_kernel<1" $O/bench_arxiv_kernel_stats.csv | cut -c1-120
tail -1 $O/timeline_noreuse.txt
