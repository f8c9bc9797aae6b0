#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-probe}
mkdir -p $O
python tools/probe_cost.py 2>/dev/null | tee $O/probe_cost_stride4.txt
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_multiprocess.py -q -m gpu -k "hoisted or bench or full_size_arxiv" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python bench.py > $O/bench_arxiv.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_arxiv_driver_args.json 2>/dev/null
python3 -c "
import json
for f in ('bench_arxiv.json','bench_arxiv_driver_args.json'):
    d=json.load(open('$O/'+f)); r=d['roofline']; print(f, d['value'], d['ms_per_step'], r['avg_launch_us'], r['launches_timed'], r['event_brackets_timed'], r['event_bracket_us'], r['training_pass']['avg_launch_us'], d['roofline_hbm']['avg_launch_us'], d['roofline_hbm']['launches_timed'])"
