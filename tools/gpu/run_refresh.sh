#!/bin/bash
# refresh after a default changed: hoist tests, full-size fused tests, the two bench lines -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-refresh}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -q -m gpu -k "hoisted or outer_iterations" > $O/tests_engine.log 2>&1; tail -3 $O/tests_engine.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "fused_outer_step or three_hops" > $O/tests_fullsize.log 2>&1; tail -3 $O/tests_fullsize.log
timeout 600 python bench.py > $O/bench_arxiv.json 2> $O/bench_arxiv.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_arxiv_driver_args.json 2>/dev/null
python3 -c "
import json
for f in ('bench_arxiv.json','bench_arxiv_driver_args.json'):
    d=json.load(open('$O/'+f)); print(f, d['value'], d['ms_per_step'], d['config']['iters_per_sec_with_forward_reuse'], d['roofline']['avg_launch_us'], d['roofline']['training_pass']['avg_launch_us'], d['roofline_hbm']['frac'])"
