#!/bin/bash
# round 5: balanced slabs of the real-side edge product (knob EDGE_PLAN): tests, stage times A/B, bench -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-edgeplan}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "balanced_edge or every_knob or per_edge_sign" > $O/tests.log 2>&1; tail -5 $O/tests.log
for v in 0 1 0 1; do
  GS_EDGE_PLAN=$v timeout 300 python tools/stage_times.py 2>/dev/null | grep -E "edge_plan|real-side|sum of|outer_step" >> $O/stage_times_edge_plan_ab.txt
done
cat $O/stage_times_edge_plan_ab.txt
timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline > $O/bench_arxiv.json 2> $O/bench_arxiv.err; python3 -c "
import json; d=json.load(open('$O/bench_arxiv.json')); print(d['value'], d['ms_per_step'], d['config']['iters_per_sec_with_forward_reuse'], d['roofline']['frac'])"
