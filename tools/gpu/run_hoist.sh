#!/bin/bash
# round 5: the next iteration's training forward hoisted beside the inner steps (gs_gcond_set_hoist_forward): tests, stage times A/B, bench -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-hoist}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "hoisted or outer_iterations or fork_order or sibling_methods_follow or dropout" > $O/tests.log 2>&1; tail -5 $O/tests.log
for v in 0 1 0 1; do
  echo "GS_HOIST_FWD=$v" >> $O/stage_times_hoist_ab.txt
  GS_HOIST_FWD=$v timeout 300 python tools/stage_times.py 2>/dev/null | grep -E "sum of|outer_step|reuse" >> $O/stage_times_hoist_ab.txt
done
cat $O/stage_times_hoist_ab.txt
timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline > $O/bench_arxiv.json 2> $O/bench_arxiv.err; python3 -c "
import json; d=json.load(open('$O/bench_arxiv.json')); print(d['value'], d['ms_per_step'], d['config']['iters_per_sec_with_forward_reuse'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline'].get('training_pass',{}).get('avg_launch_us'))"
tail -3 $O/bench_arxiv.err
