#!/bin/bash
# final build: bench lines as the driver runs them + the whole GPU suite + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-final5}
mkdir -p $O
timeout 600 python bench.py > $O/bench_arxiv.json 2> $O/bench_arxiv.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_arxiv_driver_args.json 2>/dev/null
timeout 600 python bench.py > $O/bench_arxiv_second_run.json 2>/dev/null
for f in bench_arxiv bench_arxiv_driver_args bench_arxiv_second_run; do python3 - $O/$f.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); print(sys.argv[1], d["value"], d["ms_per_step"], d.get("sustained", {}).get("value"), d["config"].get("iters_per_sec_with_forward_reuse"), d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
PY
done
timeout 300 python tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -v "^/opt" > $O/stage_times.txt; tail -3 $O/stage_times.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests_gpu_all.log 2>&1
tail -2 $O/tests_gpu_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
