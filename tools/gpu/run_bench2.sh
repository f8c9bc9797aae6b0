#!/bin/bash
# the bench line as the driver runs it (fresh process, default arguments first), twice each
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-bench2}
mkdir -p $O
timeout 600 python bench.py > $O/bench_arxiv.json 2> $O/bench_arxiv.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_arxiv_driver_args.json 2>/dev/null
timeout 600 python bench.py > $O/bench_arxiv_second.json 2>/dev/null
for f in bench_arxiv bench_arxiv_driver_args bench_arxiv_second; do python3 - $O/$f.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); print(sys.argv[1], d["value"], d["ms_per_step"], d.get("sustained", {}).get("value"), d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline_hbm"]["traffic"])
PY
done
