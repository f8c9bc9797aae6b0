#!/bin/bash
# refresh of the other workloads' bench lines on the final build -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-refother}
mkdir -p $O
timeout 600 python bench.py --workload reddit --no-cpu-baseline --eval-epochs 0 > $O/bench_reddit.json 2>/dev/null
timeout 600 python bench.py --workload cora --no-cpu-baseline --eval-epochs 0 > $O/bench_cora.json 2>/dev/null
timeout 600 python bench.py --operand-mode f32 --no-cpu-baseline --eval-epochs 0 > $O/bench_arxiv_f32.json 2>/dev/null
timeout 900 python bench.py --workload arxiv-k3 --steps 20 --warmup 3 --no-cpu-baseline --eval-epochs 0 --no-modes --sustain-s 0 > $O/bench_arxiv_three_hops.json 2> $O/bench_arxiv_three_hops.err
python3 -c "
import json
for f in ('bench_reddit.json','bench_cora.json','bench_arxiv_f32.json','bench_arxiv_three_hops.json'):
    d=json.load(open('$O/'+f)); print(f, d['value'], d['ms_per_step'], d['config'].get('iters_per_sec_with_forward_reuse'))"
