#!/bin/bash
# round 5: the default bench line three times in fresh processes (run-to-run spread on one box) -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-benchrep}
mkdir -p $O
for i in 1 2 3; do
  timeout 600 python bench.py --no-cpu-baseline --eval-epochs 0 --no-modes > $O/bench_$i.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('$O/bench_$i.json')); print('run $i:', d['value'], 'it/s', d['ms_per_step'], 'ms/step; forward reuse', d['config']['iters_per_sec_with_forward_reuse'], '; inference pass', d['roofline']['avg_launch_us'], 'us, training pass', d['roofline']['training_pass']['avg_launch_us'], 'us')" | tee -a $O/bench_repeat.txt
done
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -4 >> $O/bench_repeat.txt
