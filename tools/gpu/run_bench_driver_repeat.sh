#!/bin/bash
# round 5: the driver's own command six times in fresh processes -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-benchdrv}
mkdir -p $O
for i in 1 2 3 4 5 6; do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_$i.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('$O/bench_$i.json')); print('run $i:', d['value'], 'it/s', d['ms_per_step'], 'ms/step; sustained', d['sustained']['value'], '; forward reuse', d['config']['iters_per_sec_with_forward_reuse'], '; inference pass', d['roofline']['avg_launch_us'], 'us, training pass', d['roofline']['training_pass']['avg_launch_us'], 'us')" | tee -a $O/bench_driver_repeat.txt
done
