#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-fcheck2}
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_multiprocess.py -q -m gpu > $O/tests_mp.log 2>&1; tail -3 $O/tests_mp.log
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --eval-epochs 0 --no-cpu-baseline --no-modes > $O/bench_noeval.json 2>$O/bench_noeval.err; python3 -c "
import json; d=json.load(open('$O/bench_noeval.json')); print(d['value'], d['ms_per_step'], d['sustained']['value'])"
