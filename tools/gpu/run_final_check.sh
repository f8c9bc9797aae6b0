#!/bin/bash
# last check of the round's head: smoke, a few engine tests, the driver's bench command -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-fcheck}
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_engine.py -q -m gpu -k "hoisted or outer_iterations or sharded_engines_reproduce or dropout or golden" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_multiprocess.py -q -m gpu > $O/tests_mp.log 2>&1; tail -3 $O/tests_mp.log
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_hbm']['frac'], d['cpu_baseline']['value'])"
