#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/n
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/n/tests_all.log 2>&1
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --sustain-s 0 > gpurun_out/n/bench.json 2>/dev/null
tail -4 gpurun_out/n/tests_all.log; python3 -c "
import json;d=json.load(open('gpurun_out/n/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['isolated_launch_us'],{k:v['value'] for k,v in d['operand_modes'].items()})"
