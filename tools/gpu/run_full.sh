#!/bin/bash
# whole GPU suite + the bench line (no CPU baseline) + the generated-A1 A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/full
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/full/tests.log 2>&1
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --sustain-s 0 > gpurun_out/full/bench.json 2>gpurun_out/full/bench.err
timeout 300 python tools/stage_times.py ogbn-arxiv f32 > gpurun_out/full/stage_times_f32.txt 2>&1
tail -5 gpurun_out/full/tests.log; tail -13 gpurun_out/full/stage_times_f32.txt; python3 -c "
import json;d=json.load(open('gpurun_out/full/bench.json'));print(d['value'],d['ms_per_step'],d['roofline'],{k:v['value'] for k,v in d['operand_modes'].items()})"
