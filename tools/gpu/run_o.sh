#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/o
timeout 300 python tools/host_issue.py > gpurun_out/o/host_issue.txt 2>&1
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --sustain-s 0 > gpurun_out/o/bench.json 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_panel.py tests/test_gpu_engine.py -x -q -m gpu > gpurun_out/o/tests.log 2>&1
cat gpurun_out/o/host_issue.txt; tail -2 gpurun_out/o/tests.log; python3 -c "
import json;d=json.load(open('gpurun_out/o/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['isolated_launch_us'],{k:v['value'] for k,v in d['operand_modes'].items()})"
