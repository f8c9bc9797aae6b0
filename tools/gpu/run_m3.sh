#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/m3
timeout 1500 python -m pytest tests/test_gpu_panel.py tests/test_gpu_operand_modes.py -q -m gpu -s -k "three_way or bf16x6 or panel_against or epilogues" > gpurun_out/m3/tests.log 2>&1
timeout 300 python tools/stage_times.py ogbn-arxiv bf16x6 > gpurun_out/m3/st_bf16x6.txt 2>&1
timeout 300 python tools/stage_times.py ogbn-arxiv f32 > gpurun_out/m3/st_f32.txt 2>&1
grep -n "passed\|failed" gpurun_out/m3/tests.log | tail -2; grep "3-way\|FAILED\|Error" gpurun_out/m3/tests.log | head -20; tail -13 gpurun_out/m3/st_bf16x6.txt; tail -2 gpurun_out/m3/st_f32.txt
