#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/f
timeout 300 python tools/trace_panel.py > gpurun_out/f/trace_panel.txt 2>&1
cat gpurun_out/f/trace_panel.txt
