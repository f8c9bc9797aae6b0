#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/e
timeout 300 python tools/ab_panel.py > gpurun_out/e/ab_panel.txt 2>&1
cat gpurun_out/e/ab_panel.txt
