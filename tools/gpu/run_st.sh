#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/st
for cfg in 8192,1.5 4096,1.0 2048,0.6 1024,0.3 512,0.15 16384,3.0; do
  echo "== small route: $cfg" >> gpurun_out/st/sweep.txt
  GS_SMALL=$cfg timeout 300 python tools/stage_times.py ogbn-arxiv f32 2>/dev/null | grep -v "^/opt" >> gpurun_out/st/sweep.txt
done
cat gpurun_out/st/sweep.txt
