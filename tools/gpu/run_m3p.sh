#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m3p; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/stage_times.py ogbn-arxiv bf16x6 > /dev/null 2>&1
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv; rm -rf $O/kt
grep "panel\|edge" $O/kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c1-150
