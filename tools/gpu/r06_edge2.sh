#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06/tl2
mkdir -p $O
GS_EDGE_SRC=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/stage_times.py > $O/out.txt 2>$O/err.txt
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/kt
head -40 $O/kernel_stats.csv | cut -c1-170
