#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-tlpge}
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt2 -- python3 tools/trace_solo.py ogbn-arxiv 12 0 > /dev/null 2>&1
f=$(ls $O/kt2/*/*kernel_trace.csv | head -1); python3 tools/timeline_csv.py $f 4 > $O/timeline_solo_pge_epoch.txt 2>&1
rm -rf $O/kt2
head -40 $O/timeline_solo_pge_epoch.txt | cut -c1-130; tail -1 $O/timeline_solo_pge_epoch.txt
