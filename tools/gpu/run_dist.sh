#!/bin/bash
# matching distance shared by several workgroups per class: parity tests that touch the distance, stage times, solo kernel time
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-dist1}
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_kernels.py -x -q -m gpu -k "chained or fused or golden or metric or dist or match or sharded or train_class or sibling" > $O/tests.log 2>&1; tail -3 $O/tests.log
for r in 1 2; do
  timeout 300 python tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -E "synthetic match|outer_step" | tee -a $O/st.txt
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/trace_solo.py > /dev/null 2>&1
f=$(ls $O/kt/*/*kernel_trace.csv | head -1); python3 tools/timeline_csv.py $f 8 > $O/timeline_solo_no_overlap.txt 2>&1
rm -rf $O/kt
grep -n "gs_dist\|iteration span" $O/timeline_solo_no_overlap.txt
