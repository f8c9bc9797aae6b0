#!/bin/bash
# chained per-class products of the synthetic match (csrc/gs_match.hip): parity against the batched products, stage times both ways,
# solo timeline of one iteration
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-match1}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "chained or fused_chains" > $O/test_match.log 2>&1; tail -15 $O/test_match.log
for m in 0 1; do
  GS_CHAINED=$m timeout 300 python tools/stage_times.py ogbn-arxiv > $O/stage_times_match$m.txt 2>&1
  grep -E "synthetic match|outer_step|sum of" $O/stage_times_match$m.txt
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/trace_solo.py > /dev/null 2>$O/solo_err.txt
f=$(ls $O/kt/*/*kernel_trace.csv | head -1); python3 tools/timeline_csv.py $f 8 > $O/timeline_solo_no_overlap.txt 2>&1
rm -rf $O/kt
grep -n "gs_match\|gs_dist\|iteration span" $O/timeline_solo_no_overlap.txt
grep -n "gs_chain_h1t\|gs_chain_rdh" $O/timeline_solo_no_overlap.txt | head -4
grep -E "inner SGC" $O/stage_times_match0.txt $O/stage_times_match1.txt
