#!/bin/bash
# round 6: the de-duplication as a chain of chip-wide launches: correctness (the numpy check of test_edge_product_over_distinct_sources,
# the deep path), kernel times, and the A/B of the entry-form edge product with it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "distinct_sources or t3 or t4" 2>&1 | tail -n 12 > $O/tests_dedup_chain.log
tail -n 6 $O/tests_dedup_chain.log
: > $O/edge_src_ab2.txt
for form in 0 1; do for src in 0 1; do
  echo "== GS_DEDUP_FORM=$form GS_EDGE_SRC=$src" >> $O/edge_src_ab2.txt
  GS_DEDUP_FORM=$form GS_EDGE_SRC=$src timeout 300 python3 tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -v "^/opt" >> $O/edge_src_ab2.txt
done; done
GS_EDGE_SRC=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dd1 -- python3 tools/stage_times.py ogbn-arxiv > /dev/null 2>&1
f=$(find /tmp/dd1 -name "*kernel_stats.csv" | head -1)
echo "# rocprofv3 --stats of the GS_EDGE_SRC=1 run (chain form):" >> $O/edge_src_ab2.txt
python3 - "$f" >> $O/edge_src_ab2.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("gs_ddm", "gs_dedup", "gs_edge_gemm", "gs_sample_hop")):
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:8.1f}  max {float(r['MaxNs']) / 1e3:8.1f}")
PY
grep -E "^==|real-side|outer_step|gs_ddm|gs_dedup|gs_edge|gs_sample" $O/edge_src_ab2.txt
