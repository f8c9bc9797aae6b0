#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
: > $O/stage_times_modes.txt
for m in bf16x6 bf16x3 bf16; do
  rm -rf /tmp/km
  echo "== operand mode $m" >> $O/stage_times_modes.txt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/km -- python3 tools/stage_times.py ogbn-arxiv $m 2>/dev/null | grep -E "real-side|PGE forward|PGE backward \(feature|synthetic match|outer_step" >> $O/stage_times_modes.txt
  f=$(find /tmp/km -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $O/stage_times_modes.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("gs_edge_gemm", "gs_tall_kernel", "gs_ell_spmm", "gs_csc_spmm")):
        print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:8.1f}")
PY
done
cat $O/stage_times_modes.txt
