#!/bin/bash
# round 5: skeleton times of the persistent 256-row products (tools/triage_forms.py), one rocprofv3 kernel-stats run per triage mask,
# and the per-step stamps of the plain product (tools/trace_tall.py) -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-triage_forms}
mkdir -p $O
: > $O/triage_forms.txt
for tri in 0 1 16 32 17 48 49; do
  rm -rf /tmp/tf_$tri
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tf_$tri -- python3 tools/triage_forms.py $tri > /dev/null 2>&1
  f=$(find /tmp/tf_$tri -name "*kernel_stats.csv" | head -1)
  echo "== triage mask $tri (1 no epilogue, 16 no MFMAs, 32 no operand preparation)" >> $O/triage_forms.txt
  python3 - "$f" >> $O/triage_forms.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gs_tall_kernel" in r["Name"]:
        print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:8.1f} us")
PY
done
cat $O/triage_forms.txt
timeout 200 python3 tools/trace_tall.py > $O/trace_tall_steps.txt 2>&1
grep "^wg" $O/trace_tall_steps.txt
