#!/bin/bash
# usage (GPU box): tools/gpu/pmc_dispatches.sh COUNTER "<kernel name fragment>" <script> [args...]  -> counter value of every matching dispatch, in order
c="$1"; pat="$2"; shift 2
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out=$(mktemp -d /tmp/pmc.XXXXXX)
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc "$c" --output-format csv -d "$out" -- python3 "$root/$1" "${@:2}" ) > "$out/log.txt" 2>&1
grep -E "^triage" "$out/log.txt" | tr '\n' ' '; echo
f=$(find "$out" -name "*counter_collection.csv" | head -1)
python3 - "$f" "$pat" "$c" <<'PY'
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(sys.argv[2], r["Kernel_Name"]) and r["Counter_Name"] == sys.argv[3]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
print(" ".join(f'{float(r["Counter_Value"]) / 1024:.1f}' for r in rows), "(MiB as reported, per dispatch)")
PY
