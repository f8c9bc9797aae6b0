#!/bin/bash
# usage (on the GPU box): tools/gpu/prof_kernels.sh "<kernel name fragments, |-separated>" <script> [args...]
# rocprofv3 kernel trace + stats of `python3 <script> args`, printing calls / avg / min / max (ns) of the matching kernels
pat="$1"; shift
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out=$(mktemp -d /tmp/prof.XXXXXX)
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$root/$1" "${@:2}" ) > "$out/log.txt" 2>&1
tail -2 "$out/log.txt" | grep -v "^$" | grep -iv rocprof
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$pat" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print(f'{r["Name"][:72]:72s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  min {float(r["MinNs"]) / 1e3:8.1f}  max {float(r["MaxNs"]) / 1e3:8.1f}')
PY
