#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 counter_collection.csv / kernel_trace.csv files: python tools/pmc_summary.py DIR [DIR...]"""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            if "gs_" in k or "copy" in k or "rowseg" in k:
                print(f"{k:72s} {c:28s} n={len(v):3d} avg={sum(v) / len(v):16.1f}")
    for f in sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in sorted(agg.items()):
            if "gs_" in k or "copy" in k or "rowseg" in k:
                print(f"{k:72s} {'duration_us':28s} n={len(v):3d} avg={sum(v) / len(v):16.1f} min={min(v):.1f}")
