#!/usr/bin/env python3
"""profiles/r02/pmc_panel_summary.json from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/probe_panel.py:
per kernel the HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE reports exactly half of the
bytes read — calibrated with tools/micro/copy_calib on 4-byte, 16-byte and row-segment reads, profiles/r02/copy_calib_a.txt,
pmc_a_calib_*.csv; WRITE_SIZE is exact) and the average duration of the traced launches.
usage: make_pmc_summary.py FETCH_DIR WRITE_DIR OUT.json"""
import collections
import csv
import glob
import json
import sys


def avg(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def dur(d):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: sum(v) / len(v) for k, v in agg.items()}


fd, wd, out = sys.argv[1:4]
fetch, write, du = avg(fd, "FETCH_SIZE"), avg(wd, "WRITE_SIZE"), dur(fd)
res = {}
for k in fetch:
    if "gs_" not in k:
        continue
    name = k.replace("void ", "").split("(")[0]
    res[name] = {"fetch_size_kib_raw": round(fetch[k], 1), "fetch_kib_corrected_x2": round(2 * fetch[k], 1),
                 "write_size_kib": round(write.get(k, 0.0), 1),
                 "hbm_bytes_per_launch": int((2 * fetch[k] + write.get(k, 0.0)) * 1024),
                 "avg_duration_us_under_profiler": round(du.get(k, 0.0), 1)}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
