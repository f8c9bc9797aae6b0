#!/usr/bin/env python3
"""HBM bytes of the hop-2 gather's OWN dispatches: gs_ell_spmm_kernel<4> runs twice per iteration — hop 2
(frontier rows <- H1full rows, the large grid) and hop 1 (seeds <- frontier) — and the per-kernel PMC average mixes them.  Reads the
FETCH_SIZE / WRITE_SIZE passes of the bench (rocprofv3 counter_collection.csv), keeps the dispatches of that kernel with the LARGEST
grid, and writes {hbm_bytes_per_launch = 2 x FETCH + WRITE (gfx950 calibration, tools/make_pmc_summary.py), ...}.
usage: pmc_hop2.py FETCH_DIR WRITE_DIR OUT.json"""
import csv
import glob
import json
import sys


def per_dispatch(d, counter):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "gs_ell_spmm_kernel<4" in r["Kernel_Name"]:
                rows.append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return rows


fd, wd, out = sys.argv[1:4]
res = {}
for name, d, c in (("fetch", fd, "FETCH_SIZE"), ("write", wd, "WRITE_SIZE")):
    rows = per_dispatch(d, c)
    g = max(r[0] for r in rows)
    big = [v for gs, v in rows if gs == g]
    small = [v for gs, v in rows if gs != g]
    res[name + "_kib_hop2"] = round(sum(big) / len(big), 1)
    res[name + "_kib_hop1"] = round(sum(small) / max(len(small), 1), 1)
    res["dispatches_hop2"] = len(big)
    res["grid_hop2"] = g
res["hbm_bytes_per_launch"] = int((2 * res["fetch_kib_hop2"] + res["write_kib_hop2"]) * 1024)
res["hbm_bytes_per_launch_hop1"] = int((2 * res["fetch_kib_hop1"] + res["write_kib_hop1"]) * 1024)
res["note"] = "FETCH_SIZE doubled (gfx950 reports half the bytes read, profiles/r02/copy_calib_a.txt); WRITE_SIZE exact"
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
