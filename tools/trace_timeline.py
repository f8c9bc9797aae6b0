#!/usr/bin/env python3
"""Timeline of ONE outer iteration from a rocprofv3 rocpd database: per queue (stream), kernels in start order with start
offset, duration and the gap to the previous kernel on the same queue.  usage: trace_timeline.py results.db STEPS WARMUP [STEP]"""
import sqlite3
import sys

path, steps, warm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
pick = int(sys.argv[4]) if len(sys.argv) > 4 else steps // 2
db = sqlite3.connect(path)
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"select s.kernel_name, d.start, d.end, d.{qcol} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
# a step starts at each gs_sample_hop1_kernel launch... use the Adam kernel on X~/PGE as a delimiter instead: split by count
per = len(rows) / float(steps + warm)
lo, hi = int(per * (warm + pick)), int(per * (warm + pick + 1))
seg = rows[lo:hi]
t0 = seg[0][1]
last = {}
for name, s, e, q in seg:
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    short = name.replace("_Z", "").split("Pf")[0][:70]
    print(f"q{q} +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {short}")
