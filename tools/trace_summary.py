#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace): launches / span / kernel-time sum / union-busy per step and the
top kernels.  usage: trace_summary.py results.db STEPS WARMUP [TOP]"""
import sqlite3
import sys
from collections import defaultdict

path, steps, warm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 25
db = sqlite3.connect(path)
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
per = len(rows) / float(steps + warm)
rows = rows[int(per * warm):]
span = (rows[-1][2] - rows[0][1]) / 1e6
busy = sum(e - s for _, s, e in rows) / 1e6
ivs = sorted((s, e) for _, s, e in rows)
u, (cs, ce) = 0, ivs[0]
for s, e in ivs[1:]:
    if s > ce:
        u += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
u += ce - cs
print(f"launches/step {per:.1f}  span/step {span / steps:.3f} ms  kernel-sum/step {busy / steps:.3f} ms  union-busy/step {u / 1e6 / steps:.3f} ms")
agg = defaultdict(lambda: [0, 0])
for k, s, e in rows:
    agg[k[:100]][0] += 1
    agg[k[:100]][1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
    print(f"  {k:100s} {c / steps:6.1f}/step {t / steps / 1e3:8.1f} us/step  avg {t / c / 1e3:7.1f} us")
