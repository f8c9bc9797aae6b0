#!/usr/bin/env python3
"""Timeline of ONE outer iteration from a rocprofv3 kernel_trace.csv: kernels in start order per queue with start offset,
duration, gap to the previous kernel of the same queue and grid size.  An iteration = from one launch of the matching stage's last
kernel (gs_norm_bwd_out_kernel; the loss bookkeeping that used to follow it rides in the V kernel since round 4) to the next — so a
printed "iteration" starts with the PGE backward.  usage: timeline_csv.py kernel_trace.csv [which_iteration_from_the_end=3]"""
import csv
import re
import sys

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = [r for r in csv.DictReader(open(path))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "gs_loss_accum_kernel" in r["Kernel_Name"]]
if len(marks) <= back:
    marks = [i for i, r in enumerate(rows) if "gs_norm_bwd_out_kernel" in r["Kernel_Name"]]
lo, hi = marks[-back - 1], marks[-back]
seg = rows[lo + 1:hi + 1]
t0 = int(seg[0]["Start_Timestamp"])
last, tot = {}, {}
for r in seg:
    q, s, e = r["Queue_Id"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:64]
    tot[q] = tot.get(q, 0) + (e - s)
    print(f"q{q} +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  grid {r['Grid_Size_X']:>8}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']:<5} {name}")
span = (max(int(r["End_Timestamp"]) for r in seg) - t0) / 1e3
print(f"iteration span {span:.1f} us, {len(seg)} launches; kernel time per queue: " + ", ".join(f"q{q}: {v / 1e3:.1f} us" for q, v in tot.items()))
