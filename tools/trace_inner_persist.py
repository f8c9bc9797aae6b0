#!/usr/bin/env python3
"""Where the persistent inner-step launch (csrc/gs_inner.hip) spends its time: s_memrealtime stamps (100 MHz) of thread 0 of every
workgroup around every grid barrier — work before the barrier, drain of the write-through stores, wait for the last arriver, acquire.
usage (MI355X): trace_inner_persist.py [grid]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from graphslim_amd import _lib  # noqa: E402

L = _lib.lib()
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 48
L.gs_inner_persist_set_grid(grid)
p = bench.build_problem(os.environ.get("GS_WORKLOAD", "ogbn-arxiv"), seed=1)
eng = bench.make_engine(p, "cuda:0")
bench.init_state(p, eng)
bench.new_model(p, eng)
eng.stage_pge_forward(True)
for _ in range(5):
    eng.inner_steps()
torch.cuda.synchronize()
buf = torch.zeros(256 * 128, dtype=torch.int64, device="cuda:0")
L.gs_inner_persist_trace(_lib.ptr(buf))
eng.inner_steps()
torch.cuda.synchronize()
L.gs_inner_persist_trace(None)
t = buf.cpu().numpy().reshape(256, 128)
G = int((t[:, 0] != 0).sum())
t = t[:G].astype(np.float64) / 100.0      # us
nbar = 4 * p["inner"]
t0 = t[:, 0].min()
names = ["phase 0"] + [f"step {s} phase {ph}" for s in range(p["inner"]) for ph in "ABCD"]
print(f"grid {G}, {nbar} barriers; all times in us (workgroup mean / max), launch start = first workgroup's first stamp")
prev = t[:, 0]
tot = {"work": 0.0, "drain": 0.0, "wait": 0.0, "acquire": 0.0}
for b in range(nbar):
    e, a, r, x = (t[:, 1 + 4 * b + k] for k in range(4))
    work, drain, wait, acq = e - prev, a - e, r - a, x - r
    print(f"{names[b]:18s} work {work.mean():6.2f} / {work.max():6.2f}   drain {drain.mean():5.2f}   wait {wait.mean():5.2f} (min {wait.min():5.2f})   "
          f"acquire {acq.mean():5.2f}   barrier released at {r.max() - t0:7.2f}")
    for k, v in zip(tot, (work.max(), drain.mean(), wait.min(), acq.mean())):
        tot[k] += v
    prev = x
end = t[:, 4 * nbar + 1]
print(f"{names[nbar]:18s} work {(end - prev).mean():6.2f} / {(end - prev).max():6.2f}   launch ends at {end.max() - t0:7.2f}")
print("critical-path estimate: " + ", ".join(f"{k} {v:.1f}" for k, v in tot.items()))
