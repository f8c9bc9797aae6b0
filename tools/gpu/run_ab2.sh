#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ab2
python3 - <<'PY' > gpurun_out/ab2/ab_folded.txt 2>&1
import time, torch, sys, os
sys.path.insert(0, os.getcwd())
import bench
from graphslim_amd import _lib
L = _lib.lib()
p = bench.build_problem("ogbn-arxiv", seed=1)
eng = bench.make_engine(p, "cuda:0"); bench.init_state(p, eng); bench.new_model(p, eng)
def t(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for rnd in range(3):
    for on in (0, 1):
        L.gs_gcond_set_folded_reductions(on)
        print(f"folded={on}: PGE forward {t(lambda: eng.stage_pge_forward(False)):.1f} us, backward(feature) {t(lambda: eng.stage_pge_backward(False, True)):.1f} us, outer_step {t(lambda: eng.outer_step(10)):.1f} us")
L.gs_gcond_set_folded_reductions(1)
PY
cat gpurun_out/ab2/ab_folded.txt
