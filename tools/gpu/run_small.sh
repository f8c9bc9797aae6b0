#!/bin/bash
# A/B of the small-product routing limits (gs_gemm_set_small) at the bench shape.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/small
python3 - <<'PY' > gpurun_out/small/ab_small.txt 2>&1
import time, torch, sys, os
sys.path.insert(0, os.getcwd())
import bench
from graphslim_amd import _lib
L = _lib.lib()
p = bench.build_problem("ogbn-arxiv", seed=1)
eng = bench.make_engine(p, "cuda:0"); bench.init_state(p, eng); bench.new_model(p, eng)
def ms(ep, reps=60):
    for _ in range(5): eng.outer_step(ep)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): eng.outer_step(ep)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for rnd in range(3):
    for tiles, gf in ((8192, 1.5), (8192, 1.0), (8192, 0.6), (4096, 1.5), (2048, 1.5), (16384, 3.0)):
        L.gs_gemm_set_small(tiles, gf)
        print(f"small route <= {tiles} tiles and <= {gf} GFLOP: feature epoch {ms(10):.3f} ms   PGE epoch {ms(0):.3f} ms", flush=True)
L.gs_gemm_set_small(8192, 1.5)
PY
grep -v "^/opt" gpurun_out/small/ab_small.txt
