#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ov
python3 - <<'PY' > gpurun_out/ov/ab_overlap.txt 2>&1
import time, torch, sys, os
sys.path.insert(0, os.getcwd())
import bench
p = bench.build_problem("ogbn-arxiv", seed=1)
eng = bench.make_engine(p, "cuda:0"); bench.init_state(p, eng); bench.new_model(p, eng)
def ms(ep, reps=40):
    for _ in range(5): eng.outer_step(ep)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): eng.outer_step(ep)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for mode in ("f32", "bf16x6", "bf16x3"):
    eng.set_operand_mode(mode)
    for rnd in range(3):
        for lvl in (1, 2, 0):
            eng.set_overlap(lvl)
            print(f"{mode:7s} overlap level {lvl}: feature epoch {ms(10):.3f} ms   PGE epoch {ms(0):.3f} ms", flush=True)
eng.set_overlap(1)
PY
grep -v "^/opt" gpurun_out/ov/ab_overlap.txt
