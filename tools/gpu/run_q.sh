#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/q
timeout 1800 python -m pytest tests/test_gpu_panel.py tests/test_gpu_fullsize.py tests/test_gpu_operand_modes.py tests/test_gpu_engine.py tests/test_gpu_reduce.py -q -m gpu -x -s > gpurun_out/q/tests.log 2>&1
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --sustain-s 0 > gpurun_out/q/bench.json 2>gpurun_out/q/bench.err
python3 - <<'PY' > gpurun_out/q/ab_a1gen.txt 2>&1
import ctypes as C, time, torch, sys, os
sys.path.insert(0, os.getcwd())
import bench
from graphslim_amd import _lib
L = _lib.lib()
p = bench.build_problem("ogbn-arxiv", seed=1)
eng = bench.make_engine(p, "cuda:0"); bench.init_state(p, eng); bench.new_model(p, eng)
for mode in ("f32", "bf16x3", "bf16"):
    eng.set_operand_mode(mode)
    for rnd in range(2):
        for on in (0, 1):
            L.gs_gemm_set_panel_a1gen(on)
            for ep in (10, 0):
                for _ in range(5): eng.outer_step(ep)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(40): eng.outer_step(ep)
                torch.cuda.synchronize()
                print(f"{mode} generated-A1={on} epoch%50={ep}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms / iteration")
L.gs_gemm_set_panel_a1gen(1)
PY
tail -5 gpurun_out/q/tests.log; cat gpurun_out/q/ab_a1gen.txt; python3 -c "
import json;d=json.load(open('gpurun_out/q/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['isolated_launch_us'],{k:v['value'] for k,v in d['operand_modes'].items()})"
