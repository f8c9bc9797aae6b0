#!/usr/bin/env python3
"""A/B of the persistent 256-row exact-split kernels (csrc/gs_panel2.hip) against the 128-row kernels of csrc/gs_panel.hip:
bitwise comparison of the products (plain H1 shape, ragged shapes, the PGE forward with generated A1, the fused marginals) and
their times, same process, interleaved.  usage: ab_tall.py [reps]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphslim_amd import _lib  # noqa: E402
from graphslim_amd.engine import gemm_desc  # noqa: E402

dev = "cuda:0"
L = _lib.lib()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timed(fn, reps=REPS):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def plain(M, K, relu=True, seed=0):
    torch.manual_seed(seed)
    a = torch.randn(M, K, device=dev)
    w = torch.randn(256, K, device=dev) / K ** 0.5
    b = torch.randn(256, device=dev)
    scratch = torch.empty(256 * K * 3 // 2 + 16, device=dev)
    outs, times = [], []
    for tall in (0, 1):
        L.gs_gemm_set_panel_tall(tall)
        out = torch.full((M, 256), float("nan"), device=dev)
        d, keep = gemm_desc(a, w, out, M, 256, K, (K, 1), (1, K), 256, prec=3, scratch=scratch, bias=b, relu=relu)
        assert L.gs_gemm_route(C.byref(d)) == 5
        times.append(timed(lambda: _lib.check(L.gs_gemm_f32(C.byref(d), _lib.stream()))))
        outs.append(out)
    L.gs_gemm_set_panel_tall(1)
    ref = a.double() @ w.double().T + b.double()
    if relu:
        ref = ref.clamp_min(0)
    err = float((outs[1].double() - ref).abs().max() / ref.abs().max())
    same = torch.equal(outs[0], outs[1])
    nbad = int((outs[0] != outs[1]).sum())
    print(f"plain M={M} K={K}: old {times[0]:.1f} us, tall {times[1]:.1f} us, bitwise {same} (differing {nbad}), err vs fp64 {err:.2e}", flush=True)
    return same


ok = True
for (M, K) in [(169343, 128), (2053, 64), (4096 + 37, 256), (300, 128), (64 * 4 * 3 + 1, 512), (206116, 256)]:
    ok &= plain(M, K, relu=(K != 256))

import bench  # noqa: E402

for wl in ("ogbn-arxiv", "reddit"):
    p = bench.build_problem(wl, seed=1)
    eng = bench.make_engine(p, dev)
    bench.init_state(p, eng)
    bench.new_model(p, eng)
    eng.outer_step(10)
    eng.set_operand_mode("bf16x6")
    res = {}
    for tall in (0, 1):
        L.gs_gemm_set_panel_tall(tall)
        t_f = timed(lambda: eng.pge_forward_part(False, 1))
        eng.pge_forward_part(False, 1)
        eng.pge_forward_part(False, 2)
        eng.pge_forward_part(False, 3)
        torch.cuda.synchronize()
        E2 = eng.buffer("E2").clone()
        mu2, is2, e3 = eng.buffer("mu2").clone(), eng.buffer("is2").clone(), eng.buffer("e3").clone()
        eng.stage_syn_match()
        eng.pge_backward_part(False, True, 1)
        t_b = timed(lambda: eng.pge_backward_part(False, True, 2))
        torch.cuda.synchronize()
        dfeat = eng.buffer("d_feat").clone()
        res[tall] = (E2, mu2, is2, e3, dfeat, t_f, t_b)
        if tall == 0:      # the marginals product of BOTH forms on this forward's state (same E2, same statistics): bitwise
            margs = []
            for tb in (0, 1):
                L.gs_gemm_set_panel_tall(tb)
                eng.buffer("ar_marg").fill_(float("nan"))
                eng.pge_backward_part(False, True, 2)
                torch.cuda.synchronize()
                margs.append(eng.buffer("ar_marg").clone())
            # (the 256-row kernels regenerate the ReLU mask from the folded terms Us, Vs: same function, fp32 rounding differs)
            dm = float((margs[0] - margs[1]).abs().max() / margs[0].abs().max())
            print(f"{wl}: fused marginals max |diff| / max {dm:.2e} (differing {int((margs[0] != margs[1]).sum())} of {margs[0].numel()}, "
                  f"NaN {int(torch.isnan(margs[1]).sum())})", flush=True)
            ok &= dm < 1e-5
    L.gs_gemm_set_panel_tall(1)
    a, b = res[0], res[1]
    print(f"{wl}: forward part 1 old {a[5]:.1f} us, tall {b[5]:.1f} us; backward part 2 old {a[6]:.1f} us, tall {b[6]:.1f} us")
    de2 = float((a[0] - b[0]).abs().max() / a[0].abs().max())
    same_e2 = de2 < 2e-6
    print(f"  E2 max |diff| / max {de2:.2e} (differing {int((a[0] != b[0]).sum())} of {a[0].numel()}), "
          f"mu2 max rel {float(((a[1] - b[1]).abs() / a[1].abs().clamp_min(1e-6)).max()):.2e}, "
          f"is2 max rel {float(((a[2] - b[2]).abs() / a[2].abs()).max()):.2e}, e3 max abs {float((a[3] - b[3]).abs().max()):.2e}, "
          f"d_feat max rel-of-max {float((a[4] - b[4]).abs().max() / a[4].abs().max()):.2e}", flush=True)
    ok &= same_e2
    del eng
print("ALL OK" if ok else "MISMATCH")
