#!/usr/bin/env python3
"""Full-size oracle parity of one workload in each operand mode (the numbers tests/test_gpu_fullsize.py gates): which mode, which
tensor.  usage: diag_fullsize.py [workload] [epoch ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_fullsize as T  # noqa: E402

from graphslim_amd import _lib  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cora"
epochs = [int(a) for a in sys.argv[2:]] or [10, 0]
L = _lib.lib()
knobs = []
for env, fn in (("GS_DW1_GEN", "gs_gcond_set_dw1_generated"), ("GS_PANEL_MARG", "gs_gemm_set_panel_marg"), ("GS_PANEL_A1GEN", "gs_gemm_set_panel_a1gen"),
                ("GS_PANEL", "gs_gemm_set_panel"), ("GS_TALL", "gs_gemm_set_panel_tall")):
    if os.environ.get(env) is not None:
        getattr(L, fn)(int(os.environ[env]))
        knobs.append(f"{fn}({os.environ[env]})")
print("knobs:", knobs or "defaults", flush=True)
modes = os.environ.get("GS_MODES", "default,f32,bf16x6,bf16x3").split(",")
for mode in [None if m == "default" else m for m in modes]:
    for ep in epochs:
        r = T.run_config(wl, ep, operand_mode=mode)
        print(f"{wl} mode {mode} epoch {ep}: rel_loss {r['rel_loss']:.2e} g_feat {r['g_feat']:.2e} g_pge "
              f"{[(i, float(f'{e:.2e}')) for i, e in r['g_pge']]} upd_well_max {r['upd_well_max']:.2e} adj {r['adj']:.2e}", flush=True)
