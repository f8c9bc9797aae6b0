#!/bin/bash
# persistent kernels: every wave interleaves the next sub-step's operand preparation with its MFMAs (triage bit 10) vs the default split
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-inter1}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_panel.py -x -q -m gpu -k tall > $O/tests.log 2>&1; tail -2 $O/tests.log
for t in 0 1024 128 0 1024; do
  echo "== triage $t" | tee -a $O/inter.txt
  GS_TALL=1,0,$t timeout 300 python tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -E "PGE forward \(inference\)|real-side|outer_step \(feature" | tee -a $O/inter.txt
done
python - <<'PY' 2>/dev/null | tee -a $O/inter.txt
import ctypes as C, sys, os, torch
sys.path.insert(0, os.getcwd())
from graphslim_amd import _lib
from graphslim_amd.engine import gemm_desc
L = _lib.lib(); dev = "cuda:0"
M, K = 206116, 256
torch.manual_seed(0)
a = torch.randn(M, K, device=dev); w = torch.randn(256, K, device=dev) / 16; b = torch.randn(256, device=dev)
scratch = torch.empty(256 * K * 3 // 2 + 16, device=dev)
outs = {}
for t in (0, 1024):
    L.gs_panel_tall_debug(0, 1, t)
    out = torch.full((M, 256), float("nan"), device=dev)
    d, keep = gemm_desc(a, w, out, M, 256, K, (K, 1), (1, K), 256, prec=3, scratch=scratch, bias=b)
    for _ in range(3): _lib.check(L.gs_gemm_f32(C.byref(d), _lib.stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): _lib.check(L.gs_gemm_f32(C.byref(d), _lib.stream()))
    e1.record(); torch.cuda.synchronize()
    outs[t] = out
    print(f"plain M={M} K={K} triage {t}: {e0.elapsed_time(e1) * 1e3 / 30:.1f} us")
L.gs_panel_tall_debug(0, 1, 0)
print("bitwise equal:", torch.equal(outs[0], outs[1]))
PY
