// Phase-removal experiments on the tiled GEMM at the PGE layer-2 shape (M = 206116, N = K = 256, both operands
// k-contiguous): the product kernel compiled with GS_TRIAGE_NO_GLOAD (no global loads after the first K step) and/or
// GS_TRIAGE_NO_STORE (no epilogue stores).  Results are garbage by construction; only the time matters.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I graphslim_amd/csrc [-DGS_TRIAGE_...] tools/micro/gemm_triage.hip
#include <cstdarg>
#include <cstdio>
#include "gslim.h"
#include "../../graphslim_amd/csrc/gs_gemm.hip"
void gs_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); }
int main() {
  const int n = 454, ph = 256; const long rows = (long)n * n;
  float *A, *B, *C; hipMalloc(&A, rows * ph * 4); hipMalloc(&B, ph * ph * 4); hipMalloc(&C, rows * ph * 4);
  hipMemset(A, 0, rows * ph * 4); hipMemset(B, 0, ph * ph * 4);
  gs_gemm_desc d = {};
  d.A = A; d.B = B; d.C = C; d.M = (int)rows; d.N = ph; d.K = ph; d.sAm = ph; d.sAk = 1; d.sBk = 1; d.sBn = ph; d.ldc = ph; d.alpha = 1.f;
  for (int mode = 0; mode < 2; ++mode) {
    if (mode == 1) { d.sBk = ph; d.sBn = 1; }      // (K,M): dY1
    for (int i = 0; i < 3; ++i) gs_gemm_f32(&d, nullptr);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) gs_gemm_f32(&d, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s %s: %.1f us  %.1f TFLOP/s\n", TAG, mode ? "(K,M)" : "(K,K)", ms / 20 * 1e3, 2.0 * rows * ph * ph / (ms / 20) / 1e9);
  }
  return 0;
}
