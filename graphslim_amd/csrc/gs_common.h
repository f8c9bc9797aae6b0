// Internal helpers shared by the HIP translation units of libgslim_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/gslim.h"
#include "../../include/gslim_internal.h"

#define GS_WAVE 64

// last-error string (thread local); set by GS_FAIL / GS_HIP
void gs_set_error(const char* fmt, ...);

#define GS_FAIL(code, ...)        \
  do {                            \
    gs_set_error(__VA_ARGS__);    \
    return (code);                \
  } while (0)

#define GS_HIP(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) GS_FAIL(GS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

#define GS_LAUNCH_CHECK()                                                                          \
  do {                                                                                             \
    hipError_t e_ = hipGetLastError();                                                             \
    if (e_ != hipSuccess) GS_FAIL(GS_ERR_HIP, "kernel launch failed at %s:%d: %s", __FILE__, __LINE__, \
                                  hipGetErrorString(e_));                                          \
  } while (0)

#define GS_TRY(expr)        \
  do {                      \
    int r_ = (expr);        \
    if (r_ != GS_OK) return r_; \
  } while (0)

__host__ __device__ static inline int gs_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// One row [8] of the constants table of the generated dE2 operand (gs_panel.hip, panel_tile<..., AGEN>): per hidden column
// {is2, -mu2 is2, g2, be2, g2 is2 w2, g2 is2 dbe2 / N, g2 is2 dg2 / N, 0}.  Written by gs_panel_gen_consts_kernel or, in the
// engine's backward, by gs_pge_bwd_prep_kernel — ONE definition so that both produce the same bits.
__device__ static inline void gs_panel_gen_row(float mu2, float is2, float g2, float be2, float w2, float dbe2, float dg2, float invN,
                                               float* o) {
  const float gi = g2 * is2;
  o[0] = is2; o[1] = -mu2 * is2; o[2] = g2; o[3] = be2;
  o[4] = gi * w2; o[5] = gi * dbe2 * invN; o[6] = gi * dg2 * invN; o[7] = 0.f;
}

// ---- device helpers ------------------------------------------------------------------------
__device__ __forceinline__ float gs_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double gs_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float gs_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// counter-based RNG (splitmix64 finaliser over a 4-word key): stateless, replayable
__device__ __forceinline__ uint64_t gs_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t gs_rand_u32(uint64_t seed, uint32_t a, uint32_t b, uint32_t c) {
  uint64_t k = gs_mix64(seed ^ ((uint64_t)a << 32 | b));
  k = gs_mix64(k ^ (uint64_t)c * 0xD1342543DE82EF95ull);
  return (uint32_t)(k >> 32);
}
// unbiased-enough integer in [0, n) (n < 2^31): 64-bit multiply-shift
__device__ __forceinline__ uint32_t gs_rand_below(uint64_t seed, uint32_t a, uint32_t b, uint32_t c, uint32_t n) {
  return (uint32_t)(((uint64_t)gs_rand_u32(seed, a, b, c) * (uint64_t)n) >> 32);
}

#ifdef __HIPCC__
// One element of the image of the exact three-way bf16 split the exact-split row-panel kernels consume (gs_panel.hip,
// gs_panel2.hip): W [N][K] (element (n, k)) -> [N][K/16 sub-blocks][plane h, m, l][k half 0, 1][8] bf16, where the sub-block of k is
// 2 (k >> 5) + ((k >> 3) & 1), its half (k >> 4) & 1, its element k & 7.  h = bf16(w), m = bf16(w - h), l = bf16(w - h - m):
// 8 + 8 + 8 significant bits = the 24 of an fp32, so h + m + l == w exactly.  Shared by the stand-alone split kernel and by the
// launches that fold the split of a PGE weight into a kernel that runs before the product anyway (gs_kernels.h).
__device__ __forceinline__ void gs_split3_store(float w, int n, int k, int K, __bf16* __restrict__ out) {
  const __bf16 hb = (__bf16)w;
  const float r1 = w - (float)hb;
  const __bf16 mb = (__bf16)r1;
  const __bf16 lb = (__bf16)(r1 - (float)mb);
  const int sb = 2 * (k >> 5) + ((k >> 3) & 1), half = (k >> 4) & 1, e = k & 7;
  __bf16* o = out + ((long)n * (K >> 4) + sb) * 48 + half * 8 + e;
  o[0] = hb; o[16] = mb; o[32] = lb;
}
#endif

