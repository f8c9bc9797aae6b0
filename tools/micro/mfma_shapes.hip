// Microbenchmark (round 5): does the 16x16x32 bf16 MFMA shape deliver more FLOP/s than 32x32x16 under THIS chip's power limit?
// (the exact-split kernels are bounded by the power limit at ~1.5 PFLOP/s on random operands with 32x32x16.)
// One 512-thread workgroup per compute unit (2 waves per SIMD), register operands only (no LDS, no memory): each wave issues
// independent accumulation chains back to back.  Per iteration and wave: 8 x 32x32x16 (8 accumulators of 16 registers) or
// 16 x 16x16x32 (16 accumulators of 4 registers) — the same FLOPs.  Operands: random bf16 bits in (-2, 2) or values near 1.
// Prints time, in-kernel clock (s_memtime / s_memrealtime) and executed TFLOP/s of the chip for both shapes.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_shapes mfma_shapes.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bf16x8 operand(unsigned seed, int random_data) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned h = mix(seed * 8u + i);
    // random: sign + 7 random mantissa bits, exponent in {126, 127} (|x| in [0.5, 2)); near 1: +-1 exactly
    const unsigned short bits = random_data ? (unsigned short)(((h & 1u) << 15) | ((126u + ((h >> 1) & 1u)) << 7) | ((h >> 2) & 0x7fu))
                                            : (unsigned short)(((h & 1u) << 15) | (127u << 7));
    v[i] = __builtin_bit_cast(__bf16, bits);
  }
  return v;
}

template <int SHAPE>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, int steps, int random_data) {
  const unsigned tid = blockIdx.x * 512 + threadIdx.x;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = operand(tid * 8 + i, random_data); b[i] = operand(tid * 8 + 4 + i, random_data); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float sink = 0.f;
  if constexpr (SHAPE == 0) {
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int s = 0; s < steps; s += 4) {      // (operand indices static: no register-indexed moves in the loop)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j & 3], b[(j + u) & 3], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sink += acc[j][r];
  } else {
    f32x4 acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    for (int s = 0; s < steps; s += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[(j + u) & 3], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) sink += acc[j][r];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[tid] = sink;
  if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
void run(const char* what, int random_data) {
  const int steps = 400000, blocks = 256;
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, (size_t)blocks * 512 * 4);
  hipMalloc(&cyc, blocks * 2 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(512), 0, 0, out, cyc, steps, random_data);      // ~2 s: sustained clocks
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 5;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(512), 0, 0, out, cyc, steps, random_data);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  unsigned long long hc[512];
  hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
  double c = 0, r = 0;
  for (int b = 0; b < blocks; ++b) { c += hc[2 * b]; r += hc[2 * b + 1]; }
  const double flops = (double)blocks * 8 * steps * 8 * (32.0 * 32 * 16 * 2);      // (16 x 16x16x32 = the same FLOPs)
  printf("%-12s %-22s %8.3f ms  clock %.2f GHz  cycles per 32x32x16-equivalent %5.1f  %7.1f TFLOP/s executed\n", what,
         random_data ? "random operands" : "operands +-1", ms, (c / r) * 0.1, c / blocks / ((double)steps * 8) * 1.0 / 2.0, flops / ms / 1e9);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int rd = 0; rd < 2; ++rd) {
    run<0>("32x32x16", rd);
    run<1>("16x16x32", rd);
  }
  return 0;
}
