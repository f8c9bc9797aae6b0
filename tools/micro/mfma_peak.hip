// Microbenchmark: what the fp32 matrix pipe delivers with NO memory traffic (v_mfma_f32_32x32x2_f32, 4 independent
// accumulators per wave like the GEMM's 64x64 wave tile), for 1..4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs_per_cu) {
  int iters = 2000, blocks = 256 * wgs_per_cu;
  float* out; hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 8 * NACC * (32.0 * 32 * 2 * 2);
  printf("NACC %d, %d workgroups/CU (%d waves/SIMD): %.3f ms  %.1f TFLOP/s\n", NACC, wgs_per_cu, wgs_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  for (int w = 1; w <= 4; ++w) run<4>(w);
  run<1>(1); run<1>(4); run<2>(2);
  return 0;
}
