// Semantics probe of ds_read_b64_tr_b16 (gfx950) through __builtin_amdgcn_ds_read_tr16_b64_v4i16: LDS holds value = 100 row + col of a
// [64][64] 16-bit image; lane 16 g + 4 q + p supplies the address of (row 4 (g >> 1) + q, columns 16 (g & 1) + 4 p ..).  Prints what
// every lane receives: expected lane i of a group = column i of the block's 4 rows, row q in element q.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)(100 * (i / 64) + (i % 64));
  __syncthreads();
  const int L = threadIdx.x, g = L >> 4, q = (L >> 2) & 3, p = L & 3;
  const short* addr = lds + (4 * (g >> 1) + q) * 64 + 16 * (g & 1) + 4 * p;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int e = 0; e < 4; ++e) out[L * 4 + e] = v[e];
}
int main() {
  short* d; short h[256];
  (void)hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int L = 0; L < 64; ++L) printf("lane %2d: %4d %4d %4d %4d\n", L, h[4 * L], h[4 * L + 1], h[4 * L + 2], h[4 * L + 3]);
  return 0;
}
