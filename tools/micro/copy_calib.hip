// FETCH_SIZE / WRITE_SIZE calibration for gfx950 (MI355X_MICROARCH.md, HBM section: "FETCH_SIZE reports exactly 1/2 of the
// bytes of a wide coalesced streaming read — calibrate on a known byte count in your own access pattern").
// Three kernels over the SAME 206 116 x 256 fp32 tensor (201.3 MiB = the A operand / the output of the PGE layer-2 product):
//   copy16  : 16 B per lane streaming read + 16 B per lane streaming write   (the streaming PGE kernels' pattern)
//   copy4   : 4 B per lane read + write
//   rowseg  : each lane reads 16 B at a 1 KiB row stride (64-B-per-4-lanes row segments: the tiled GEMM loader's pattern)
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE`; compare KiB with 206 116.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/copy_calib.hip -o tools/micro/copy_calib
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void copy16(const float4* __restrict__ a, float4* __restrict__ o, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) o[i] = a[i];
}
__global__ void copy4(const float* __restrict__ a, float* __restrict__ o, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) o[i] = a[i];
}
// rows x 256 floats; a workgroup of 256 threads covers 32 rows x one 32-float k slice per step (8 lanes x 16 B per row)
__global__ void rowseg(const float* __restrict__ a, float* __restrict__ o, long rows) {
  const int t = threadIdx.x, r = t >> 3, kq = (t & 7) * 4;
  float4 acc = make_float4(0, 0, 0, 0);
  for (long r0 = (long)blockIdx.x * 32; r0 < rows; r0 += (long)gridDim.x * 32) {
    const long row = r0 + r < rows ? r0 + r : rows - 1;
    for (int k0 = 0; k0 < 256; k0 += 32) {
      const float4 v = *reinterpret_cast<const float4*>(a + row * 256 + k0 + kq);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  o[(long)blockIdx.x * 256 + t] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
  const long rows = 206116, n = rows * 256;
  float *a, *o;
  hipMalloc(&a, n * 4);
  hipMalloc(&o, n * 4);
  hipMemset(a, 0, n * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(copy16, dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)o, n / 4);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("copy16 : %.1f us  read %.1f MiB + write %.1f MiB  -> %.2f TB/s\n", ms * 1e3, n * 4 / 1048576.0, n * 4 / 1048576.0, 2.0 * n * 4 / ms / 1e9);
    hipEventRecord(e0);
    hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, 0, a, o, n);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("copy4  : %.1f us  -> %.2f TB/s\n", ms * 1e3, 2.0 * n * 4 / ms / 1e9);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rowseg, dim3(2048), dim3(256), 0, 0, a, o, rows);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("rowseg : %.1f us  read %.1f MiB -> %.2f TB/s\n", ms * 1e3, n * 4 / 1048576.0, 1.0 * n * 4 / ms / 1e9);
  }
  return 0;
}
