// Cost of a device-wide barrier inside one persistent kernel (all workgroups co-resident): release fence + one atomic per
// workgroup + spin on the counter + acquire fence, against the ~5.5 us a dependent tiny launch costs in an in-order stream.
// Each phase every workgroup also writes a few KB and reads what the others wrote in the previous phase (checked at the end), so
// that the fences are doing real work.  usage: grid_barrier [workgroups=120] [phases=200]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();                                                      // release this workgroup's writes (agent scope)
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 20000000) { ok = false; break; }                      // never hang the box: bail out
    }
    __threadfence();
  }
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(256) void persistent(unsigned* counter, float* buf, int per_wg, int phases, int* bad) {
  const int G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
  for (int ph = 0; ph < phases; ++ph) {
    // write my slice for this phase, then after the barrier read the slice of the NEXT workgroup and check it
    float* mine = buf + ((long)(ph & 1) * G + b) * per_wg;
    for (int i = t; i < per_wg; i += 256) mine[i] = (float)(ph * 1000 + b) + 0.001f * i;
    if (!grid_barrier(counter, (unsigned)(G * (ph + 1)))) { if (t == 0) atomicAdd(bad, 1000000); return; }
    const int nb = (b + 1) % G;
    const float* other = buf + ((long)(ph & 1) * G + nb) * per_wg;
    for (int i = t; i < per_wg; i += 256)
      if (other[i] != (float)(ph * 1000 + nb) + 0.001f * i) atomicAdd(bad, 1);
  }
}

__global__ void tiny(float* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.f; }

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 120, phases = argc > 2 ? atoi(argv[2]) : 200, per_wg = 2048;
  unsigned* counter; float* buf; int* bad;
  hipMalloc(&counter, 4); hipMalloc(&buf, sizeof(float) * 2 * G * per_wg); hipMalloc(&bad, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(counter, 0, 4); hipMemset(bad, 0, 4);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(persistent, dim3(G), dim3(256), 0, 0, counter, buf, per_wg, phases, bad);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    int hb = 0; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("persistent: %d workgroups, %d phases: %.1f us total, %.2f us per phase (write 8 KB + barrier + read 8 KB), mismatches %d\n",
           G, phases, ms * 1e3, ms * 1e3 / phases, hb);
  }
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    for (int i = 0; i < phases; ++i) hipLaunchKernelGGL(tiny, dim3(G), dim3(256), 0, 0, buf);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("launch chain: %d dependent launches of %d workgroups: %.2f us per launch\n", phases, G, ms * 1e3 / phases);
  }
  return 0;
}
