// Microbenchmark (round 5): how should the non-MFMA work of an exact-split K step be issued beside its 48 MFMAs?
// One 512-thread workgroup per compute unit, 2 waves per SIMD, each wave a 32 x 256 tile (128 accumulators) like csrc/gs_panel2.hip:
// per 16-k step a wave issues 48 v_mfma_f32_32x32x16_bf16 (8 N tiles x 6 products, B fragments by ds_read_b128 from a 24-KB LDS image)
// and PREPARES the next step's A operand (2 ds_read_b128 + the three-way bf16 split of 8 floats: ~50 VALU).  No global traffic.
//   order 0: every wave [48 MFMA | prepare], one barrier per step                       (gs_tall_kernel SPLIT = 0)
//   order 1: waves 0-3 [prepare | 48 MFMA], waves 4-7 [48 MFMA | prepare], barrier     (SPLIT = 1, the product form of round 4)
//   order 2: every wave, prepare cut into 48 pieces FENCED between the MFMAs (sched_barrier around every MFMA), barrier
//   order 3: as 2 without the barrier
//   order 4: MFMAs only (no prepare), barrier: the floor
//   order 5: as 1 without the barrier
// Prints time per step, cycles per step (s_memtime of wave 0 / wave 4) and the executed TFLOP/s of the chip.
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_order issue_order.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define SLOT 24576

struct prep_state {      // the split of 8 floats in flight: inputs, residuals, outputs (pairs of bf16)
  float x[8], r[8];
  bf16x8 h, m, l;
};

// stage s (0..5) of pair pr (0..3): one or two VALU operations each
template <int PR, int ST>
__device__ __forceinline__ void prep_piece(prep_state& p) {
  constexpr int e0 = 2 * PR, e1 = 2 * PR + 1;
  if constexpr (ST == 0) { p.h[e0] = (__bf16)p.x[e0]; p.h[e1] = (__bf16)p.x[e1]; }
  if constexpr (ST == 1) { p.r[e0] = p.x[e0] - (float)p.h[e0]; p.r[e1] = p.x[e1] - (float)p.h[e1]; }
  if constexpr (ST == 2) { p.m[e0] = (__bf16)p.r[e0]; p.m[e1] = (__bf16)p.r[e1]; }
  if constexpr (ST == 3) { p.r[e0] = p.r[e0] - (float)p.m[e0]; p.r[e1] = p.r[e1] - (float)p.m[e1]; }
  if constexpr (ST == 4) { p.l[e0] = (__bf16)p.r[e0]; p.l[e1] = (__bf16)p.r[e1]; }
}
template <int K>
__device__ __forceinline__ void prep_slot(prep_state& p) {      // K = 0..47: slot behind MFMA K
  constexpr int PR = K / 12, ST = (K % 12) / 2;
  if constexpr ((K % 12) % 2 == 0 && ST < 5 && PR < 4) prep_piece<PR, ST>(p);
}
__device__ __forceinline__ void prep_all(prep_state& p) {
  prep_piece<0, 0>(p); prep_piece<1, 0>(p); prep_piece<2, 0>(p); prep_piece<3, 0>(p);
  prep_piece<0, 1>(p); prep_piece<1, 1>(p); prep_piece<2, 1>(p); prep_piece<3, 1>(p);
  prep_piece<0, 2>(p); prep_piece<1, 2>(p); prep_piece<2, 2>(p); prep_piece<3, 2>(p);
  prep_piece<0, 3>(p); prep_piece<1, 3>(p); prep_piece<2, 3>(p); prep_piece<3, 3>(p);
  prep_piece<0, 4>(p); prep_piece<1, 4>(p); prep_piece<2, 4>(p); prep_piece<3, 4>(p);
}

template <int J, int Q, bool FENCE, bool PREP>
__device__ __forceinline__ void mfma_one(f32x16 (&acc)[8], const bf16x8& ah, const bf16x8& am, const bf16x8& al, const bf16x8 (&b)[3], prep_state& p) {
  if constexpr (Q == 0) acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b[0], acc[J], 0, 0, 0);
  if constexpr (Q == 1) acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[2], acc[J], 0, 0, 0);
  if constexpr (Q == 2) acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[1], acc[J], 0, 0, 0);
  if constexpr (Q == 3) acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[0], acc[J], 0, 0, 0);
  if constexpr (Q == 4) acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[1], acc[J], 0, 0, 0);
  if constexpr (Q == 5) acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[0], acc[J], 0, 0, 0);
  if constexpr (FENCE) {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PREP) prep_slot<6 * J + Q>(p);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int J, bool FENCE, bool PREP>
__device__ __forceinline__ void mfma_tile(f32x16 (&acc)[8], const bf16x8& ah, const bf16x8& am, const bf16x8& al, const char* bs3, const int (&boff)[3],
                                          bf16x8 (&bq)[2][3], prep_state& p) {
  mfma_one<J, 0, FENCE, PREP>(acc, ah, am, al, bq[J & 1], p);
  mfma_one<J, 1, FENCE, PREP>(acc, ah, am, al, bq[J & 1], p);
  if constexpr (J + 1 < 8) {
#pragma unroll
    for (int P = 0; P < 3; ++P) bq[(J + 1) & 1][P] = *reinterpret_cast<const bf16x8*>(bs3 + (J + 1) * (32 * 96) + boff[P]);
    if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
  }
  mfma_one<J, 2, FENCE, PREP>(acc, ah, am, al, bq[J & 1], p);
  mfma_one<J, 3, FENCE, PREP>(acc, ah, am, al, bq[J & 1], p);
  mfma_one<J, 4, FENCE, PREP>(acc, ah, am, al, bq[J & 1], p);
  mfma_one<J, 5, FENCE, PREP>(acc, ah, am, al, bq[J & 1], p);
}

template <bool FENCE, bool PREP>
__device__ __forceinline__ void mfma_phase(f32x16 (&acc)[8], const bf16x8& ah, const bf16x8& am, const bf16x8& al, const char* bs3, const int (&boff)[3],
                                           prep_state& p) {
  bf16x8 bq[2][3];
#pragma unroll
  for (int P = 0; P < 3; ++P) bq[0][P] = *reinterpret_cast<const bf16x8*>(bs3 + boff[P]);
  mfma_tile<0, FENCE, PREP>(acc, ah, am, al, bs3, boff, bq, p);
  mfma_tile<1, FENCE, PREP>(acc, ah, am, al, bs3, boff, bq, p);
  mfma_tile<2, FENCE, PREP>(acc, ah, am, al, bs3, boff, bq, p);
  mfma_tile<3, FENCE, PREP>(acc, ah, am, al, bs3, boff, bq, p);
  mfma_tile<4, FENCE, PREP>(acc, ah, am, al, bs3, boff, bq, p);
  mfma_tile<5, FENCE, PREP>(acc, ah, am, al, bs3, boff, bq, p);
  mfma_tile<6, FENCE, PREP>(acc, ah, am, al, bs3, boff, bq, p);
  mfma_tile<7, FENCE, PREP>(acc, ah, am, al, bs3, boff, bq, p);
  if constexpr (!FENCE) {
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      if (j + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
  }
}

__device__ __forceinline__ void load_x(prep_state& p, const char* ab, int h, int u, int swa) {
  const f32x4 x0 = *reinterpret_cast<const f32x4*>(ab + (((4 * h + 2 * u) ^ swa) << 4));
  const f32x4 x1 = *reinterpret_cast<const f32x4*>(ab + (((4 * h + 2 * u + 1) ^ swa) << 4));
#pragma unroll
  for (int e = 0; e < 4; ++e) { p.x[e] = x0[e]; p.x[4 + e] = x1[e]; }
}

// one copy of the loop per role (a branch inside the step makes the compiler merge the accumulators of both orders: spills).
// ROLE: 0 [MFMA | prepare], 1 [prepare | MFMA], 2 prepare fenced between the MFMAs, 4 MFMAs only
template <int ROLE, bool BARRIER>
__device__ __forceinline__ float role_loop(const char* smem, int steps, int wave, int li, int h) {
  f32x16 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int sh3 = 3 * ((li >> 3) & 1);
  int boff[3];
#pragma unroll
  for (int P = 0; P < 3; ++P) boff[P] = ((2 * P + h + sh3) % 6) * 16;
  const int swa = (li >> 1) & 7;
  prep_state p;
  load_x(p, smem + 3 * SLOT + wave * 4096 + li * 128, h, 0, swa);
  prep_all(p);
  bf16x8 ah = p.h, am = p.m, al = p.l;
  int slot = 0;
  for (int s = 0; s < steps; ++s) {
    if constexpr (BARRIER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const char* bs3 = smem + slot * SLOT + li * 96;
    const char* ab = smem + 3 * SLOT + ((s & 1) * 8 + wave) * 4096 + li * 128;
    if constexpr (ROLE == 0) {
      mfma_phase<false, false>(acc, ah, am, al, bs3, boff, p);
      __builtin_amdgcn_sched_barrier(0);
      load_x(p, ab, h, s & 1, swa);
      prep_all(p);
    } else if constexpr (ROLE == 1) {
      load_x(p, ab, h, s & 1, swa);
      prep_all(p);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" :: "v"(p.h), "v"(p.m), "v"(p.l));
      mfma_phase<false, false>(acc, ah, am, al, bs3, boff, p);
    } else if constexpr (ROLE == 2) {
      load_x(p, ab, h, s & 1, swa);
      __builtin_amdgcn_sched_barrier(0);
      mfma_phase<true, true>(acc, ah, am, al, bs3, boff, p);
    } else {
      mfma_phase<false, false>(acc, ah, am, al, bs3, boff, p);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ROLE != 4) { ah = p.h; am = p.m; al = p.l; }
    slot = slot == 2 ? 0 : slot + 1;
  }
  float sa = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) sa += acc[j][r];
  return sa;
}

template <int ORDER>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int steps, int random_data) {
  __shared__ __attribute__((aligned(16))) char smem[3 * SLOT + 65536];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  for (int i = tid; i < (3 * SLOT + 65536) / 4; i += 512) {
    unsigned v = (unsigned)(i * 2654435761u) ^ (blockIdx.x * 40503u);
    if (random_data) {      // every mantissa / sign bit random, exponents spread over 2^-8 .. 2^7: what real operands look like to the multipliers
      v ^= v >> 15; v *= 2246822519u; v ^= v >> 13; v *= 3266489917u; v ^= v >> 16;
      const unsigned b0 = (v & 0x807fu) | ((119u + ((v >> 20) & 15u)) << 7), b1 = ((v >> 16) & 0x807fu) | ((119u + ((v >> 24) & 15u)) << 7);
      reinterpret_cast<unsigned*>(smem)[i] = i < 3 * SLOT / 4 ? (b0 | (b1 << 16)) : ((v & 0x807fffffu) | ((119u + ((v >> 27) & 15u)) << 23));
    } else
    reinterpret_cast<float*>(smem)[i] = i < 3 * SLOT / 4 ? __uint_as_float(0x3f803f80u ^ (v & 0x007f007fu))      // two bf16 near 1
                                                          : 1.f + (float)(v & 0xffff) * (1.f / 65536.f);
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float sacc;
  if constexpr (ORDER == 0) sacc = role_loop<0, true>(smem, steps, wave, li, h);
  else if constexpr (ORDER == 1) { if (wave < 4) sacc = role_loop<1, true>(smem, steps, wave, li, h); else sacc = role_loop<0, true>(smem, steps, wave, li, h); }
  else if constexpr (ORDER == 5) { if (wave < 4) sacc = role_loop<1, false>(smem, steps, wave, li, h); else sacc = role_loop<0, false>(smem, steps, wave, li, h); }
  else if constexpr (ORDER == 2) sacc = role_loop<2, true>(smem, steps, wave, li, h);
  else if constexpr (ORDER == 3) sacc = role_loop<2, false>(smem, steps, wave, li, h);
  else sacc = role_loop<4, true>(smem, steps, wave, li, h);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[(long)blockIdx.x * 512 + tid] = sacc;
  if (lane == 0 && (wave == 0 || wave == 4)) cyc[blockIdx.x * 2 + (wave >> 2)] = t1 - t0;
}

static int g_random = 0;
template <int ORDER>
void run(const char* what) {
  const int steps = 2000, blocks = 256;
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, (size_t)blocks * 512 * 4);
  hipMalloc(&cyc, blocks * 2 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<ORDER>, dim3(blocks), dim3(512), 0, 0, out, cyc, steps, g_random);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 5;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<ORDER>, dim3(blocks), dim3(512), 0, 0, out, cyc, steps, g_random);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  unsigned long long hc[512];
  hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
  double c0 = 0, c4 = 0;
  for (int b = 0; b < blocks; ++b) { c0 += hc[2 * b]; c4 += hc[2 * b + 1]; }
  c0 /= blocks * (double)steps; c4 /= blocks * (double)steps;
  const double flops = (double)blocks * 8 * steps * 48 * (32.0 * 32 * 16 * 2);
  printf("order %d  %-58s %7.3f us/step  cycles/step wave0 %6.0f wave4 %6.0f  (MFMA floor 3072)  clock %.2f GHz  %6.1f TFLOP/s executed\n", ORDER, what,
         ms * 1e3 / steps, c0, c4, c0 / (ms * 1e3 / steps) / 1e3, flops / ms / 1e9);
  hipFree(out); hipFree(cyc);
}
int main(int argc, char** argv) {
  g_random = argc > 1 ? atoi(argv[1]) : 0;
  printf("operand data: %s\n", g_random ? "random bits" : "values near 1 (few mantissa bits set)");
  run<4>("MFMAs only, barrier");
  run<0>("every wave [MFMA | prepare], barrier");
  run<1>("waves 0-3 [prepare | MFMA], 4-7 [MFMA | prepare], barrier");
  run<5>("the same, no barrier");
  run<2>("prepare fenced between the MFMAs, barrier");
  run<3>("prepare fenced between the MFMAs, no barrier");
  return 0;
}
