// The inner SGC steps of one outer iteration as ONE persistent launch (VERDICT r05 "next" 1).
//
// Reference: graphslim/condensation/gcond.py:63-72 — `inner_loop` steps of {forward on (X~, A^), NLL over all n' rows, backward, Adam on
// theta}; the model is SGC with two transforms (models/sgc.py:37-57), theta = [W1 d x h | b1 h | Wl h x C | bl C].  As launches
// (gs_engine.hip inner_steps_fused) a step is seven dependent launches of ~0.12 GFLOP: 21 launches, 139 us alone on the chip and
// 210-250 us beside the hoisted training forward — each launch fills and drains 256 CUs for 454 rows.  Here the whole stage is one
// launch of G ~ 48 workgroups (what the hoisted forward's 13/16 grid leaves free) that stay resident; a step is FOUR phases
// separated by grid barriers:
//
//   phase A  (row block of 16 nodes per workgroup)   H1^T = relu(W1^T X^T + b1)   and, chained through the accumulator registers,
//                                                    T^T = Wl^T H1^T                     -> publishes H1^T, T^T
//   phase B  (row block)                             Z^T = T^T Q^T + bl (x) rs ; G = (softmax(Z) - Y) / n'      -> publishes G^T
//   phase C  (row block)                             R^T = G^T Q ; dH1^T = (Wl R^T) * [H1 > 0]                   -> publishes R^T, dH1^T
//   phase D  (one 16 x 16 output tile per wave)      gW1 = X^T dH1, gWl = H1^T R, gb1 = 1^T dH1, gbl = rs^T G, each with the Adam
//                                                    step of its elements in the epilogue            -> publishes W1^T, Wl^T, b1, bl
//
// (the algebra of gs_chain.hip's header: the n' x n' x h products re-associated into n' x n' x C ones).  Everything is exact fp32 on
// the fp32 matrix pipe (v_mfma_f32_16x16x4_f32), every sum in a fixed order: two runs are bitwise equal.
//
// Layouts.  Every operand a phase reads along its K dimension is stored K-CONTIGUOUS, so that each lane fetches 16 bytes per four
// MFMA k-steps: the published activations are kept TRANSPOSED ([feature][node], row stride npad = ceil16(n'), pad columns zero) —
// which is also what the MFMA result layout wants: a lane of a 16x16x4 result holds 4 consecutive ROWS of one column, so storing the
// transpose turns them into one 16-byte store (via a small LDS tile where the lane dimension is the node).  theta is kept twice:
// canonical (what the rest of the engine reads; plain stores, nobody else reads it inside the launch) and as the transposed operands
// W1^T [h][dpad], Wl^T [Cp][h] the next step's phase A / C read.
//
// Inter-workgroup visibility (MI355X_MICROARCH.md "Workgroup dispatch ... visibility", cdna_hip_programming.md Guideline 16): every
// byte another workgroup reads is stored WRITE-THROUGH (sc1: raw buffer store with aux 16, or a relaxed agent-scope atomic store),
// every storing wave drains (s_waitcnt vmcnt(0)) before the workgroup's barrier, ONE lane then adds to a monotonic arrival counter,
// polls it relaxed, and issues ONE agent-scope acquire (L1 invalidate) + wait before the workgroup's second barrier; plain loads
// behind it.  The counter is never reset inside a launch (target = base + phase * G, base handed in by the host), spins are bounded:
// a timeout sets the error word gs_gcond_read_loss checks.
#include "gs_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u_t __attribute__((ext_vector_type(4), aligned(4)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

#define GS_IP_MAX_STEPS 16
#define GS_IP_THREADS 512

struct gs_inner_args {
  int n, npad, d, dpad, h, C, Cp, steps, G;
  const float* X; long ldx;            // X~ [n][ldx]
  const float* Q; const float* QT;     // [n][npad] each: Q = A^ A^ and its transpose, pad columns zero
  const int* labels;                   // [n]
  float* theta; float* m; float* v;    // canonical parameters and Adam moments
  float* g;                            // canonical gradient of the LAST step (the engine's "g_model" buffer)
  long oW1, ob1, oWl, obl;
  float* XT;                           // [dpad][npad]
  float* W1T;                          // [h][dpad]
  float* WlT;                          // [Cp][h]
  float* pb;                           // [h + Cp]: b1 | bl (published copies)
  float* TT; float* GT; float* RT;     // [Cp][npad]
  float* H1T; float* dH1T;             // [h][npad]
  float* rs;                           // [npad] row sums of Q
  unsigned* sync;                      // [0] arrival counter, [16] error word
  unsigned base;                       // counter value at launch (host keeps it in step)
  unsigned long long* trace;           // optional [G][4 * barriers + 2] s_memrealtime stamps of thread 0 (gs_inner_persist_trace)
  float step_size[GS_IP_MAX_STEPS], inv_sqrt_bc2[GS_IP_MAX_STEPS];
  float b1, b2, eps, inv_n;
};

namespace {

__device__ __forceinline__ void st16_sc1(float* base, long nbytes, long off_bytes, f32x4 v) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)nbytes, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), r, (int)off_bytes, 0, 16);
}
__device__ __forceinline__ void st4_sc1(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// grid barrier number `idx` (1-based) of this launch
__device__ __forceinline__ void grid_sync(const gs_inner_args& a, unsigned idx) {
  unsigned long long* tr = a.trace ? a.trace + (long)blockIdx.x * 128 + 4 * (idx - 1) + 1 : nullptr;
  if (tr && threadIdx.x == 0) tr[0] = wall_clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned target = a.base + idx * (unsigned)a.G;
    if (tr) tr[1] = wall_clock64();
    __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    // (wrap-safe comparison: the counter runs ahead of `target` by at most one phase)
    while ((int)(__hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 40000000L) {      // seconds: a workgroup of this grid never became resident, or the counter is out of step
        __hip_atomic_store(a.sync + 16, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    if (tr) tr[2] = wall_clock64();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (tr && threadIdx.x == 0) tr[3] = wall_clock64();
}

// 16 bytes along k of a K-contiguous operand row whose length (in floats, multiple of 4 up to the padded extent) is `lim`
__device__ __forceinline__ f32x4 ldk(const float* __restrict__ row, int k, int lim) {
  if (k + 4 <= lim) return *reinterpret_cast<const f32x4u_t*>(row + k);
  f32x4 r;
  r.x = k < lim ? row[k] : 0.f; r.y = k + 1 < lim ? row[k + 1] : 0.f; r.z = k + 2 < lim ? row[k + 2] : 0.f; r.w = k + 3 < lim ? row[k + 3] : 0.f;
  return r;
}

#define GS_MFMA4(acc, av, bv)                                             \
  do {                                                                    \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0); \
  } while (0)

// 64 x 64 tile of src [R][Cc] (row stride sld) -> dst transposed, zero beyond src's extent, 16-byte write-through stores.
// dst has dR rows (>= Cc, padded) of stride dld (>= R padded, multiple of 16).
__device__ __forceinline__ void transpose_tile(const float* __restrict__ src, long sld, int R, int Cc, int r0, int c0, float* dst, long dld,
                                               int dR, long dst_bytes, float (*tb)[65]) {
  const int t = threadIdx.x;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4096 / GS_IP_THREADS; ++q) {
    const int i = (t >> 6) + (GS_IP_THREADS / 64) * q, j = t & 63;
    const int r = r0 + i, c = c0 + j;
    tb[i][j] = (r < R && c < Cc) ? src[(long)r * sld + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 1024 / GS_IP_THREADS; ++q) {
    const int idx = t + GS_IP_THREADS * q;      // 0 .. 1023: (j, i4)
    const int j = idx >> 4, i4 = (idx & 15) * 4;
    const int dr = c0 + j, dc = r0 + i4;
    if (dr < dR && dc < dld) {
      f32x4 v;
      v.x = tb[i4][j]; v.y = tb[i4 + 1][j]; v.z = tb[i4 + 2][j]; v.w = tb[i4 + 3][j];
      st16_sc1(dst, dst_bytes, ((long)dr * dld + dc) * 4, v);
    }
  }
}

}  // namespace

#define GS_IP_WAVES (GS_IP_THREADS / 64)
// Per-phase lane coordinates from an OPAQUE thread id: what is derived from the plain one (row pointers, clamped indices of every phase)
// is loop-invariant across the step loop, and left visible the compiler keeps all of it live across the phases (spills at 512 threads).
#define GS_IP_LANES()                                                                                        \
  int tid_ = threadIdx.x;                                                                                    \
  asm volatile("" : "+v"(tid_));                                                                             \
  const int tid = tid_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lm = lane & 15, kq = lane >> 4; \
  const int my_node = min(node0 + lm, n - 1);                                                                \
  const bool node_live = node0 + lm < n

// LDS (floats): hbuf [256][20] | red [8 waves][3][64] f32x4 | zt [48][20] | rsl [16]; phase 0's transpose tile [64][65] aliases hbuf + red
__global__ __launch_bounds__(GS_IP_THREADS) void gs_inner_persist_kernel(gs_inner_args a) {
  __shared__ __attribute__((aligned(16))) float smem[256 * 20 + GS_IP_WAVES * 3 * 64 * 4 + 48 * 20 + 16];
  float* const hbuf = smem;                                                       // [256][20]
  f32x4* const red = reinterpret_cast<f32x4*>(smem + 256 * 20);                   // [waves][3][64]
  float* const zt = smem + 256 * 20 + GS_IP_WAVES * 3 * 64 * 4;                   // [48][20]
  float* const rsl = zt + 48 * 20;                                                // [16]
  const int tid = threadIdx.x;
  const int wg = blockIdx.x, G = a.G;
  const int n = a.n, npad = a.npad, d = a.d, dpad = a.dpad, h = a.h, C = a.C, Cp = a.Cp;
  const int nrb = npad / 16, nht = h / 16, nct = Cp / 16, ndt = dpad / 16, nkb = npad / 16;
  const long actT_bytes = (long)Cp * npad * 4, hidT_bytes = (long)h * npad * 4;
  unsigned bar = 0;
  if (a.trace && tid == 0) a.trace[(long)wg * 128] = wall_clock64();

  // ---- phase 0: the transposed operands of the current theta and of X~ ------------------------------------------------------
  {
    float (*tb)[65] = reinterpret_cast<float (*)[65]>(smem);      // (aliases hbuf + red: 64 x 65 floats, see the static_assert below)
    const int tx_r = (n + 63) / 64, tx_c = (d + 63) / 64;          // X [n][d] -> XT [dpad][npad]
    const int tw_r = (d + 63) / 64, tw_c = (h + 63) / 64;          // W1 [d][h] -> W1T [h][dpad]
    const int tl_r = (h + 63) / 64;                                // Wl [h][C] -> WlT [Cp][h]
    const int nt = tx_r * tx_c + tw_r * tw_c + tl_r;
    // (the theta tiles first: phase A needs them; the X~ tiles are only read in phase D)
#pragma unroll 1
    for (int t = wg; t < nt; t += G) {
      if (t < tw_r * tw_c) {
        transpose_tile(a.theta + a.oW1, h, d, h, (t / tw_c) * 64, (t % tw_c) * 64, a.W1T, dpad, h, (long)h * dpad * 4, tb);
      } else if (t < tw_r * tw_c + tl_r) {
        const int u = t - tw_r * tw_c;
        transpose_tile(a.theta + a.oWl, C, h, C, u * 64, 0, a.WlT, h, Cp, (long)Cp * h * 4, tb);
      } else {
        const int u = t - tw_r * tw_c - tl_r;
        transpose_tile(a.X, a.ldx, n, d, (u / tx_c) * 64, (u % tx_c) * 64, a.XT, npad, dpad, (long)dpad * npad * 4, tb);
      }
    }
    if (wg == G - 1) {
      for (int i = tid; i < h + Cp; i += GS_IP_THREADS)
        st4_sc1(a.pb + i, i < h ? a.theta[a.ob1 + i] : (i - h < C ? a.theta[a.obl + (i - h)] : 0.f));
    }
    __syncthreads();
  }
  grid_sync(a, ++bar);

  const bool rowwg = wg < nrb;
  const int node0 = wg * 16;
  // (the node on a lane is clamped: rows beyond n' are computed on finite data and published as zeros)
  unsigned mbits = 0;                               // [H1 > 0] of this lane's 2 x 4 hidden entries (phase A -> phase C)

#pragma unroll 1
  for (int step = 0; step < a.steps; ++step) {
    // ================= phase A: H1^T tiles of this row block, T^T chained through the registers =================
    if (rowwg) {
      GS_IP_LANES();
      f32x4 accT[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) accT[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* __restrict__ xrow = a.X + (long)my_node * a.ldx;
      // this wave's hidden tiles: wave, wave + 8 (clamped to a valid tile: the duplicate's results are dropped)
      const int t0i = min(wave, nht - 1), t1i = min(wave + GS_IP_WAVES, nht - 1);
      const bool has0 = wave < nht, has1 = wave + GS_IP_WAVES < nht;
      const float* __restrict__ w0 = a.W1T + (long)(16 * t0i + lm) * dpad;      // A': W1^T[hh][k]
      const float* __restrict__ w1 = a.W1T + (long)(16 * t1i + lm) * dpad;
      f32x4 acc0 = *reinterpret_cast<const f32x4*>(a.pb + 16 * t0i + 4 * kq);   // + b1 (row = hidden unit 4 kq + reg)
      f32x4 acc1 = *reinterpret_cast<const f32x4*>(a.pb + 16 * t1i + 4 * kq);
      f32x4 wl0[3], wl1[3];                                                     // A' of the chained product: Wl^T[c][hh]
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const long r = (long)(16 * min(j, nct - 1) + lm) * h + 4 * kq;
        wl0[j] = *reinterpret_cast<const f32x4*>(a.WlT + r + 16 * t0i);
        wl1[j] = *reinterpret_cast<const f32x4*>(a.WlT + r + 16 * t1i);
      }
#pragma unroll 1
      for (int kb0 = 0; kb0 < ndt; kb0 += 8) {      // one round of loads per 128 k: X~ fragment once, both tiles' W1^T fragments
        f32x4 xv[8], wa[8], wb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int kb = min(kb0 + q, ndt - 1);
          const int k = 16 * kb + 4 * kq;
          xv[q] = ldk(xrow, k, d);
          wa[q] = *reinterpret_cast<const f32x4*>(w0 + k);
          wb[q] = *reinterpret_cast<const f32x4*>(w1 + k);
          if (kb0 + q >= ndt) xv[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) { GS_MFMA4(acc0, wa[q], xv[q]); GS_MFMA4(acc1, wb[q], xv[q]); }
      }
      mbits = 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool has = i ? has1 : has0;
        const f32x4 acc = i ? acc1 : acc0;
        const int hh0 = 16 * (i ? t1i : t0i);
        // relu; the tile (rows = hidden units hh0 + 4 kq + r, column = node lm) is the B operand of T^T's k block
        f32x4 hv;
        hv.x = fmaxf(acc.x, 0.f); hv.y = fmaxf(acc.y, 0.f); hv.z = fmaxf(acc.z, 0.f); hv.w = fmaxf(acc.w, 0.f);
        if (has) {
          mbits |= ((hv.x > 0.f ? 1u : 0u) | (hv.y > 0.f ? 2u : 0u) | (hv.z > 0.f ? 4u : 0u) | (hv.w > 0.f ? 8u : 0u)) << (4 * i);
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            if (i) GS_MFMA4(accT[j], wl1[j], hv);
            else GS_MFMA4(accT[j], wl0[j], hv);
          }
          // H1^T tile -> LDS [hh][node] for the 16-byte publish below
#pragma unroll
          for (int r = 0; r < 4; ++r) hbuf[(hh0 + 4 * kq + r) * 20 + lm] = node_live ? hv[r] : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) red[(wave * 3 + j) * 64 + lane] = accT[j];
      __syncthreads();
      // publish H1^T rows (16 bytes: 4 nodes of one hidden unit)
      for (int idx = tid; idx < h * 4; idx += GS_IP_THREADS) {
        const int hh = idx >> 2, q4 = (idx & 3) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&hbuf[hh * 20 + q4]);
        st16_sc1(a.H1T, hidT_bytes, ((long)hh * npad + node0 + q4) * 4, v);
      }
      // T^T: sum of the waves' partial tiles (wave order), then 16-byte publish through zt
      if (tid < 64 * nct) {
        const int j = tid >> 6, l = tid & 63;
        f32x4 s = red[(0 * 3 + j) * 64 + l];
#pragma unroll
        for (int w = 1; w < GS_IP_WAVES; ++w) s += red[(w * 3 + j) * 64 + l];
        const int lml = l & 15, kql = l >> 4;
        const bool live = node0 + lml < n;
#pragma unroll
        for (int r = 0; r < 4; ++r) zt[(16 * j + 4 * kql + r) * 20 + lml] = live ? s[r] : 0.f;
      }
      __syncthreads();
      if (tid < Cp * 4) {
        const int c = tid >> 2, q4 = (tid & 3) * 4;
        st16_sc1(a.TT, actT_bytes, ((long)c * npad + node0 + q4) * 4, *reinterpret_cast<const f32x4*>(&zt[c * 20 + q4]));
      }
    }
    grid_sync(a, ++bar);

    // ================= phases B and C: [Cp x 16] = act^T [Cp x n'] . (Q or Q^T rows of this block)^T =================
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
      if (rowwg) {
        GS_IP_LANES();
        const float* __restrict__ act = ph == 0 ? a.TT : a.GT;
        const float* __restrict__ qrow = (ph == 0 ? a.Q : a.QT) + (long)my_node * npad;     // B': Q[node][j] / Q^T[node][j]
        f32x4 acc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float rsum = 0.f;
        const bool want_rs = ph == 0 && step == 0;
        // phase C's second product reads Wl^T along the lanes (4-byte loads): requested here, ahead of the k loop's round trip
        f32x4 wlc[2][3];
        if (ph == 1) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int hh0 = 16 * min(wave + GS_IP_WAVES * i, nht - 1);
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
              const float* wp = a.WlT + (long)(16 * min(s2, nct - 1) + 4 * kq) * h + hh0 + lm;
              wlc[i][s2].x = wp[0]; wlc[i][s2].y = wp[h]; wlc[i][s2].z = wp[2 * h]; wlc[i][s2].w = wp[3 * h];
            }
          }
        }
#pragma unroll 1
        for (int kb0 = wave; kb0 < nkb; kb0 += 4 * GS_IP_WAVES) {      // this wave's k blocks: wave, wave + 8, ...: four per round of loads
          f32x4 bq[4], aq[4][3];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int kb = min(kb0 + GS_IP_WAVES * q, nkb - 1);
            const int k = 16 * kb + 4 * kq;
            bq[q] = *reinterpret_cast<const f32x4*>(qrow + k);
#pragma unroll
            for (int j = 0; j < 3; ++j) aq[q][j] = *reinterpret_cast<const f32x4*>(act + (long)(16 * min(j, nct - 1) + lm) * npad + k);
            if (kb0 + GS_IP_WAVES * q >= nkb) bq[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (want_rs) rsum += (bq[q].x + bq[q].y) + (bq[q].z + bq[q].w);
#pragma unroll
            for (int j = 0; j < 3; ++j) GS_MFMA4(acc[j], aq[q][j], bq[q]);
          }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) red[(wave * 3 + j) * 64 + lane] = acc[j];
        if (want_rs) {      // rs = Q 1 of this block's nodes: lanes of one node over kq, then the waves (fixed order)
          rsum += __shfl_xor(rsum, 16, 64);
          rsum += __shfl_xor(rsum, 32, 64);
          if (lane < 16) hbuf[wave * 16 + lane] = rsum;
        }
        __syncthreads();
        if (want_rs && tid < 16) {
          float r = hbuf[tid];
#pragma unroll
          for (int w = 1; w < GS_IP_WAVES; ++w) r += hbuf[16 * w + tid];
          rsl[tid] = r;
          st4_sc1(a.rs + node0 + tid, node0 + tid < n ? r : 0.f);
        }
        if (tid < 64 * nct) {
          const int j = tid >> 6, l = tid & 63;
          f32x4 s = red[(0 * 3 + j) * 64 + l];
#pragma unroll
          for (int w = 1; w < GS_IP_WAVES; ++w) s += red[(w * 3 + j) * 64 + l];
          const int lml = l & 15, kql = l >> 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) zt[(16 * j + 4 * kql + r) * 20 + lml] = s[r];
        }
        __syncthreads();
        if (ph == 0) {
          // ---- softmax over the classes + NLL gradient; wave 0: lane = (node lm, part kq), classes kq, kq + 4, ... ----
          if (wave == 0) {
            const float rsn = rsl[lm];
            const int y = a.labels[my_node];
            float z[12];
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
              const int c = kq + 4 * i;
              z[i] = (c < C) ? zt[c * 20 + lm] + rsn * a.pb[h + c] : -INFINITY;
              mx = fmaxf(mx, z[i]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float se = 0.f;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
              z[i] = (kq + 4 * i < C) ? expf(z[i] - mx) : 0.f;
              se += z[i];
            }
            se += __shfl_xor(se, 16, 64);
            se += __shfl_xor(se, 32, 64);
            const float inv = 1.f / se;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
              const int c = kq + 4 * i;
              if (c < Cp) zt[c * 20 + lm] = (c < C && node_live) ? (z[i] * inv - (c == y ? 1.f : 0.f)) * a.inv_n : 0.f;
            }
          }
          __syncthreads();
          if (tid < Cp * 4) {
            const int c = tid >> 2, q4 = (tid & 3) * 4;
            st16_sc1(a.GT, actT_bytes, ((long)c * npad + node0 + q4) * 4, *reinterpret_cast<const f32x4*>(&zt[c * 20 + q4]));
          }
        } else {
          // ---- R^T of this block is in zt: publish it, then dH1^T = (Wl R^T) * [H1 > 0] for this wave's hidden tiles ----
          if (tid < Cp * 4) {
            const int c = tid >> 2, q4 = (tid & 3) * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(&zt[c * 20 + q4]);
            if (node0 + q4 + 3 >= n) {
              if (node0 + q4 >= n) v.x = 0.f;
              if (node0 + q4 + 1 >= n) v.y = 0.f;
              if (node0 + q4 + 2 >= n) v.z = 0.f;
              if (node0 + q4 + 3 >= n) v.w = 0.f;
            }
            st16_sc1(a.RT, actT_bytes, ((long)c * npad + node0 + q4) * 4, v);
          }
          f32x4 rb[3];      // B': R^T[c = 16 s + 4 kq + t][node lm]
#pragma unroll
          for (int s2 = 0; s2 < 3; ++s2) {
            const int sc = min(s2, nct - 1);
            rb[s2].x = zt[(16 * sc + 4 * kq + 0) * 20 + lm]; rb[s2].y = zt[(16 * sc + 4 * kq + 1) * 20 + lm];
            rb[s2].z = zt[(16 * sc + 4 * kq + 2) * 20 + lm]; rb[s2].w = zt[(16 * sc + 4 * kq + 3) * 20 + lm];
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int ti = wave + GS_IP_WAVES * i;
            if (ti < nht) {
              const int hh0 = 16 * ti;
              f32x4 acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int s2 = 0; s2 < 3; ++s2)
                if (s2 < nct) GS_MFMA4(acc2, wlc[i][s2], rb[s2]);      // A': Wl[hh0 + lm][c] (rows c >= C of Wl^T are zero)
              const unsigned mb = (mbits >> (4 * i)) & 15u;
#pragma unroll
              for (int r = 0; r < 4; ++r) hbuf[(hh0 + 4 * kq + r) * 20 + lm] = (((mb >> r) & 1u) && node_live) ? acc2[r] : 0.f;
            }
          }
          __syncthreads();
          for (int idx = tid; idx < h * 4; idx += GS_IP_THREADS) {
            const int hh = idx >> 2, q4 = (idx & 3) * 4;
            st16_sc1(a.dH1T, hidT_bytes, ((long)hh * npad + node0 + q4) * 4, *reinterpret_cast<const f32x4*>(&hbuf[hh * 20 + q4]));
          }
        }
      }
      grid_sync(a, ++bar);
    }

    // ================= phase D: one 16 x 16 gradient tile per PAIR of waves (each half of K), Adam in the epilogue =================
    {
      // (bias task L sums row block L of dH1^T (gb1) and, for L < nct, row block L of G^T against rs (gbl): ogbn-arxiv has 128 + 48 + 16 =
      // 192 tasks = one round of the 4 x 48 wave pairs)
      const int nW1 = ndt * nht, nWl = nht * nct, nb1 = nht;
      const int ntask = nW1 + nWl + nb1;
      const int per_round = (GS_IP_WAVES / 2) * G;
      const int rounds = (ntask + per_round - 1) / per_round;
      const float ss = a.step_size[step], isb = a.inv_sqrt_bc2[step];
      GS_IP_LANES();
      (void)my_node; (void)node_live;
      const int pair = wave >> 1, half = wave & 1;
      const int kb_lo = half ? (nkb + 1) / 2 : 0, kb_hi = half ? nkb : (nkb + 1) / 2;
#pragma unroll 1
      for (int r0 = 0; r0 < rounds; ++r0) {
        const int task = (r0 * G + wg) * (GS_IP_WAVES / 2) + pair;
        const bool live = task < ntask;                    // wave-uniform
        const bool tile = task < nW1 + nWl, isW1 = task < nW1;
        int mt = 0, ntile = 0;
        const float *Arows = a.XT, *Brows = a.dH1T;
        if (isW1) { mt = task / nht; ntile = task - mt * nht; }
        else if (tile) { const int u = task - nW1; mt = u / nct; ntile = u - mt * nct; Arows = a.H1T; Brows = a.RT; }
        else if (live) { mt = 0; ntile = task - nW1 - nWl; Arows = a.dH1T; Brows = a.GT; }      // bias task: row block `ntile` of dH1^T and of G^T
        const float* __restrict__ ar = Arows + (long)(16 * (tile ? mt : ntile) + lm) * npad;      // A'[m][k = node]
        const bool has_bl = !tile && ntile < nct;
        const float* __restrict__ br = Brows + (long)(16 * (has_bl || tile ? ntile : 0) + lm) * npad;      // B'[k = node][n]  (bias tasks: row block of G^T)
        // this lane's 4 elements: rows 16 mt + 4 kq + r, column 16 ntile + lm
        const int col = 16 * ntile + lm, row0 = 16 * mt + 4 * kq;
        long idx[4];
        bool ok[4];
        float pm[4], pv[4], pp[4];
        if (live && tile && !half) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (isW1) { ok[r] = row0 + r < d; idx[r] = a.oW1 + (long)(row0 + r) * h + col; }
            else { ok[r] = col < C; idx[r] = a.oWl + (long)(row0 + r) * C + col; }
            if (ok[r]) { pm[r] = a.m[idx[r]]; pv[r] = a.v[idx[r]]; pp[r] = a.theta[idx[r]]; }
          }
        }
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        float bsum = 0.f, bsum2 = 0.f;
        if (live) {
#pragma unroll 1
          for (int kb0 = kb_lo; kb0 < kb_hi; kb0 += 8) {
            f32x4 av[8], bv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int kb = min(kb0 + q, kb_hi - 1);
              const int k = 16 * kb + 4 * kq;
              av[q] = *reinterpret_cast<const f32x4*>(ar + k);
              bv[q] = *reinterpret_cast<const f32x4*>(br + k);
              if (kb0 + q >= kb_hi) { av[q] = (f32x4){0.f, 0.f, 0.f, 0.f}; if (!tile) bv[q] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            }
            if (tile) {
#pragma unroll
              for (int q = 0; q < 8; ++q) GS_MFMA4(acc, av[q], bv[q]);
            } else {
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const int k = 16 * min(kb0 + q, kb_hi - 1) + 4 * kq;
                const f32x4 w = *reinterpret_cast<const f32x4*>(a.rs + k);
                bsum += (av[q].x + av[q].y) + (av[q].z + av[q].w);
                bsum2 += (bv[q].x * w.x + bv[q].y * w.y) + (bv[q].z * w.z + bv[q].w * w.w);
              }
            }
          }
        }
        if (!tile) {
          bsum += __shfl_xor(bsum, 16, 64);
          bsum += __shfl_xor(bsum, 32, 64);
          bsum2 += __shfl_xor(bsum2, 16, 64);
          bsum2 += __shfl_xor(bsum2, 32, 64);
          acc.x = bsum; acc.y = has_bl ? bsum2 : 0.f;
        }
        if (half) red[pair * 64 + lane] = acc;       // the second half of K hands its partial tile to its partner
        __syncthreads();
        if (live && !half) {
          acc += red[pair * 64 + lane];
          if (tile) {
            f32x4 nw;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float th = 0.f;
              if (ok[r]) {
                const float g = acc[r];
                const float mi = pm[r] + (1.f - a.b1) * (g - pm[r]);
                const float vi = a.b2 * pv[r] + (1.f - a.b2) * g * g;
                const float denom = sqrtf(vi) * isb + a.eps;
                th = pp[r] - ss * (mi / denom);
                a.theta[idx[r]] = th; a.m[idx[r]] = mi; a.v[idx[r]] = vi;
                if (step + 1 == a.steps) a.g[idx[r]] = g;
              }
              nw[r] = th;
            }
            // the transposed operand of the next step: [col][row0 .. row0 + 3]
            if (isW1) st16_sc1(a.W1T, (long)h * dpad * 4, ((long)col * dpad + row0) * 4, nw);
            else st16_sc1(a.WlT, (long)Cp * h * 4, ((long)col * h + row0) * 4, nw);
          } else {
            // bias gradients: gb1[hh] = sum_node dH1^T[hh][node];  gbl[c] = sum_node rs[node] G^T[c][node]
            const int u = 16 * ntile + lm;
#pragma unroll
            for (int which = 0; which < 2; ++which) {
              const bool isb1 = which == 0;
              if (kq == 0 && (isb1 ? u < h : (has_bl && u < C))) {
                const float sg = isb1 ? acc.x : acc.y;
                const long i = (isb1 ? a.ob1 : a.obl) + u;
                const float qm = a.m[i], qv = a.v[i], qp = a.theta[i];
                const float mi = qm + (1.f - a.b1) * (sg - qm);
                const float vi = a.b2 * qv + (1.f - a.b2) * sg * sg;
                const float th = qp - ss * (mi / (sqrtf(vi) * isb + a.eps));
                a.theta[i] = th; a.m[i] = mi; a.v[i] = vi;
                if (step + 1 == a.steps) a.g[i] = sg;
                st4_sc1(a.pb + (isb1 ? u : h + u), th);
              }
            }
          }
        }
        __syncthreads();
      }
    }
    if (step + 1 < a.steps) grid_sync(a, ++bar);
  }
  if (a.trace && tid == 0) a.trace[(long)wg * 128 + 4 * bar + 1] = wall_clock64();
}

static_assert(256 * 20 * 4 + (GS_IP_THREADS / 64) * 3 * 64 * 16 >= 64 * 65 * 4, "phase 0's transpose tile aliases hbuf + red");

// ---- host side ---------------------------------------------------------------------------------------------------------------
// Workspace (floats) the launch needs behind the engine's own buffers; layout fixed here so that gs_engine.hip only carves one block.
extern "C" long gs_inner_persist_ws_floats(int n, int d, int h, int C) {
  const long npad = (n + 15) / 16 * 16, dpad = (d + 15) / 16 * 16, Cp = (C + 15) / 16 * 16;
  // XT | W1T | WlT | pb | TT GT RT | H1T dH1T | rs | Q QT | sync (64 words)
  return dpad * npad + h * dpad + Cp * h + (h + Cp) + 3 * Cp * npad + 2L * h * npad + npad + 2 * (long)n * npad + 64 + 64;
}

// eligibility: SGC with two transforms on the fp32 chain's shapes (the caller checks model / BatchNorm / dropout / structure)
extern "C" int gs_inner_persist_eligible(int n, int d, int h, int C, int steps) {
  return n >= 1 && d >= 4 && h >= 16 && h <= 256 && h % 16 == 0 && C >= 1 && C <= 48 && steps >= 1 && steps <= GS_IP_MAX_STEPS &&
         (n + 15) / 16 <= 96;
}

static unsigned long long* g_inner_trace = nullptr;      // device buffer [G][128] u64 (measurement: tools/trace_inner_persist.py)
extern "C" int gs_inner_persist_trace(void* buf) {
  g_inner_trace = (unsigned long long*)buf;
  return GS_OK;
}
static int g_inner_grid = 48;      // workgroups of the persistent launch (gs_inner_persist_set_grid): what the hoisted forward's 13/16 grid leaves free
extern "C" int gs_inner_persist_set_grid(int g) {
  g_inner_grid = g < 1 ? 1 : (g > 256 ? 256 : g);
  return GS_OK;
}

// Q / QT point INTO ws (the caller's products write them there with row stride npad); everything else is carved here.
struct gs_inner_ws {
  float *XT, *W1T, *WlT, *pb, *TT, *GT, *RT, *H1T, *dH1T, *rs, *Q, *QT;
  unsigned* sync;
  int npad, dpad, Cp;
};
extern "C" void gs_inner_persist_carve(float* ws, int n, int d, int h, int C, gs_inner_ws* o) {
  const long npad = (n + 15) / 16 * 16, dpad = (d + 15) / 16 * 16, Cp = (C + 15) / 16 * 16;
  o->npad = (int)npad; o->dpad = (int)dpad; o->Cp = (int)Cp;
  float* p = ws;
  o->sync = reinterpret_cast<unsigned*>(p); p += 64;      // (first: the block the create-time memset zeroes starts at the allocation's start)
  o->XT = p; p += dpad * npad;
  o->W1T = p; p += h * dpad;
  o->WlT = p; p += Cp * h;
  o->pb = p; p += ((h + Cp + 15) / 16) * 16;
  o->TT = p; p += Cp * npad; o->GT = p; p += Cp * npad; o->RT = p; p += Cp * npad;
  o->H1T = p; p += (long)h * npad; o->dH1T = p; p += (long)h * npad;
  o->rs = p; p += npad;
  o->Q = p; p += (long)n * npad; o->QT = p; p += (long)n * npad;
}

int gs_inner_persist_launch(const gs_inner_ws& w, int n, int d, int h, int C, int steps, const float* X, long ldx, const int* labels,
                            float* theta, float* m, float* v, float* g, long oW1, long ob1, long oWl, long obl, const float* step_size,
                            const float* inv_sqrt_bc2, float b1, float b2, float eps, unsigned* counter_host, hipStream_t s) {
  gs_inner_args a;
  memset(&a, 0, sizeof(a));
  a.n = n; a.npad = w.npad; a.d = d; a.dpad = w.dpad; a.h = h; a.C = C; a.Cp = w.Cp; a.steps = steps;
  const int nrb = w.npad / 16;
  a.G = g_inner_grid > nrb ? g_inner_grid : nrb;
  a.X = X; a.ldx = ldx; a.Q = w.Q; a.QT = w.QT; a.labels = labels;
  a.theta = theta; a.m = m; a.v = v; a.g = g; a.oW1 = oW1; a.ob1 = ob1; a.oWl = oWl; a.obl = obl;
  a.XT = w.XT; a.W1T = w.W1T; a.WlT = w.WlT; a.pb = w.pb; a.TT = w.TT; a.GT = w.GT; a.RT = w.RT; a.H1T = w.H1T; a.dH1T = w.dH1T; a.rs = w.rs;
  a.sync = w.sync; a.base = *counter_host; a.trace = g_inner_trace;
  for (int i = 0; i < steps; ++i) { a.step_size[i] = step_size[i]; a.inv_sqrt_bc2[i] = inv_sqrt_bc2[i]; }
  a.b1 = b1; a.b2 = b2; a.eps = eps; a.inv_n = 1.f / (float)n;
  const unsigned nbar = 4u * (unsigned)steps;      // phase 0 + four per step - the one behind the last phase D
  *counter_host += nbar * (unsigned)a.G;
  hipLaunchKernelGGL(gs_inner_persist_kernel, dim3(a.G), dim3(GS_IP_THREADS), 0, s, a);
  GS_LAUNCH_CHECK();
  return GS_OK;
}
