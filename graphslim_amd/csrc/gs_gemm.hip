// Batched / grouped fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// result = k-ordered fmaf chain, so parity with an fp32 CPU reference is reassociation-level only).
//
//   C[b] (M x N) = alpha * A[b] (M x K) * B[b] (K x N)  [+ bias[n]] [relu] [* (mask>0)]  + beta * C[b]
//
// A and B are addressed through element strides (so any transpose is a stride choice), with
// optional row/col gather indices (A(m,k) -> A[idx_m[m]*sAm + idx_k[k]*sAk]) for the sampled-edge GEMM,
// two batch dimensions (b = b1*nb2 + b2) with uniform strides, optional per-b1 element offsets and
// per-b1 M/K extents read from device memory (grouped GEMM over classes of different size), and an
// optional split of K over b2 (partial products go to slabs; gs_reduce_slabs_f32 sums them in a fixed
// order -> bitwise reproducible, unlike float atomics).
//
// Tiling: 256 threads = 4 waves (64 lanes), block tile BM x BN, wave tile WM x WN built from 32x32
// MFMA tiles; K step 16; operands staged global -> registers -> LDS [k][m] (pad 4) so that the
// per-lane MFMA operand read (lane = row, k = lane>>5) is a conflict-free ds_read_b32; the next
// tile's global loads are issued before the current tile's MFMAs (register double buffer, one
// barrier pair per K step).
#include <type_traits>

#include "gs_common.h"
#include "gs_small.h"      // f32x16, f32x4u and the latency-oriented small-product tile

#define GS_PAD 4   // pad of an m-contiguous LDS image [k][BM + 4]; a k-contiguous image uses rows of BK + 4 floats
                   // (16-B aligned rows; 20 or 36 dwords per row keep the 16 lanes of a ds_read_b128 group on 16 distinct
                   // 4-bank slots -> conflict-free)

// operand access modes (chosen on the host from the strides)
#define GS_MODE_K 0   // k contiguous in memory  -> global float4 along k, LDS [m][20], fragment = one ds_read_b128
#define GS_MODE_M 1   // m contiguous in memory  -> global float4 along m, LDS [k][BM+4], fragment = 4 ds_read_b32
#define GS_MODE_G 2   // anything else (scalar, clamped, branch-free)

// Within a K tile of BK (16 or 32) the MFMA steps consume k in the order k = 8q + 4*(lane>>5) + t  (q < BK/8; t = 0..3): both
// operands use the same permutation, so a k-contiguous operand gets its 4 values for t = 0..3 with one 16-byte read.

template <int BMN, int MODE, int BK>
struct TileLds {
  static constexpr int SIZE = (MODE == GS_MODE_K) ? BMN * (BK + 4) : BK * (BMN + GS_PAD);
};

// Loads one operand tile (BMN rows of the operand's "outer" dimension x 16 k) into registers.
//   outer index o in [0, BMN) (+ o0), k in [k0, k0+16);  element(o, k) = base[row(o)*s_o + kidx(k)*s_k]
// interior == true (block-uniform): every element in range and 16-byte vector loads are legal -> no selects.
template <int BMN, int MODE, int BK>
struct TileLoader {
  static constexpr int KS = BK + 4, KQ = BK / 4;
  static constexpr int UNITS = (MODE == GS_MODE_G) ? BMN * BK : BMN * BK / 4;   // float4 units or scalars
  static constexpr int PER = (UNITS + 255) / 256;
  float4 r4[(MODE == GS_MODE_G) ? 1 : PER];
  float r1[(MODE == GS_MODE_G) ? PER : 1];

  __device__ __forceinline__ void load(const float* __restrict__ base, long s_o, long s_k, const int* __restrict__ g_o,
                                       const int* __restrict__ g_k, int o0, int n_o, int k0, int k_hi, bool interior, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int u = tid + i * 256;
      if (UNITS % 256 != 0 && u >= UNITS) break;
      if (MODE == GS_MODE_K) {
        const int o = u / KQ, kq = (u % KQ) * 4;
        const int go = min(o0 + o, n_o - 1);
        const long ro = (g_o ? (long)g_o[go] : (long)go) * s_o;
        if (interior) {
          { const f32x4u t = *reinterpret_cast<const f32x4u*>(base + ro + (k0 + kq)); r4[i] = make_float4(t.x, t.y, t.z, t.w); }
        } else {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = k0 + kq + j;
            const float x = base[ro + min(k, k_hi - 1)];
            v[j] = (o0 + o < n_o && k < k_hi) ? x : 0.f;
          }
          r4[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
      } else if (MODE == GS_MODE_M) {
        constexpr int Q = BMN / 4;
        const int k = u / Q, o4 = (u % Q) * 4;
        const int gk = min(k0 + k, k_hi - 1);
        const long rk = (g_k ? (long)g_k[gk] : (long)gk) * s_k;
        if (interior) {
          { const f32x4u t = *reinterpret_cast<const f32x4u*>(base + rk + (o0 + o4)); r4[i] = make_float4(t.x, t.y, t.z, t.w); }
        } else {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int o = o0 + o4 + j;
            const float x = base[rk + min(o, n_o - 1)];
            v[j] = (o < n_o && k0 + k < k_hi) ? x : 0.f;
          }
          r4[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
      } else {
        const int o = u % BMN, k = u / BMN;
        const int go = min(o0 + o, n_o - 1), gk = min(k0 + k, k_hi - 1);
        const long a = (g_o ? (long)g_o[go] : (long)go) * s_o + (g_k ? (long)g_k[gk] : (long)gk) * s_k;
        const float x = base[a];
        r1[i] = (o0 + o < n_o && k0 + k < k_hi) ? x : 0.f;
      }
    }
  }

  __device__ __forceinline__ void store(float* __restrict__ lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int u = tid + i * 256;
      if (UNITS % 256 != 0 && u >= UNITS) break;
      if (MODE == GS_MODE_K) {
        *reinterpret_cast<float4*>(lds + (u / KQ) * KS + (u % KQ) * 4) = r4[i];
      } else if (MODE == GS_MODE_M) {
        constexpr int Q = BMN / 4;
        *reinterpret_cast<float4*>(lds + (u / Q) * (BMN + GS_PAD) + (u % Q) * 4) = r4[i];
      } else {
        lds[(u / BMN) * (BMN + GS_PAD) + (u % BMN)] = r1[i];
      }
    }
  }
};

// fragment of the 4 MFMA steps t = 0..3 of half-tile q for operand row `r`
template <int BMN, int MODE, int BK>
__device__ __forceinline__ void read_frag(const float* __restrict__ lds, int r, int q, int kh, float (&f)[4]) {
  if (MODE == GS_MODE_K) {
    const float4 v = *reinterpret_cast<const float4*>(lds + r * (BK + 4) + q * 8 + kh * 4);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) f[t] = lds[(q * 8 + kh * 4 + t) * (BMN + GS_PAD) + r];
  }
}

// Generated B operand of the real-side edge GEMM (never materialised):
//   B(k, n) = w[k] * [mask[idx[k]][n] > 0] * Y[k / div][n]          (n contiguous: the LDS image of GS_MODE_M)
// = d loss / d pre-activation of sampled edge k (gcond_base.py:223 autograd through relu and the hop-2 block).
template <int BN, int BK>
struct BGenLoader {
  static constexpr int Q = BN / 4, UNITS = BK * Q, PER = (UNITS + 255) / 256;
  float4 r4[PER];
  __device__ __forceinline__ void load(const float* __restrict__ Y, long ldy, const float* __restrict__ mask, long ldm,
                                       const int* __restrict__ idx, const float* __restrict__ w, int div, int n0, int N, int k0,
                                       int k_hi, bool interior, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int u = tid + i * 256;
      if (UNITS % 256 != 0 && u >= UNITS) break;
      const int k = u / Q, o4 = (u % Q) * 4;
      const int gk = min(k0 + k, k_hi - 1);
      const long yr = (long)(gk / div) * ldy, mr = (long)idx[gk] * ldm;
      const float wk = (k0 + k < k_hi) ? w[gk] : 0.f;
      if (interior) {
        const f32x4u y = *reinterpret_cast<const f32x4u*>(Y + yr + n0 + o4);
        const f32x4u m = *reinterpret_cast<const f32x4u*>(mask + mr + n0 + o4);
        r4[i] = make_float4(m.x > 0.f ? wk * y.x : 0.f, m.y > 0.f ? wk * y.y : 0.f, m.z > 0.f ? wk * y.z : 0.f,
                            m.w > 0.f ? wk * y.w : 0.f);
      } else {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int o = min(n0 + o4 + j, N - 1);
          const float y = Y[yr + o], m = mask[mr + o];
          v[j] = (n0 + o4 + j < N && m > 0.f) ? wk * y : 0.f;
        }
        r4[i] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int u = tid + i * 256;
      if (UNITS % 256 != 0 && u >= UNITS) break;
      *reinterpret_cast<float4*>(lds + (u / Q) * (BN + GS_PAD) + (u % Q) * 4) = r4[i];
    }
  }
};

// Occupancy bound: left alone the 128x128 tile compiles to 188 VGPR + 64 AGPR (2 waves/SIMD); asking for 3 brings it to 168
// registers with a handful of scalar spills and measures 10-16 % faster on the PGE / H1 shapes (tools/ab_occ.py, r01);
// 4 waves/SIMD spills ~100 registers and is slower again.
constexpr int gs_min_waves(int bm, int bn) { return bm * bn >= 128 * 128 ? 3 : 1; }

template <int BM, int BN, int WM, int WN, int AMODE, int BMODE, int BK, bool BGEN = false>
__global__ __launch_bounds__(256, gs_min_waves(BM, BN)) void gs_gemm_f32_kernel(gs_gemm_desc p) {
  constexpr int GS_BK = BK;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
  constexpr int ASZ = TileLds<BM, AMODE, BK>::SIZE, BSZ = TileLds<BN, BMODE, BK>::SIZE;
  __shared__ __attribute__((aligned(16))) float As[ASZ];
  __shared__ __attribute__((aligned(16))) float Bs[BSZ];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, b1 = b / p.nb2, b2 = b - b1 * p.nb2;
  int M = p.M, K = p.K;
  if (p.Mb) M = p.Mb[b1];
  if (p.Kb) K = p.Kb[b1];
  int k_lo = 0, k_hi = K;
  if (p.kchunk > 0) {
    k_lo = b2 * p.kchunk;
    k_hi = min(K, k_lo + p.kchunk);
  }
  // Tile of this workgroup.  The grid is flat in x (no 65 535 limit on the M-tile count: n'^2 / BM row tiles of the PGE
  // products exceed it from n' ~ 2 900) and XCD-aware: workgroups are dealt round-robin over the 8 XCDs (ids b and b + 8
  // share one), so XCD x gets the CONTIGUOUS chunk x of the tile sequence (n fastest) — the N tiles that share an A row
  // block run on one XCD back to back and the second one finds the block in that XCD's L2 instead of pulling it through
  // the fabric again (r01: the PGE layer-2 product fetched A twice).  Bijective for any tile count.
  const int tn = (p.N + BN - 1) / BN;
  int tile;
  {
    const int T = gridDim.x, L = blockIdx.x, xcd = L & 7, slot = L >> 3, q = T >> 3, r = T & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int mt = tile / tn;
  const int m0 = mt * BM, n0 = (tile - mt * tn) * BN;
  if (m0 >= M) return;

  const float* __restrict__ A = p.A + (long)b1 * p.sAb1 + (p.kchunk > 0 ? 0 : (long)b2 * p.sAb2) + (p.offA ? p.offA[b1] : 0);
  const float* __restrict__ B = p.B + (long)b1 * p.sBb1 + (p.kchunk > 0 ? 0 : (long)b2 * p.sBb2) + (p.offB ? p.offB[b1] : 0);
  float* __restrict__ C = p.C + (long)b1 * p.sCb1 + (long)b2 * p.sCb2 + (p.offC ? p.offC[b1] : 0);
  const int* gAm = p.gatherA_m ? p.gatherA_m + (long)b1 * p.sGb1 : nullptr;
  const int* gAk = p.gatherA_k ? p.gatherA_k + (long)b1 * p.sGb1 : nullptr;
  const int* gBk = p.gatherB_k ? p.gatherB_k + (long)b1 * p.sGb1 : nullptr;

  const bool vecA = true, vecB = true;   // 4-byte alignment suffices for the global vector loads (f32x4u)
  const bool fullM = (m0 + BM <= M), fullN = (n0 + BN <= p.N);

  TileLoader<BM, AMODE, BK> la;
  typename std::conditional<BGEN, BGenLoader<BN, BK>, TileLoader<BN, BMODE, BK>>::type lb;
  const int* bg_idx = BGEN ? p.bgen_idx + (long)b1 * p.sGb1 : nullptr;
  const float* bg_w = BGEN ? p.bgen_w + (long)b1 * p.sGb1 : nullptr;
  auto load_tiles = [&](int k0) {
    const bool fullK = (k0 + GS_BK <= k_hi);
    la.load(A, p.sAm, p.sAk, gAm, gAk, m0, M, k0, k_hi, vecA && fullM && fullK, tid);
    if constexpr (BGEN)
      lb.load(B, p.sBk, p.bgen_mask, p.bgen_ldm, bg_idx, bg_w, p.bgen_div, n0, p.N, k0, k_hi, fullN && fullK, tid);
    else
      lb.load(B, p.sBn, p.sBk, nullptr, gBk, n0, p.N, k0, k_hi, vecB && fullN && fullK, tid);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  const int li = lane & 31, kh = lane >> 5;

  auto mfma_tile = [&](const float* __restrict__ as, const float* __restrict__ bs) {
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      float a[TM][4], bb[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) read_frag<BM, AMODE, BK>(as, wm0 + i * 32 + li, q, kh, a[i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) read_frag<BN, BMODE, BK>(bs, wn0 + j * 32 + li, q, kh, bb[j]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], bb[j][t], acc[i][j], 0, 0, 0);
    }
  };
  if (k_lo < k_hi) {
    load_tiles(k_lo);
    // (two LDS images with one barrier per K step were tried: no gain, 293 vs 293 us — the barriers are not the limiter)
    for (int k0 = k_lo; k0 < k_hi; k0 += GS_BK) {
      la.store(As, tid);
      lb.store(Bs, tid);
      __syncthreads();
#ifndef GS_TRIAGE_NO_GLOAD                             // (tools/micro/gemm_triage.hip: phase-removal experiments)
      if (k0 + GS_BK < k_hi) load_tiles(k0 + GS_BK);   // in flight under the MFMAs below
#endif
      mfma_tile(As, Bs);
      __syncthreads();
    }
  }

  // epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  const bool has_bias = p.bias != nullptr, has_mask = p.mask != nullptr, has_beta = p.beta != 0.f;
  const bool has_stat = p.colstat != nullptr;
  float cs1[TN], cs2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) cs1[j] = cs2[j] = 0.f;
  // Fast path (block-uniform): the tile lies inside C and every element offset fits 32 bits, so each of the 16 values of a
  // 32x32 tile costs one fma, one max, one 32-bit offset add and the store — the generic path below spends ~8 VALU
  // instructions of 64-bit index arithmetic per element, which at K = 128..256 is a quarter of the kernel's issue slots.
  const bool fit32 = (long)M * p.ldc < (1L << 29) && (!has_mask || (long)M * p.ldmask < (1L << 29));
#ifdef GS_TRIAGE_NO_STORE
  if (acc[0][0][0] != 12345.678f) return;
#endif
  if (fullM && fullN && fit32) {
    const unsigned ldc = (unsigned)p.ldc, ldm = (unsigned)p.ldmask;
    // (the bias values are fetched BEFORE the first store: a load between the stores of two tiles would wait — vmcnt retires
    // in order — for every store issued before it, i.e. a write round trip per tile)
    float bvj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bvj[j] = has_bias ? p.bias[(long)b1 * p.sbias_b1 + n0 + wn0 + j * 32 + li] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned col = n0 + wn0 + j * 32 + li, row0 = m0 + wm0 + i * 32 + 4 * kh;
        const float bv = bvj[j];
        const unsigned off = row0 * ldc + col;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaf(p.alpha, acc[i][j][r], bv);
        if (p.relu) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (has_mask) {
          const unsigned moff = row0 * ldm + col;
          float mk[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) mk[r] = p.mask[moff + ((r & 3) + 8 * (r >> 2)) * ldm];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = (mk[r] > 0.f) ? v[r] : 0.f;
        }
        if (has_beta) {
          float cv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) cv[r] = C[off + ((r & 3) + 8 * (r >> 2)) * ldc];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = fmaf(p.beta, cv[r], v[r]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) C[off + ((r & 3) + 8 * (r >> 2)) * ldc] = v[r];
        if (has_stat) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            cs1[j] += v[r];
            cs2[j] = fmaf(v[r], v[r], cs2[j]);
          }
        }
      }
  } else {
    // Edge tiles / huge outputs.  Optional inputs (bias / mask / old C) are fetched 16 at a time behind block-uniform
    // branches with clamped addresses, so no load sits behind a per-element branch.
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn0 + j * 32 + li;
        const int colc = min(col, p.N - 1);
        const int rbase = m0 + wm0 + i * 32 + 4 * kh;
        float bv = 0.f;
        if (has_bias) bv = p.bias[(long)b1 * p.sbias_b1 + colc];
        float mk[16], cv[16];
        if (has_mask) {
#pragma unroll
          for (int r = 0; r < 16; ++r) mk[r] = p.mask[(long)min(rbase + (r & 3) + 8 * (r >> 2), M - 1) * p.ldmask + colc];
        }
        if (has_beta) {
#pragma unroll
          for (int r = 0; r < 16; ++r) cv[r] = C[(long)min(rbase + (r & 3) + 8 * (r >> 2), M - 1) * p.ldc + colc];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          float v = fmaf(p.alpha, acc[i][j][r], bv);
          if (p.relu) v = fmaxf(v, 0.f);
          if (has_mask) v = (mk[r] > 0.f) ? v : 0.f;
          if (has_beta) v = fmaf(p.beta, cv[r], v);
          if (row < M && col < p.N) {
            C[(long)row * p.ldc + col] = v;
            cs1[j] += v;
            cs2[j] = fmaf(v, v, cs2[j]);
          }
        }
      }
  }
  if (has_stat) {
    // per-block column sums of the written tile: colstat[(m_tile*2 + q)*N + col], q = 0: sum, 1: sum of squares
    // (BatchNorm statistics without a second pass over C).  Reduce: lane halves -> waves along M (through LDS) -> global.
    constexpr int WAVES_M = BM / WM;
    __syncthreads();                       // the operand tiles are dead: reuse As as scratch
    float* red = As;                       // [2][WAVES_M][BN]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float a = cs1[j] + __shfl_xor(cs1[j], 32, 64), b = cs2[j] + __shfl_xor(cs2[j], 32, 64);
      if (kh == 0) {
        const int c = wn0 + j * 32 + li;
        red[(0 * WAVES_M + wave / WAVES_N) * BN + c] = a;
        red[(1 * WAVES_M + wave / WAVES_N) * BN + c] = b;
      }
    }
    __syncthreads();
    for (int c = tid; c < 2 * BN; c += 256) {
      const int q = c / BN, cc = c % BN;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES_M; ++w) t += red[(q * WAVES_M + w) * BN + cc];
      if (n0 + cc < p.N) p.colstat[((long)mt * 2 + q) * p.N + n0 + cc] = t;
    }
  }
}

template <bool AVEC, bool BVEC>
__global__ __launch_bounds__(256) void gs_gemm_small_kernel(gs_gemm_desc p) {
  __shared__ float red[4][16][64];
  gs_small_tile<AVEC, BVEC>(p, blockIdx.x, blockIdx.y, blockIdx.z, red);
}

// Up to GS_GROUP_MAX independent latency-bound products in ONE launch (descriptors travel by value in the kernel arguments):
// the synthetic side and the inner steps are chains of ~6 us launches, and products that do not depend on each other
// (gW2, gb2 and dS of an inner step; U and V of the PGE; ...) need not pay that latency one after the other.
// blockIdx.z enumerates (group, batch); x / y cover the largest tile grid of the group, the others exit at once.
#define GS_GROUP_MAX 4
struct gs_gemm_group_arg {
  gs_gemm_desc d[GS_GROUP_MAX];
  int z0[GS_GROUP_MAX + 1];   // first blockIdx.z of each group
  int n;
};
// the same with an optional SECOND product per member that accumulates into the member's output (beta = 1, same M / N / batching /
// output addressing): C = A1 B1 ; C += A2 B2 by the workgroup that owns the tile — the thread that stored an element of the first
// product is the one that reads it back for the second, so the result is bitwise that of two dependent launches
struct gs_gemm_group2_arg {
  gs_gemm_desc d[GS_GROUP_MAX];
  gs_gemm_desc d2[GS_GROUP_MAX];
  int z0[GS_GROUP_MAX + 1];
  int has2[GS_GROUP_MAX];
  int n;
};

__device__ __forceinline__ void gs_small_any(const gs_gemm_desc& p, int b, float (&red)[4][16][64]) {
  const bool av = p.sAk == 1, bv = p.sBk == 1;   // block-uniform
  if (av && bv) gs_small_tile<true, true>(p, blockIdx.x, blockIdx.y, b, red);
  else if (av) gs_small_tile<true, false>(p, blockIdx.x, blockIdx.y, b, red);
  else if (bv) gs_small_tile<false, true>(p, blockIdx.x, blockIdx.y, b, red);
  else gs_small_tile<false, false>(p, blockIdx.x, blockIdx.y, b, red);
}

__global__ __launch_bounds__(256) void gs_gemm_small_group_kernel(gs_gemm_group_arg g) {
  __shared__ float red[4][16][64];
  const int z = blockIdx.z;
  if (z < g.z0[1]) gs_small_any(g.d[0], z, red);
  else if (z < g.z0[2]) gs_small_any(g.d[1], z - g.z0[1], red);
  else if (z < g.z0[3]) gs_small_any(g.d[2], z - g.z0[2], red);
  else gs_small_any(g.d[3], z - g.z0[3], red);
}

__global__ __launch_bounds__(256) void gs_gemm_small_group2_kernel(gs_gemm_group2_arg g) {
  __shared__ float red[4][16][64];
  const int z = blockIdx.z;
  int i = 0;
  if (z >= g.z0[3]) i = 3;
  else if (z >= g.z0[2]) i = 2;
  else if (z >= g.z0[1]) i = 1;
  gs_small_any(g.d[i], z - g.z0[i], red);
  if (g.has2[i]) {
    __syncthreads();      // the partial-sum scratch is reused (and this thread's own stores of the first product are ordered before its loads)
    gs_small_any(g.d2[i], z - g.z0[i], red);
  }
}

// gs_gemm_small_group2_kernel for members that are batched over classes of very different row counts (Mb): the launch enumerates
// the M tiles that exist — tile_b[t] = batch, tile_my[t] = its M tile, built once on the host from the class sizes — instead of
// max-size tiles for every class (40 classes x 4 tiles where 48 exist: three quarters of the workgroups used to exit at once).
// grid (widest member's N tiles, tiles, members).  Same values as the rectangular launch.
__global__ __launch_bounds__(256) void gs_gemm_small_group2_tiles_kernel(gs_gemm_group2_arg g, const int* __restrict__ tile_b,
                                                                         const int* __restrict__ tile_my, int b_first) {
  __shared__ float red[4][16][64];
  const int i = blockIdx.z;
  const gs_gemm_desc& p = g.d[i];
  if ((int)blockIdx.x * 32 >= p.N) return;
  const int b = tile_b[blockIdx.y] - b_first, by = tile_my[blockIdx.y];
  {
    const bool av = p.sAk == 1, bv = p.sBk == 1;
    if (av && bv) gs_small_tile<true, true>(p, blockIdx.x, by, b, red);
    else if (av) gs_small_tile<true, false>(p, blockIdx.x, by, b, red);
    else if (bv) gs_small_tile<false, true>(p, blockIdx.x, by, b, red);
    else gs_small_tile<false, false>(p, blockIdx.x, by, b, red);
  }
  if (g.has2[i]) {
    __syncthreads();
    const gs_gemm_desc& q = g.d2[i];
    const bool av = q.sAk == 1, bv = q.sBk == 1;
    if (av && bv) gs_small_tile<true, true>(q, blockIdx.x, by, b, red);
    else if (av) gs_small_tile<true, false>(q, blockIdx.x, by, b, red);
    else if (bv) gs_small_tile<false, true>(q, blockIdx.x, by, b, red);
    else gs_small_tile<false, false>(q, blockIdx.x, by, b, red);
  }
}

bool gs_edge_eligible(const gs_gemm_desc& p);             // gs_edge.hip
int gs_edge_launch(const gs_gemm_desc& d, hipStream_t s);
bool gs_panel_eligible(const gs_gemm_desc& p);            // gs_panel.hip
int gs_panel_launch(const gs_gemm_desc& d, hipStream_t s);
int gs_panel_tiles(int M);

static long g_small_tiles = 8192;      // route to the small kernel when the 32x32 tile count is at most this ...
static double g_small_gflop = 1.5;     // ... and the product is at most this many GFLOP

extern "C" int gs_gemm_set_small(long max_tiles, double max_gflop) {
  g_small_tiles = max_tiles;
  g_small_gflop = max_gflop;
  return GS_OK;
}

static bool small_route(const gs_gemm_desc& p, int nbatch) {
  if (p.bgen_mask || p.colstat || p.gatherA_k || p.gatherB_k) return false;
  const long t32 = (long)gs_cdiv(p.M, 32) * gs_cdiv(p.N, 32) * nbatch;
  if (t32 > g_small_tiles) return false;
  const double keff = p.kchunk > 0 ? (double)p.kchunk : (double)p.K;
  return 2.0 * p.M * p.N * keff * nbatch <= g_small_gflop * 1e9;
}

static int launch_small(const gs_gemm_desc& p, int nbatch, hipStream_t s) {
  dim3 grid(gs_cdiv(p.N, 32), gs_cdiv(p.M, 32), nbatch);
  const bool av = p.sAk == 1, bv = p.sBk == 1;
  if (av && bv) hipLaunchKernelGGL((gs_gemm_small_kernel<true, true>), grid, dim3(256), 0, s, p);
  else if (av) hipLaunchKernelGGL((gs_gemm_small_kernel<true, false>), grid, dim3(256), 0, s, p);
  else if (bv) hipLaunchKernelGGL((gs_gemm_small_kernel<false, true>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((gs_gemm_small_kernel<false, false>), grid, dim3(256), 0, s, p);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

__global__ void gs_reduce_slabs_kernel(float* __restrict__ out, const float* __restrict__ slabs, long n, int ns,
                                       long slab_stride, float beta) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < ns; ++k) s += slabs[(long)k * slab_stride + i];   // fixed order: reproducible
  out[i] = (beta != 0.f ? beta * out[i] : 0.f) + s;
}

// 16-byte form: four adjacent elements per thread, the slab loop unrolled so that 8 loads are in flight (the scalar form
// above issued one dependent-looking 4-byte load per slab: 33 us for 40 slabs of 58 112 elements).  Same summation order.
__global__ void gs_reduce_slabs_v4_kernel(float* __restrict__ out, const float* __restrict__ slabs, long n4, int ns,
                                          long slab_stride4, float beta) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4* sl = reinterpret_cast<const float4*>(slabs) + i;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int k = 0;
  for (; k + 8 <= ns; k += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = sl[(long)(k + u) * slab_stride4];
#pragma unroll
    for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  for (; k < ns; ++k) {
    const float4 v = sl[(long)k * slab_stride4];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float4* o = reinterpret_cast<float4*>(out) + i;
  if (beta != 0.f) {
    const float4 c = *o;
    s.x += beta * c.x; s.y += beta * c.y; s.z += beta * c.z; s.w += beta * c.w;
  }
  *o = s;
}

extern "C" int gs_reduce_slabs_f32(float* out, const float* slabs, long n, int n_slabs, long slab_stride, float beta, gs_stream_t stream);
// two independent slab reductions (beta = 0) in ONE launch: blockIdx.y selects the reduction; same summation order as
// gs_reduce_slabs_v4_kernel (bitwise the two-launch values)
__global__ void gs_reduce_slabs_pair_kernel(float* __restrict__ out0, const float* __restrict__ slabs0, long n4_0, int ns0, long stride4_0,
                                            float* __restrict__ out1, const float* __restrict__ slabs1, long n4_1, int ns1, long stride4_1) {
  const bool second = blockIdx.y != 0;
  const long n4 = second ? n4_1 : n4_0, stride4 = second ? stride4_1 : stride4_0;
  const int ns = second ? ns1 : ns0;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4* sl = reinterpret_cast<const float4*>(second ? slabs1 : slabs0) + i;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int k = 0;
  for (; k + 8 <= ns; k += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = sl[(long)(k + u) * stride4];
#pragma unroll
    for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  for (; k < ns; ++k) {
    const float4 v = sl[(long)k * stride4];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  reinterpret_cast<float4*>(second ? out1 : out0)[i] = s;
}

// engine-internal: out0 = sum of ns0 slabs, out1 = sum of ns1 slabs (fixed order); one launch when both are 16-byte shaped
int gs_reduce_slabs_pair(float* out0, const float* slabs0, long n0, int ns0, long stride0, float* out1, const float* slabs1, long n1, int ns1,
                         long stride1, gs_stream_t stream) {
  const bool v4 = ((n0 | n1 | stride0 | stride1) & 3) == 0 && ((((uintptr_t)out0) | ((uintptr_t)slabs0) | ((uintptr_t)out1) | ((uintptr_t)slabs1)) & 15) == 0;
  if (!v4 || n0 <= 0 || n1 <= 0 || ns0 < 1 || ns1 < 1) {
    GS_TRY(gs_reduce_slabs_f32(out0, slabs0, n0, ns0, stride0, 0.f, stream));
    return gs_reduce_slabs_f32(out1, slabs1, n1, ns1, stride1, 0.f, stream);
  }
  const long m = n0 > n1 ? n0 : n1;
  hipLaunchKernelGGL(gs_reduce_slabs_pair_kernel, dim3(gs_cdiv(m >> 2, 256), 2), dim3(256), 0, (hipStream_t)stream, out0, slabs0, n0 >> 2, ns0,
                     stride0 >> 2, out1, slabs1, n1 >> 2, ns1, stride1 >> 2);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_reduce_slabs_f32(float* out, const float* slabs, long n, int n_slabs, long slab_stride, float beta,
                                   gs_stream_t stream) {
  if (n <= 0) return GS_OK;
  if (!out || !slabs || n_slabs < 1) GS_FAIL(GS_ERR_ARG, "gs_reduce_slabs_f32: bad argument");
  if ((n & 3) == 0 && (slab_stride & 3) == 0 && ((((uintptr_t)out) | ((uintptr_t)slabs)) & 15) == 0) {
    hipLaunchKernelGGL(gs_reduce_slabs_v4_kernel, dim3(gs_cdiv(n >> 2, 256)), dim3(256), 0, (hipStream_t)stream, out, slabs, n >> 2,
                       n_slabs, slab_stride >> 2, beta);
    GS_LAUNCH_CHECK();
    return GS_OK;
  }
  hipLaunchKernelGGL(gs_reduce_slabs_kernel, dim3(gs_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, out, slabs, n,
                     n_slabs, slab_stride, beta);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

__global__ void gs_reduce_slabs_batched_kernel(float* __restrict__ out, long out_stride, const float* __restrict__ slabs,
                                               long n, int ns, long slab_stride, float beta) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = blockIdx.y;
  const float* sl = slabs + (long)b * ns * slab_stride;
  float s = 0.f;
  for (int k = 0; k < ns; ++k) s += sl[(long)k * slab_stride + i];
  float* o = out + (long)b * out_stride;
  o[i] = (beta != 0.f ? beta * o[i] : 0.f) + s;
}

// The same with 16-byte accesses and a per-batch count of slabs that can be non-zero: slabs at or beyond
// ceil(kcount[b] / kchunk) hold exact zeros (their K range is empty) and are not read.  Same summation order as above.
__global__ void gs_reduce_slabs_active_kernel(float* __restrict__ out, long out_stride, const float* __restrict__ slabs, long n4,
                                              int ns, long slab_stride, const int* __restrict__ kcount, int kchunk) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int b = blockIdx.y;
  const int na = min(ns, (kcount[b] + kchunk - 1) / kchunk);
  const float4* sl = reinterpret_cast<const float4*>(slabs + (long)b * ns * slab_stride) + i;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < na; ++k) {
    const float4 v = sl[(long)k * (slab_stride >> 2)];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  reinterpret_cast<float4*>(out + (long)b * out_stride)[i] = s;
}

// engine-internal (gs_engine.hip): n, slab_stride, out_stride multiples of 4, 16-byte aligned pointers
int gs_reduce_slabs_active(float* out, long out_stride, const float* slabs, long n, int n_slabs, long slab_stride, int nbatch,
                           const int* kcount, int kchunk, gs_stream_t stream) {
  if (n <= 0 || nbatch <= 0) return GS_OK;
  hipLaunchKernelGGL(gs_reduce_slabs_active_kernel, dim3(gs_cdiv(n >> 2, 256), nbatch), dim3(256), 0, (hipStream_t)stream, out,
                     out_stride, slabs, n >> 2, n_slabs, slab_stride, kcount, kchunk);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_reduce_slabs_batched_f32(float* out, long out_stride, const float* slabs, long n, int n_slabs,
                                           long slab_stride, int nbatch, float beta, gs_stream_t stream) {
  if (n <= 0 || nbatch <= 0) return GS_OK;
  if (!out || !slabs || n_slabs < 1) GS_FAIL(GS_ERR_ARG, "gs_reduce_slabs_batched_f32: bad argument");
  hipLaunchKernelGGL(gs_reduce_slabs_batched_kernel, dim3(gs_cdiv(n, 256), nbatch), dim3(256), 0, (hipStream_t)stream, out,
                     out_stride, slabs, n, n_slabs, slab_stride, beta);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

static int g_bk_small = 32, g_bk_big = 0;   // 0 = by operand mode
   // K-tile depth of the small (latency-bound) and the 128x128 tiles

extern "C" int gs_gemm_set_tuning(int bk_small, int bk_big) {
  if ((bk_small != 16 && bk_small != 32) || (bk_big != 0 && bk_big != 16 && bk_big != 32))
    GS_FAIL(GS_ERR_ARG, "gs_gemm_set_tuning: BK must be 16 or 32 (bk_big = 0: chosen by operand mode)");
  g_bk_small = bk_small;
  g_bk_big = bk_big;
  return GS_OK;
}

template <int BM, int BN, int WM, int WN, int AM, int BMD, int BK>
static int launch_gemm3(const gs_gemm_desc& p, int nbatch, hipStream_t s) {
  dim3 grid((unsigned)((long)gs_cdiv(p.N, BN) * gs_cdiv(p.M, BM)), 1, nbatch);
  hipLaunchKernelGGL((gs_gemm_f32_kernel<BM, BN, WM, WN, AM, BMD, BK>), grid, dim3(256), 0, s, p);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

template <int BM, int BN, int WM, int WN, int AM, int BK>
static int launch_gemm_bgen(const gs_gemm_desc& p, int nbatch, hipStream_t s) {
  dim3 grid((unsigned)((long)gs_cdiv(p.N, BN) * gs_cdiv(p.M, BM)), 1, nbatch);
  hipLaunchKernelGGL((gs_gemm_f32_kernel<BM, BN, WM, WN, AM, GS_MODE_M, BK, true>), grid, dim3(256), 0, s, p);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

template <int BM, int BN, int WM, int WN, int AM, int BMD>
static int launch_gemm2(const gs_gemm_desc& p, int nbatch, hipStream_t s) {
  // the 128x128 tile: K step 32 when both operands are k-contiguous (two ds_read_b128 per fragment pair, half the
  // barriers: 264 vs 277 us at the PGE layer-2 shape), 16 otherwise (m-contiguous images get slower with 32: 311 vs 293 us)
  constexpr bool BIG = BM == 128 && BN == 128;
  int bk = BIG ? g_bk_big : g_bk_small;
  if (BIG && g_bk_big == 0) bk = (AM == GS_MODE_K && BMD == GS_MODE_K) ? 32 : 16;
  if (bk == 32) return launch_gemm3<BM, BN, WM, WN, AM, BMD, 32>(p, nbatch, s);
  return launch_gemm3<BM, BN, WM, WN, AM, BMD, 16>(p, nbatch, s);
}

template <int BM, int BN, int WM, int WN>
static int launch_gemm(const gs_gemm_desc& p, int nbatch, hipStream_t s) {
  if (p.bgen_mask) {   // generated B operand: A must be the m-contiguous gathered operand of the edge GEMM
    const int bk = (BM == 128 && BN == 128) ? g_bk_big : g_bk_small;
    if (bk == 32) return launch_gemm_bgen<BM, BN, WM, WN, GS_MODE_M, 32>(p, nbatch, s);
    return launch_gemm_bgen<BM, BN, WM, WN, GS_MODE_M, 16>(p, nbatch, s);
  }
  int am = (p.sAk == 1 && !p.gatherA_k) ? GS_MODE_K : ((p.sAm == 1 && !p.gatherA_m) ? GS_MODE_M : GS_MODE_G);
  int bm = (p.sBk == 1 && !p.gatherB_k) ? GS_MODE_K : ((p.sBn == 1) ? GS_MODE_M : GS_MODE_G);
  if (am == GS_MODE_G || bm == GS_MODE_G) return launch_gemm2<BM, BN, WM, WN, GS_MODE_G, GS_MODE_G>(p, nbatch, s);
  if (am == GS_MODE_K && bm == GS_MODE_K) return launch_gemm2<BM, BN, WM, WN, GS_MODE_K, GS_MODE_K>(p, nbatch, s);
  if (am == GS_MODE_K && bm == GS_MODE_M) return launch_gemm2<BM, BN, WM, WN, GS_MODE_K, GS_MODE_M>(p, nbatch, s);
  if (am == GS_MODE_M && bm == GS_MODE_K) return launch_gemm2<BM, BN, WM, WN, GS_MODE_M, GS_MODE_K>(p, nbatch, s);
  return launch_gemm2<BM, BN, WM, WN, GS_MODE_M, GS_MODE_M>(p, nbatch, s);
}

static int pick_tile(const gs_gemm_desc& p, int nbatch) {   // 0: 128x32, 1: 32x128, 2: 64x64, 3: 128x128
  if (p.N <= 32) return 0;
  if (p.M <= 32) return 1;
  long tiles128 = (long)gs_cdiv(p.M, 128) * gs_cdiv(p.N, 128) * nbatch;
  if (p.M <= 64 || p.N <= 64 || tiles128 < 192) return 2;
  return 3;
}

// which kernel gs_gemm_f32 launches for this descriptor: 0 = 128x32 tile, 1 = 32x128, 2 = 64x64, 3 = 128x128 (the tile of the
// PGE / H1full / edge products), 4 = the latency-oriented one-tile-per-workgroup kernel.  Lets the parity tests assert that a
// shape really exercises the instantiation it is meant to cover.
extern "C" int gs_gemm_route(const gs_gemm_desc* d) {
  if (!d) return -1;
  gs_gemm_desc p = *d;
  if (p.nb1 < 1) p.nb1 = 1;
  if (p.nb2 < 1) p.nb2 = 1;
  const int nb = p.nb1 * p.nb2;
  if (gs_panel_eligible(p)) return 5;
  if (gs_edge_eligible(p)) return 6;
  if (small_route(p, nb)) return 4;
  return pick_tile(p, nb);
}

extern "C" int gs_gemm_tiles_m(const gs_gemm_desc* d) {
  if (!d) return 0;
  const int nb = (d->nb1 < 1 ? 1 : d->nb1) * (d->nb2 < 1 ? 1 : d->nb2);
  if (gs_panel_eligible(*d)) return gs_panel_tiles(d->M);
  static const int bm[4] = {128, 32, 64, 128};
  return gs_cdiv(d->M, bm[pick_tile(*d, nb)]);
}

extern "C" int gs_gemm_f32(const gs_gemm_desc* d, gs_stream_t stream) {
  if (!d || !d->A || !d->B || !d->C) GS_FAIL(GS_ERR_ARG, "gs_gemm_f32: null operand");
  gs_gemm_desc p = *d;
  if (p.M <= 0 || p.N <= 0) return GS_OK;
  if (p.K < 0) GS_FAIL(GS_ERR_ARG, "gs_gemm_f32: negative K");
  if (p.colstat && (p.nb1 > 1 || p.nb2 > 1)) GS_FAIL(GS_ERR_ARG, "gs_gemm_f32: colstat is for unbatched calls");
  if (p.bgen_mask && (!p.bgen_idx || !p.bgen_w || p.bgen_div < 1 || p.sAm != 1 || p.gatherA_m || p.gatherB_k))
    GS_FAIL(GS_ERR_ARG, "gs_gemm_f32: bgen needs idx, w, div >= 1 and an m-contiguous A");
  if (p.nb1 < 1) p.nb1 = 1;
  if (p.nb2 < 1) p.nb2 = 1;
  const int nbatch = p.nb1 * p.nb2;
  if ((long)nbatch > 65535) GS_FAIL(GS_ERR_ARG, "gs_gemm_f32: too many batches (%d)", nbatch);
  hipStream_t s = (hipStream_t)stream;
  if (gs_panel_eligible(p)) return gs_panel_launch(p, s);
  if (p.signbits) GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gemm_f32: signbits is an output of the row-panel kernel only (N 128 / 256, k-contiguous operands, unbatched)");
  if (gs_edge_eligible(p)) return gs_edge_launch(p, s);
  if (small_route(p, nbatch)) return launch_small(p, nbatch, s);
  // tile choice: fill the 256 CUs; narrow outputs get the tall/flat tiles
  switch (pick_tile(p, nbatch)) {
    case 0: return launch_gemm<128, 32, 32, 32>(p, nbatch, s);
    case 1: return launch_gemm<32, 128, 32, 32>(p, nbatch, s);
    case 2: return launch_gemm<64, 64, 32, 32>(p, nbatch, s);
    default: return launch_gemm<128, 128, 64, 64>(p, nbatch, s);   // (128x64 / 64x128 at 4 waves/SIMD measured within +-5 %)
  }
}

static bool small_route(const gs_gemm_desc& p, int nbatch);
// engine-internal: members descs[i] (mutually independent) with an optional dependent second product second[i] (second[i].A == null:
// none) accumulating into the same output with beta = 1.  One launch when everything is routed to the latency-oriented kernel,
// otherwise the same products one launch after the other — bitwise the same values either way.
int gs_gemm_group2_f32(const gs_gemm_desc* descs, const gs_gemm_desc* second, int n, gs_stream_t stream) {
  if (n <= 0) return GS_OK;
  bool groupable = n <= GS_GROUP_MAX;
  gs_gemm_group2_arg a;
  int gx = 1, gy = 1, z = 0;
  long valid = 0;
  for (int i = 0; groupable && i < n; ++i) {
    gs_gemm_desc p = descs[i];
    if (!p.A || !p.B || !p.C || p.K < 0) { groupable = false; break; }
    if (p.nb1 < 1) p.nb1 = 1;
    if (p.nb2 < 1) p.nb2 = 1;
    const int nbatch = p.nb1 * p.nb2;
    if (p.M <= 0 || p.N <= 0 || !small_route(p, nbatch)) { groupable = false; break; }
    a.d[i] = p; a.has2[i] = 0; a.d2[i] = p;
    if (second && second[i].A) {
      gs_gemm_desc q = second[i];
      if (q.nb1 < 1) q.nb1 = 1;
      if (q.nb2 < 1) q.nb2 = 1;
      if (q.C != p.C || q.M != p.M || q.N != p.N || q.nb1 != p.nb1 || q.nb2 != p.nb2 || q.ldc != p.ldc || q.sCb1 != p.sCb1 || q.sCb2 != p.sCb2 ||
          q.offC != p.offC || q.Mb != p.Mb || q.beta != 1.f || q.kchunk > 0 || p.kchunk > 0 || !small_route(q, nbatch)) { groupable = false; break; }
      a.d2[i] = q; a.has2[i] = 1;
    }
    a.z0[i] = z;
    z += nbatch;
    valid += (long)gs_cdiv(p.N, 32) * gs_cdiv(p.M, 32) * nbatch;
    gx = gx > gs_cdiv(p.N, 32) ? gx : gs_cdiv(p.N, 32);
    gy = gy > gs_cdiv(p.M, 32) ? gy : gs_cdiv(p.M, 32);
  }
  if (groupable && (z > 65535 || (long)gx * gy * z > 4 * valid)) groupable = false;
  if (!groupable) {
    for (int i = 0; i < n; ++i) {
      GS_TRY(gs_gemm_f32(&descs[i], stream));
      if (second && second[i].A) GS_TRY(gs_gemm_f32(&second[i], stream));
    }
    return GS_OK;
  }
  for (int i = n; i <= GS_GROUP_MAX; ++i) a.z0[i] = z;
  for (int i = n; i < GS_GROUP_MAX; ++i) { a.d[i] = a.d[0]; a.d2[i] = a.d[0]; a.has2[i] = 0; }
  a.n = n;
  hipLaunchKernelGGL(gs_gemm_small_group2_kernel, dim3(gx, gy, z), dim3(256), 0, (hipStream_t)stream, a);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// engine-internal: gs_gemm_group2_f32 for members that all run over the SAME batches with per-batch row counts (Mb, nb2 = 1), given the
// list of M tiles that exist (tile_b: batch index + b_first, tile_my: M tile; n_tiles entries).  Falls back to gs_gemm_group2_f32.
int gs_gemm_group2_tiles_f32(const gs_gemm_desc* descs, const gs_gemm_desc* second, int n, const int* tile_b, const int* tile_my, int n_tiles,
                             int b_first, gs_stream_t stream) {
  if (n <= 0 || n_tiles <= 0) return GS_OK;
  bool ok = n <= GS_GROUP_MAX && tile_b && tile_my;
  gs_gemm_group2_arg a;
  int gx = 1;
  for (int i = 0; ok && i < n; ++i) {
    gs_gemm_desc p = descs[i];
    if (!p.A || !p.B || !p.C || p.K < 0 || !p.Mb || p.nb1 < 1 || p.nb2 > 1 || p.kchunk > 0 || p.M <= 0 || p.N <= 0) { ok = false; break; }
    if (p.nb1 != descs[0].nb1 || p.Mb != descs[0].Mb) { ok = false; break; }
    p.nb2 = 1;
    if (!small_route(p, p.nb1)) { ok = false; break; }
    a.d[i] = p; a.has2[i] = 0; a.d2[i] = p;
    if (second && second[i].A) {
      gs_gemm_desc q = second[i];
      if (q.nb1 < 1) q.nb1 = 1;
      q.nb2 = 1;
      if (q.C != p.C || q.M != p.M || q.N != p.N || q.nb1 != p.nb1 || q.ldc != p.ldc || q.sCb1 != p.sCb1 || q.offC != p.offC || q.Mb != p.Mb ||
          q.beta != 1.f || q.kchunk > 0 || !small_route(q, p.nb1)) { ok = false; break; }
      a.d2[i] = q; a.has2[i] = 1;
    }
    gx = gx > gs_cdiv(p.N, 32) ? gx : gs_cdiv(p.N, 32);
  }
  if (!ok || n_tiles > 65535) return gs_gemm_group2_f32(descs, second, n, stream);
  for (int i = 0; i <= GS_GROUP_MAX; ++i) a.z0[i] = 0;
  for (int i = n; i < GS_GROUP_MAX; ++i) { a.d[i] = a.d[0]; a.d2[i] = a.d[0]; a.has2[i] = 0; }
  a.n = n;
  hipLaunchKernelGGL(gs_gemm_small_group2_tiles_kernel, dim3(gx, n_tiles, n), dim3(256), 0, (hipStream_t)stream, a, tile_b, tile_my, b_first);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// Independent products in one launch when every one of them is routed to the latency-oriented kernel; otherwise (or for a
// single product) they are launched one after the other.  The caller guarantees that no product reads another's output.
extern "C" int gs_gemm_group_f32(const gs_gemm_desc* descs, int n, gs_stream_t stream) {
  if (n <= 0) return GS_OK;
  if (!descs) GS_FAIL(GS_ERR_ARG, "gs_gemm_group_f32: null descriptors");
  bool groupable = n > 1 && n <= GS_GROUP_MAX;
  gs_gemm_group_arg g;
  int gx = 1, gy = 1, z = 0;
  long valid = 0;
  if (groupable) {
    for (int i = 0; i < n; ++i) {
      gs_gemm_desc p = descs[i];
      if (!p.A || !p.B || !p.C || p.K < 0) { groupable = false; break; }
      if (p.nb1 < 1) p.nb1 = 1;
      if (p.nb2 < 1) p.nb2 = 1;
      const int nbatch = p.nb1 * p.nb2;
      if (p.M <= 0 || p.N <= 0 || !small_route(p, nbatch)) { groupable = false; break; }
      g.d[i] = p;
      g.z0[i] = z;
      z += nbatch;
      valid += (long)gs_cdiv(p.N, 32) * gs_cdiv(p.M, 32) * nbatch;
      gx = gx > gs_cdiv(p.N, 32) ? gx : gs_cdiv(p.N, 32);
      gy = gy > gs_cdiv(p.M, 32) ? gy : gs_cdiv(p.M, 32);
    }
    if (z > 65535) groupable = false;
    // the launch covers the LARGEST tile grid for every (product, batch): very unequal shapes would mostly launch workgroups
    // that exit at once (measured: a 160x mismatch cost more than the launch it saved)
    if ((long)gx * gy * z > 4 * valid) groupable = false;
  }
  if (!groupable) {
    for (int i = 0; i < n; ++i) GS_TRY(gs_gemm_f32(&descs[i], stream));
    return GS_OK;
  }
  for (int i = n; i <= GS_GROUP_MAX; ++i) g.z0[i] = z;
  for (int i = n; i < GS_GROUP_MAX; ++i) g.d[i] = g.d[0];
  g.n = n;
  hipLaunchKernelGGL(gs_gemm_small_group_kernel, dim3(gx, gy, z), dim3(256), 0, (hipStream_t)stream, g);
  GS_LAUNCH_CHECK();
  return GS_OK;
}
