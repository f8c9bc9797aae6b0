// The latency-oriented small-product tile (one 32x32 output tile per workgroup, K split four ways over the waves, fragments
// fetched straight from global memory) — shared by gs_gemm.hip (gs_gemm_small_kernel / gs_gemm_small_group_kernel) and
// gs_chain.hip (the fused inner-step / matching chains, which add an epilogue to it).
#pragma once
#include "gs_common.h"

#ifndef GS_F32X16_DEFINED
#define GS_F32X16_DEFINED
typedef float f32x16 __attribute__((ext_vector_type(16)));
// 16-byte global load that only needs 4-byte alignment (gfx950 runs in unaligned-access mode; hipcc emits
// global_load_dwordx4 for it): rows whose stride is not a multiple of 4 floats (n' = 454) stay on the vector path
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
#endif

// Optional fused Adam step on the product's output (torch.optim.Adam, the arithmetic of gs_adam_kernel): the output tile is a
// slice of a flat gradient vector starting at gbase; the element at the same offset of theta / m / v is stepped by the workgroup
// that produced its gradient (valid because a small-product tile owns its output elements: no K split across workgroups).
struct gs_adam_ep {
  float* theta; float* m; float* v; const float* gbase;
  float step_size, inv_sqrt_bc2, b1, b2, eps;
};

// ---------------------------------------------------------------------------------------------------------------
// Latency-bound shapes (the synthetic-graph side: n' x {d, h, C, n'} with K <= a few thousand).  With 64x64 tiles such a
// product fills 30-60 of the 256 CUs and every K step pays one global-load latency (~1.5 us): 17-35 us for 30-190 MFLOP.
// This kernel gives each workgroup ONE 32x32 output tile, splits K four ways over its waves, and lets each wave fetch its
// MFMA fragments straight from global memory (no LDS staging, no barriers in the K loop) 64 k at a time — 16 loads per
// operand in flight per lane — so the whole product costs about one memory latency + K/8 MFMAs.  The four partial tiles
// are summed through LDS in wave order (fixed order: reproducible).  Same k permutation as the tiled kernel
// (k = 8q + 4*(lane>>5) + t), same epilogue.  Not for gathers along k, generated B or column statistics.
template <bool VEC>
__device__ __forceinline__ void small_frag(const float* __restrict__ rowp, long sk, int kb, int w_hi, bool whole, float (&f)[4]) {
  if (whole) {            // wave-uniform: the q-step lies inside [.., w_hi)
    if (VEC) {
      const f32x4u t = *reinterpret_cast<const f32x4u*>(rowp + kb);
      f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) f[t] = rowp[(long)(kb + t) * sk];
    }
  } else {                // the ragged last q-step of the K range: clamped addresses, zero fill
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float x = rowp[(long)min(kb + t, w_hi - 1) * sk];
      f[t] = (kb + t < w_hi) ? x : 0.f;
    }
  }
}

template <bool AVEC, bool BVEC, bool ADAM = false>
__device__ __forceinline__ void gs_small_tile(const gs_gemm_desc& p, int bx, int by, int b, float (&red)[4][16][64],
                                              const gs_adam_ep* ep = nullptr) {
  constexpr int QS = 8;   // q-steps (8 k each) per round
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int b1 = b / p.nb2, b2 = b - b1 * p.nb2;
  int M = p.M, K = p.K;
  if (p.Mb) M = p.Mb[b1];
  if (p.Kb) K = p.Kb[b1];
  int k_lo = 0, k_hi = K;
  if (p.kchunk > 0) {
    k_lo = b2 * p.kchunk;
    k_hi = min(K, k_lo + p.kchunk);
  }
  const int m0 = by * 32, n0 = bx * 32;
  if (m0 >= M || n0 >= p.N) return;
  const float* __restrict__ A = p.A + (long)b1 * p.sAb1 + (p.kchunk > 0 ? 0 : (long)b2 * p.sAb2) + (p.offA ? p.offA[b1] : 0);
  const float* __restrict__ B = p.B + (long)b1 * p.sBb1 + (p.kchunk > 0 ? 0 : (long)b2 * p.sBb2) + (p.offB ? p.offB[b1] : 0);
  float* __restrict__ C = p.C + (long)b1 * p.sCb1 + (long)b2 * p.sCb2 + (p.offC ? p.offC[b1] : 0);
  const int am = min(m0 + li, M - 1), bn = min(n0 + li, p.N - 1);
  const long arow = p.gatherA_m ? (long)p.gatherA_m[(long)b1 * p.sGb1 + am] : (long)am;
  const float* __restrict__ ap = A + arow * p.sAm;
  const float* __restrict__ bp = B + (long)bn * p.sBn;

  // this wave's K range: a quarter of [k_lo, k_hi), in whole q-steps
  const int per = ((k_hi - k_lo + 31) / 32) * 8;
  const int w_lo = k_lo + wave * per, w_hi = min(k_hi, w_lo + per);

  // epilogue operands (bias, ReLU-derivative mask, beta * C) do not depend on the product: they are requested BEFORE the K loop, so
  // that their round trip overlaps the fragment loads instead of following the LDS reduction (one dependent memory round trip less
  // per launch of a latency-bound chain)
  const int col = n0 + li, colc = min(col, p.N - 1);
  float bv = 0.f;
  if (p.bias) bv = p.bias[(long)b1 * p.sbias_b1 + colc];
  const bool has_mask = p.mask != nullptr, has_beta = p.beta != 0.f;
  float mk[4], cv[4];
  const int rbase = m0 + 8 * wave + 4 * kh;
  if (has_mask) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) mk[rr] = p.mask[(long)min(rbase + rr, M - 1) * p.ldmask + colc];
  }
  if (has_beta) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) cv[rr] = C[(long)min(rbase + rr, M - 1) * p.ldc + colc];
  }

  float adm[4], adv[4], adp[4];      // (fused Adam: this lane's moments and parameters, requested ahead as well)
  if constexpr (ADAM) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const long i = (C + (long)min(rbase + rr, M - 1) * p.ldc + colc) - ep->gbase;
      adm[rr] = ep->m[i]; adv[rr] = ep->v[i]; adp[rr] = ep->theta[i];
    }
  }

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = w_lo; k0 < w_hi; k0 += 8 * QS) {
    float a[QS][4], bb[QS][4];
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      const int ks = k0 + 8 * q;
      if (ks < w_hi) {                       // wave-uniform
        const bool whole = ks + 8 <= w_hi;
        small_frag<AVEC>(ap, p.sAk, ks + 4 * kh, w_hi, whole, a[q]);
        small_frag<BVEC>(bp, p.sBk, ks + 4 * kh, w_hi, whole, bb[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      if (k0 + 8 * q < w_hi) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][t], bb[q][t], acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  // wave w finishes accumulator registers r = 4w .. 4w+3: rows m0 + 8w + {0..3} + 4*(lane>>5), column n0 + (lane&31)
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = 4 * wave + rr;
    const float sum = ((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane];
    float v = p.alpha * sum + bv;
    if (p.relu) v = fmaxf(v, 0.f);
    if (has_mask) v = (mk[rr] > 0.f) ? v : 0.f;
    if (has_beta) v += p.beta * cv[rr];
    const int row = rbase + rr;
    if (row < M && col < p.N) {
      C[(long)row * p.ldc + col] = v;
      if constexpr (ADAM) {
        const long i = (C + (long)row * p.ldc + col) - ep->gbase;
        float mi = adm[rr], vi = adv[rr];
        const float pi = adp[rr];
        mi = mi + (1.f - ep->b1) * (v - mi);
        vi = ep->b2 * vi + (1.f - ep->b2) * v * v;
        const float denom = sqrtf(vi) * ep->inv_sqrt_bc2 + ep->eps;
        ep->theta[i] = pi - ep->step_size * (mi / denom);
        ep->m[i] = mi;
        ep->v[i] = vi;
      }
    }
  }
}

