// Fused launches of the latency-bound chains of the GCond iteration (VERDICT r02 "next" 5): the inner SGC steps
// (condensation/gcond.py:63-72 -> models/base.py fit loop: forward, NLL, backward, Adam on the n' synthetic rows) and the
// class-independent head of the synthetic match (gcond_base.py:226-230).  Every product here is tiny (n' = 454 rows, <= 0.2 GFLOP):
// a launch costs what its memory round trips cost, so the chain is shortened — fewer, lighter dependent launches — rather than
// made faster per launch:
//   * gs_chain_zg_kernel   Z = A B (+ rs (x) bias) with the row softmax and the NLL gradient in the epilogue: the logits never
//                          reach memory; a workgroup owns 32 rows x all C <= 64 classes.
//   * gs_chain_group_kernel  up to four independent small products in ONE flat tile list (no empty workgroups, unlike the
//                          rectangular grid of gs_gemm_small_group_kernel) with an optional fused Adam step on the outputs
//                          (gs_adam_ep): the gradient products of a step update theta themselves.
// Algebra of one SGC step with two transforms (theta = [W1, b1, Wl, bl]; Q = A^ A^, rs = Q 1):
//     H1 = relu(X W1 + b1)         T = H1 Wl            Z = Q T + rs (x) bl      G = (softmax(Z) - Y) / n'
//     R  = Q^T G                   dH1 = (R Wl^T) * [H1 > 0]
//     gWl = H1^T R                 gbl = rs^T G          gW1 = X^T dH1            gb1 = 1^T dH1
// i.e. the two n' x n' x h products of the direct form (S = Q H1, Q^T dS) become n' x n' x C ones (C = 40 against h = 256) by
// associativity: (Q H1) Wl = Q (H1 Wl) and Q^T (G Wl^T) = (Q^T G) Wl^T.  Six launches per step instead of nine, 0.12 GFLOP instead
// of 0.30.  Same value up to fp32 re-association (tests/test_gpu_engine.py::test_fused_inner_chain_*).
#include "gs_small.h"

struct gs_zg_args {
  const float* A; long sAm, sAk;     // [n x K]
  const float* B; long sBk, sBn;     // [K x C]
  int n, K, C;
  const float* rs;                   // [n] row factor of the bias (rs (x) bias); null = 1
  const float* bias;                 // [C]
  const int* labels; const int* syn_lo; const int* syn_hi;
  int divide_by_n;                   // 1: mean over all n rows (inner steps); 0: mean over the rows of the row's class (matching)
  float* P;                          // optional [n x C] softmax
  float* G;                          // [n x C] (P - onehot) / count
};

// NW waves split K (8 for the n'-deep products: one round of loads and <= 32 MFMA k-steps per wave; the first form's 4 waves needed
// two dependent rounds and ran 16.6 us on its 15 workgroups)
template <bool AVEC, bool BVEC, int NT, int NW>
__global__ __launch_bounds__(64 * NW) void gs_chain_zg_kernel(gs_zg_args p) {
  constexpr int QS = 8;
  __shared__ float red[NW][NT][16][64];
  __shared__ float zt[32][32 * NT + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int m0 = blockIdx.x * 32;
  const int am = min(m0 + li, p.n - 1);
  const float* __restrict__ ap = p.A + (long)am * p.sAm;
  const float* bp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bp[t] = p.B + (long)min(32 * t + li, p.C - 1) * p.sBn;
  const int per = ((p.K + 8 * NW - 1) / (8 * NW)) * 8;
  const int w_lo = min(wave * per, p.K), w_hi = min(p.K, w_lo + per);
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int k0 = w_lo; k0 < w_hi; k0 += 8 * QS) {
    float a[QS][4], bb[NT][QS][4];
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      const int ks = k0 + 8 * q;
      if (ks < w_hi) {
        const bool whole = ks + 8 <= w_hi;
        small_frag<AVEC>(ap, p.sAk, ks + 4 * kh, w_hi, whole, a[q]);
#pragma unroll
        for (int t = 0; t < NT; ++t) small_frag<BVEC>(bp[t], p.sBk, ks + 4 * kh, w_hi, whole, bb[t][q]);
      }
    }
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      if (k0 + 8 * q < w_hi) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][u], bb[t][q][u], acc[t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][t][r][lane] = acc[t][r];
  __syncthreads();
  // accumulator register r of tile t, lane: row (r & 3) + 8 (r >> 2) + 4 kh, column 32 t + li; the NW partial tiles are summed in
  // wave order (fixed: reproducible); the 16 NT registers are dealt round-robin to the waves
  for (int q = wave; q < 16 * NT; q += NW) {
    const int t = q >> 4, r = q & 15;
    float sum = red[0][t][r][lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) sum += red[w][t][r][lane];
    zt[(r & 3) + 8 * (r >> 2) + 4 * kh][32 * t + li] = sum;
  }
  __syncthreads();
  if (wave >= 4) return;
  // row softmax + NLL gradient: wave w < 4 owns rows 8w .. 8w+7, lane = class.  The per-row scalars (bias factor, label, class size)
  // are fetched by lanes 0..7 in ONE round trip before the row loop (fetched inside it they were eight dependent round trips:
  // 17.8 us for this launch) and broadcast with a lane read.
  const bool on = lane < p.C && lane < 32 * NT;
  const float bk = on ? p.bias[lane] : 0.f;
  float r_l = 1.f, invn_l = 0.f;
  int y_l = 0;
  {
    const int row = min(m0 + 8 * wave + (lane & 7), p.n - 1);
    if (p.rs) r_l = p.rs[row];
    y_l = p.labels[row];
    invn_l = 1.f / (float)(p.divide_by_n ? p.n : (p.syn_hi[y_l] - p.syn_lo[y_l]));
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int row = m0 + 8 * wave + q;
    const float r = __shfl(r_l, q, 64), invn = __shfl(invn_l, q, 64);
    const int y = __shfl(y_l, q, 64);
    const float z = on ? zt[8 * wave + q][lane] + r * bk : -INFINITY;
    const float mx = gs_wave_max(z);
    const float ex = on ? expf(z - mx) : 0.f;
    const float se = gs_wave_sum(ex);
    if (on && row < p.n) {
      const float pr = ex * (1.f / se);
      if (p.P) p.P[(long)row * p.C + lane] = pr;
      p.G[(long)row * p.C + lane] = (pr - (lane == y ? 1.f : 0.f)) * invn;
    }
  }
}

// engine-internal.  GS_ERR_UNSUPPORTED for C > 64 (the caller keeps the product + softmax launches).
int gs_chain_zg(const gs_zg_args* a, hipStream_t s) {
  if (!a || a->C > 64 || a->C < 1 || a->n < 1 || a->K < 1) return GS_ERR_UNSUPPORTED;
  const dim3 grid(gs_cdiv(a->n, 32));
  const bool av = a->sAk == 1, bv = a->sBk == 1;
  const bool deep = a->K > 256;      // 8 waves when 4 would need more than one round of fragment loads
#define GS_ZG(AV_, BV_)                                                                                              \
  do {                                                                                                               \
    if (a->C > 32 && deep) hipLaunchKernelGGL((gs_chain_zg_kernel<AV_, BV_, 2, 8>), grid, dim3(512), 0, s, *a);      \
    else if (a->C > 32) hipLaunchKernelGGL((gs_chain_zg_kernel<AV_, BV_, 2, 4>), grid, dim3(256), 0, s, *a);         \
    else if (deep) hipLaunchKernelGGL((gs_chain_zg_kernel<AV_, BV_, 1, 8>), grid, dim3(512), 0, s, *a);              \
    else hipLaunchKernelGGL((gs_chain_zg_kernel<AV_, BV_, 1, 4>), grid, dim3(256), 0, s, *a);                        \
  } while (0)
  if (av && bv) GS_ZG(true, true);
  else if (av) GS_ZG(true, false);
  else if (bv) GS_ZG(false, true);
  else GS_ZG(false, false);
#undef GS_ZG
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// ---- flat group of small products -------------------------------------------------------------------------------------
#define GS_FLAT_MAX 4
struct gs_flat_group {
  gs_gemm_desc d[GS_FLAT_MAX];
  int t0[GS_FLAT_MAX + 1];     // first flat tile of each product
  int gx[GS_FLAT_MAX];         // tiles along N of each product
  int gxy[GS_FLAT_MAX];        // tiles per batch of each product
};

template <bool ADAM>
__device__ __forceinline__ void flat_any(const gs_gemm_desc& p, int local, int gx, int gxy, float (&red)[4][16][64], const gs_adam_ep* ep) {
  const int b = local / gxy, t = local - b * gxy;
  const int by = t / gx, bx = t - by * gx;
  const bool av = p.sAk == 1, bv = p.sBk == 1;   // block-uniform
  if (av && bv) gs_small_tile<true, true, ADAM>(p, bx, by, b, red, ep);
  else if (av) gs_small_tile<true, false, ADAM>(p, bx, by, b, red, ep);
  else if (bv) gs_small_tile<false, true, ADAM>(p, bx, by, b, red, ep);
  else gs_small_tile<false, false, ADAM>(p, bx, by, b, red, ep);
}

template <bool ADAM>
__global__ __launch_bounds__(256) void gs_chain_group_kernel(gs_flat_group g, gs_adam_ep ep) {
  __shared__ float red[4][16][64];
  const int t = blockIdx.x;
  if (t < g.t0[1]) flat_any<ADAM>(g.d[0], t, g.gx[0], g.gxy[0], red, &ep);
  else if (t < g.t0[2]) flat_any<ADAM>(g.d[1], t - g.t0[1], g.gx[1], g.gxy[1], red, &ep);
  else if (t < g.t0[3]) flat_any<ADAM>(g.d[2], t - g.t0[2], g.gx[2], g.gxy[2], red, &ep);
  else flat_any<ADAM>(g.d[3], t - g.t0[3], g.gx[3], g.gxy[3], red, &ep);
}

// engine-internal: n <= 4 mutually independent small products (no K split across workgroups, no gathers along k, no generated
// operand, no column statistics) in one launch of exactly their tiles; adam != null: every output element also steps its theta.
int gs_chain_group(const gs_gemm_desc* descs, int n, const gs_adam_ep* adam, hipStream_t s) {
  if (n < 1 || n > GS_FLAT_MAX || !descs) GS_FAIL(GS_ERR_ARG, "gs_chain_group: 1 .. 4 products");
  gs_flat_group g;
  int z = 0;
  for (int i = 0; i < n; ++i) {
    gs_gemm_desc p = descs[i];
    if (p.nb1 < 1) p.nb1 = 1;
    if (p.nb2 < 1) p.nb2 = 1;
    if (!p.A || !p.B || !p.C || p.M < 1 || p.N < 1 || p.K < 0 || p.bgen_mask || p.colstat || p.gatherA_k || p.gatherB_k ||
        (adam && (p.kchunk > 0 || p.nb1 * p.nb2 != 1)))
      GS_FAIL(GS_ERR_ARG, "gs_chain_group: product %d is not a plain small product", i);
    g.d[i] = p;
    g.t0[i] = z;
    g.gx[i] = gs_cdiv(p.N, 32);
    g.gxy[i] = g.gx[i] * gs_cdiv(p.M, 32);
    z += g.gxy[i] * p.nb1 * p.nb2;
  }
  for (int i = n; i <= GS_FLAT_MAX; ++i) g.t0[i] = z;
  for (int i = n; i < GS_FLAT_MAX; ++i) { g.d[i] = g.d[0]; g.gx[i] = g.gx[0]; g.gxy[i] = g.gxy[0]; }
  gs_adam_ep ep;
  memset(&ep, 0, sizeof(ep));
  if (adam) {
    ep = *adam;
    hipLaunchKernelGGL((gs_chain_group_kernel<true>), dim3(z), dim3(256), 0, s, g, ep);
  } else {
    hipLaunchKernelGGL((gs_chain_group_kernel<false>), dim3(z), dim3(256), 0, s, g, ep);
  }
  GS_LAUNCH_CHECK();
  return GS_OK;
}

