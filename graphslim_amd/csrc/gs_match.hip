// Per-class products of the synthetic match for the two-transform SGC condense model (gcond_base.py:226-239 and the closed-form
// second-order backward of oracle/analytic.py:syn_match_sgc), as TWO chained launches instead of six batched products:
//
//   first   dHp_c = (Q[lo:hi]^T T[lo:hi]) * [H1 > 0]   (T = G Wl^T: K = n_c instead of the C-deep (Q^T G) Wl^T, by associativity)
//           gW1_c = X^T dHp_c ,  gb1_c = 1^T dHp_c
//   second  A1m_c = (X D1_c + 1 Db1_c) * [H1 > 0]      (never written)     Bc_c = A1m_c Wl
//           dX slab_c = dHp_c D1_c^T
//
// The batched latency-oriented tile (gs_small.h) re-fetches both operands of every 32 x 32 output tile from L2: at 40 classes x
// (454 x 256 x 128) these products ran at ~30 TFLOP/s (81 + 28 + 19 + 18 us).  Here a wave keeps the tile it has just produced in
// its accumulator registers and feeds it to the next product as the B operand: the C / D layout of v_mfma_f32_32x32x2_f32 (lane =
// column, register r = row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) IS a valid B-operand schedule when the chained dimension is the
// first product's M — step s of the second product takes k = row(s, lane >> 5) = accumulator register s — so the intermediate never
// touches LDS or memory, and each operand that comes from memory is fetched once per workgroup.  fp32 MFMA throughout: results are
// fp32 products summed in a fixed order (reproducible), re-associated relative to the batched kernels.
#include "gs_small.h"
#include "gs_match.h"

// grid (h / 32, Cl, ceil(d / 128)); 4 waves, each takes every fourth 32-row chunk of the n' rows
__global__ __launch_bounds__(256, 2) void gs_match_first_kernel(gs_match1_args p) {
  __shared__ float red[4][64][64];      // the four waves' partial [128 x 32] tiles
  __shared__ float cs[4][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int c = blockIdx.y, col0 = blockIdx.x * 32, j0 = blockIdx.z * 128;
  const int lo = (int)p.lo[c], nc = p.nc[c], n = p.n, h = p.h;
  const float* __restrict__ Qc = p.Q + (long)lo * n;
  const float* __restrict__ Tc = p.T + (long)lo * h + col0 + li;
  float* __restrict__ dHp = p.dHp + (long)c * p.s_dhp;
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float csum = 0.f;
  int jc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) jc[t] = min(j0 + 32 * t + li, p.d - 1);
  const int nchunk = (n + 31) >> 5;
  for (int rc = wave; rc < nchunk; rc += 4) {
    const int r0 = rc * 32;
    // this chunk's X fragments (A operand of the second product: m = feature column, k = row) — requested first
    float xa[16][4];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int row = min(r0 + (s & 3) + 8 * (s >> 2) + 4 * kh, n - 1);
#pragma unroll
      for (int t = 0; t < 4; ++t) xa[s][t] = p.X[(long)row * p.ldx + jc[t]];
    }
    float mk[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) mk[r] = p.mask[(long)min(r0 + (r & 3) + 8 * (r >> 2) + 4 * kh, n - 1) * h + col0 + li];
    // dHp tile = Q[lo:hi, r0:r0+32]^T T[lo:hi, col0:col0+32]
    f32x16 t16;
#pragma unroll
    for (int r = 0; r < 16; ++r) t16[r] = 0.f;
    const int rm = min(r0 + li, n - 1);
    for (int k0 = 0; k0 < nc; k0 += 16) {
      float qa[8], tb[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int k = k0 + 2 * s + kh, kk = min(k, nc - 1);
        const float q = Qc[(long)kk * n + rm];
        qa[s] = k < nc ? q : 0.f;
        tb[s] = Tc[(long)kk * h];
      }
#pragma unroll
      for (int s = 0; s < 8; ++s)
        if (k0 + 2 * s < nc) t16 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[s], tb[s], t16, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      const float v = (mk[r] > 0.f && row < n) ? t16[r] : 0.f;
      t16[r] = v;
      csum += v;
      if (blockIdx.z == 0 && row < n) dHp[(long)row * h + col0 + li] = v;
    }
    // gW1 tile += X[r0:r0+32, j0:j0+128]^T dHp tile: the accumulator registers ARE the B operand (register s = k of step s)
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[s][t], t16[s], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][16 * t + r][lane] = acc[t][r];
  csum += __shfl_xor(csum, 32, 64);
  if (lane < 32) cs[wave][lane] = csum;
  __syncthreads();
  // wave w finishes feature tile w (fixed wave order: reproducible)
  float* __restrict__ gW = p.gW1 + (long)c * p.s_gw;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int q = 16 * wave + r;
    const float v = ((red[0][q][lane] + red[1][q][lane]) + red[2][q][lane]) + red[3][q][lane];
    const int j = j0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kh;
    if (j < p.d) gW[(long)j * h + col0 + li] = v;
  }
  if (p.gb1 && blockIdx.z == 0 && tid < 32) p.gb1[(long)c * p.s_gw + col0 + tid] = ((cs[0][tid] + cs[1][tid]) + cs[2][tid]) + cs[3][tid];
}

// grid ceil(n / 32) * Cl (flat); 4 waves.  HT = h / 128 hidden tiles per wave, CT = ceil(C / 32) <= 2.  Fragment loads run one round
// (32 k) ahead of the MFMAs in two register buffers; the first round of the next product is requested before the current one ends.
template <int HT, int CT>
__global__ __launch_bounds__(256, 2) void gs_match_second_kernel(gs_match2_args p) {
  __shared__ float red[4][16 * CT][64];
  __shared__ float zt[32][32 * CT + 1];
  constexpr int QS = 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
  // flat grid, XCD-aware: workgroups are dealt round-robin over the 8 XCDs, so XCD x takes the CONTIGUOUS chunk x of the class-major
  // tile sequence — the 15 row blocks of a class run on one XCD and find D1_c (read twice per workgroup) in that XCD's L2
  int tile;
  {
    const int T = gridDim.x, L = blockIdx.x, xcd = L & 7, slot = L >> 3, q = T >> 3, r = T & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int nrb = (p.n + 31) >> 5;
  const int c = tile / nrb, r0 = (tile - c * nrb) * 32, n = p.n, h = p.h, d = p.d;
  const float* __restrict__ D1 = p.D + (long)c * p.sD + p.oW1;
  const float* __restrict__ Db1 = p.D + (long)c * p.sD + p.ob1;
  const int rl = min(r0 + li, n - 1);            // this lane's row where rows are the lane dimension
  const int hb = 32 * HT * wave;                 // this wave's hidden range [hb, hb + 32 HT)

  // ---- A1m^T tile [hidden x 32 rows] = D1_c^T X^T (+ Db1) * mask : M = hidden (registers), N = row (lanes), K = d
  f32x16 a1[HT];
#pragma unroll
  for (int i = 0; i < HT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) a1[i][r] = 0.f;
  const float* __restrict__ xr = p.X + (long)rl * p.ldx;
  auto load1 = [&](int k0, float (&xb)[QS][4], float (&da)[HT][QS][4]) {
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      const int ks = k0 + 8 * q;
      if (ks < d) {
        const bool whole = ks + 8 <= d;
        small_frag<true>(xr, 1, ks + 4 * kh, d, whole, xb[q]);
#pragma unroll
        for (int i = 0; i < HT; ++i) small_frag<false>(D1 + hb + 32 * i + li, h, ks + 4 * kh, d, whole, da[i][q]);
      }
    }
  };
  auto mma1 = [&](int k0, const float (&xb)[QS][4], const float (&da)[HT][QS][4]) {
#pragma unroll
    for (int q = 0; q < QS; ++q)
      if (k0 + 8 * q < d) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < HT; ++i) a1[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(da[i][q][u], xb[q][u], a1[i], 0, 0, 0);
      }
  };
  {
    float xb0[QS][4], da0[HT][QS][4], xb1[QS][4], da1[HT][QS][4];
    load1(0, xb0, da0);
    for (int k0 = 0; k0 < d; k0 += 16 * QS) {
      if (k0 + 8 * QS < d) load1(k0 + 8 * QS, xb1, da1);
      mma1(k0, xb0, da0);
      if (k0 + 16 * QS < d) load1(k0 + 16 * QS, xb0, da0);
      if (k0 + 8 * QS < d) mma1(k0 + 8 * QS, xb1, da1);
    }
  }
  // Wl fragments of the next product (A operand: m = class, k = hidden = this wave's accumulator rows) and the epilogue operands
  float wa[HT][16][CT];
#pragma unroll
  for (int i = 0; i < HT; ++i)
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        wa[i][s][ct] = p.Wl[(long)(hb + 32 * i + (s & 3) + 8 * (s >> 2) + 4 * kh) * p.C + min(32 * ct + li, p.C - 1)];
#pragma unroll
  for (int i = 0; i < HT; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int hh = hb + 32 * i + 8 * g + 4 * kh;
      const f32x4u bv = *reinterpret_cast<const f32x4u*>(Db1 + hh);
      const f32x4u mv = *reinterpret_cast<const f32x4u*>(p.mask + (long)rl * h + hh);
      a1[i][4 * g + 0] = mv.x > 0.f ? a1[i][4 * g + 0] + bv.x : 0.f;
      a1[i][4 * g + 1] = mv.y > 0.f ? a1[i][4 * g + 1] + bv.y : 0.f;
      a1[i][4 * g + 2] = mv.z > 0.f ? a1[i][4 * g + 2] + bv.z : 0.f;
      a1[i][4 * g + 3] = mv.w > 0.f ? a1[i][4 * g + 3] + bv.w : 0.f;
    }
  // the third product's operands: both k-contiguous (k = hidden); its first round is requested here, ahead of the second product
  const float* __restrict__ hp = p.dHp + (long)c * p.s_dhp + (long)rl * h;
  float* __restrict__ slab = p.slabs + (long)c * p.s_slab;
  const int jt = (d + 31) >> 5;
  auto load3 = [&](const float* dr, int k0, float (&ha)[QS][4], float (&db)[QS][4]) {
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      small_frag<true>(hp, 1, k0 + 8 * q + 4 * kh, h, true, ha[q]);
      small_frag<true>(dr, 1, k0 + 8 * q + 4 * kh, h, true, db[q]);
    }
  };
  float ha0[QS][4], db0[QS][4], ha1[QS][4], db1[QS][4];
  const float* __restrict__ dr = D1 + (long)min(32 * wave + li, d - 1) * h;
  if (wave < jt && p.slabs) load3(dr, 0, ha0, db0);
  // ---- Bc^T [C x 32 rows] = Wl^T A1m^T over this wave's hidden range; the four waves' partials are summed in wave order
  f32x16 bc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) bc[ct][r] = 0.f;
#pragma unroll
  for (int i = 0; i < HT; ++i)
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) bc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[i][s][ct], a1[i][s], bc[ct], 0, 0, 0);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][16 * ct + r][lane] = bc[ct][r];
  __syncthreads();
  for (int q = wave; q < 16 * CT; q += 4) {
    const int ct = q >> 4, r = q & 15;
    const float v = ((red[0][q][lane] + red[1][q][lane]) + red[2][q][lane]) + red[3][q][lane];
    zt[li][32 * ct + (r & 3) + 8 * (r >> 2) + 4 * kh] = v;      // [row][class]
  }
  __syncthreads();
  {
    float* __restrict__ Bc = p.Bc + (long)c * p.s_bc;
    const int rows = min(32, n - r0);
    for (int i = tid; i < rows * p.C; i += 256) {
      const int row = i / p.C, cc = i - row * p.C;
      Bc[(long)(r0 + row) * p.C + cc] = zt[row][cc];
    }
  }
  // ---- dX slab tile [32 rows x 32 features] = dHp_c[rows] D1_c^T
  if (!p.slabs) return;      // (d loss / d X~ not wanted: PGE epochs of the fused step)
  for (int t = wave; t < jt; t += 4) {
    f32x16 ax;
#pragma unroll
    for (int r = 0; r < 16; ++r) ax[r] = 0.f;
    auto mma3 = [&](const float (&ha)[QS][4], const float (&db)[QS][4]) {
#pragma unroll
      for (int q = 0; q < QS; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) ax = __builtin_amdgcn_mfma_f32_32x32x2f32(ha[q][u], db[q][u], ax, 0, 0, 0);
    };
    for (int k0 = 0; k0 < h; k0 += 16 * QS) {      // h is a multiple of 128: whole double rounds
      load3(dr, k0 + 8 * QS, ha1, db1);
      mma3(ha0, db0);
      const bool more = k0 + 16 * QS < h;
      const float* drn = dr;
      if (!more && t + 4 < jt) drn = D1 + (long)min(32 * (t + 4) + li, d - 1) * h;      // the next tile's first round
      if (more || t + 4 < jt) load3(drn, more ? k0 + 16 * QS : 0, ha0, db0);
      mma3(ha1, db1);
      dr = drn;
    }
    const int j = 32 * t + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (row < n && j < d) slab[(long)row * d + j] = ax[r];
    }
  }
}

bool gs_match_class_eligible(int n, int h, int d, int C) { return n >= 1 && d >= 1 && C >= 1 && C <= 64 && (h == 128 || h == 256); }

// engine-internal (gs_engine.hip)
int gs_match_first(const gs_match1_args* a, int Cl, hipStream_t s) {
  if (Cl <= 0) return GS_OK;
  hipLaunchKernelGGL(gs_match_first_kernel, dim3(a->h / 32, Cl, gs_cdiv(a->d, 128)), dim3(256), 0, s, *a);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

int gs_match_second(const gs_match2_args* a, int Cl, hipStream_t s) {
  if (Cl <= 0) return GS_OK;
  const dim3 grid(gs_cdiv(a->n, 32) * Cl);
  const bool c2 = a->C > 32;
  if (a->h == 256) {
    if (c2) hipLaunchKernelGGL((gs_match_second_kernel<2, 2>), grid, dim3(256), 0, s, *a);
    else hipLaunchKernelGGL((gs_match_second_kernel<2, 1>), grid, dim3(256), 0, s, *a);
  } else {
    if (c2) hipLaunchKernelGGL((gs_match_second_kernel<1, 2>), grid, dim3(256), 0, s, *a);
    else hipLaunchKernelGGL((gs_match_second_kernel<1, 1>), grid, dim3(256), 0, s, *a);
  }
  GS_LAUNCH_CHECK();
  return GS_OK;
}
