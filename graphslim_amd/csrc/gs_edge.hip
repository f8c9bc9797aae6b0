// The real-side first-layer gradient of SGC (T = 2) as ONE edge-wise product per class (gcond_base.py:223 autograd through
// relu and the hop-2 block):
//
//     gW1_c[m][n] = sum over sampled edges e of class c:  X[src(e)][m] * ( w_e * [H1[src(e)][n] > 0] * Y1_c[row(e)][n] )
//
// m < d (features), n < h = 256 (hidden units), e < kcount[c] <= cap1 * f2 (14 080 at ogbn-arxiv): 37 GFLOP per iteration,
// the longest kernel of the real-side branch, which is the critical path of the iteration's first half (timeline r02).
// Dedicated kernel (the general tiled kernel ran this product at 37 % of the fp32 matrix peak):
//   * one workgroup = one (class, K slab, 128-feature tile) over ALL 256 hidden units: the X rows and the mask / Y1 rows of
//     an edge are gathered once (the 128x128 tile gathered X twice and generated B per N tile);
//   * wave w owns features [32 w, 32 w + 32) x 256 hidden units: 8 MFMA tiles, 128 accumulator registers, 2 workgroups / CU;
//   * K step = 16 edges: raw rows -> registers one step ahead (16 B per lane, whole 512 B / 1 KiB rows per 32 / 64 lanes),
//     B generated in registers, both operands written to a double-buffered LDS image [k][m] / [k][n] (row pad 4 floats:
//     conflict-free ds_write_b128 and ds_read_b32), ONE barrier per K step;
//   * partial products go to slabs [class][slab][d][h] reduced in a fixed order by gs_reduce_slabs_batched_f32 (no atomics).
// Operand precision as in gs_panel.hip: 0 exact fp32 (v_mfma_f32_32x32x2_f32), 1 split bf16 x3, 2 bf16, 3 the EXACT three-way
// bf16 split with six products (operands converted when the fragments are read; fp32 accumulation).
#include "gs_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define EG_BK 16
#define EG_LDA 132
#define EG_LDB 260
#define EG_STAGE (EG_BK * (EG_LDA + EG_LDB))   // floats per stage

struct gs_edge_args {
  const float* X; long ldx; int d;
  const int* col; const float* w; long sK;       // [class][sK] source node ids / edge weights
  const float* H1; long ldm;                     // mask rows: H1[src][n] > 0
  const float* Y; long sYc; int f2;              // Y1 rows: Y[class * sYc + (k / f2) * 256 + n]
  const int* kcount; int kchunk;
  float* out; long sOc, sOs;                     // slabs: out[class * sOc + slab * sOs + m * 256 + n]
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void gs_edge_gemm_kernel(gs_edge_args p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * EG_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  const int slab = blockIdx.x, c = blockIdx.y, m0 = blockIdx.z * 128;
  const int kc = p.kcount[c];
  const int k_lo = slab * p.kchunk, k_hi = min(kc, k_lo + p.kchunk);
  float* __restrict__ out = p.out + (long)c * p.sOc + (long)slab * p.sOs;
  const int col_n = li;     // this lane's column inside a 32-wide N tile
  f32x16 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  if (k_lo < k_hi) {      // (block-uniform; an empty slab still writes zeros below: the reduction reads every slab)
    const int* __restrict__ col = p.col + (long)c * p.sK;
    const float* __restrict__ wv = p.w + (long)c * p.sK;
    const float* __restrict__ Y = p.Y + (long)c * p.sYc;
    const int row = tid >> 4, quad = tid & 15;
    const bool m_ok0 = m0 + 4 * quad < p.d, m_ok1 = m0 + 64 + 4 * quad < p.d;      // d % 4 == 0 (checked by the launcher)
    f32x4 ra[2], rm[4], ry[4];
    float rw;
    auto load_raw = [&](int k0) {
      const int k = k0 + row;
      const int kk = min(k, k_hi - 1);
      const int src = col[kk];
      rw = k < k_hi ? wv[kk] : 0.f;
      const float* xr = p.X + (long)src * p.ldx + m0 + 4 * quad;
      ra[0] = m_ok0 ? *reinterpret_cast<const f32x4*>(xr) : f32x4{0.f, 0.f, 0.f, 0.f};
      ra[1] = m_ok1 ? *reinterpret_cast<const f32x4*>(xr + 64) : f32x4{0.f, 0.f, 0.f, 0.f};
      const float* mr = p.H1 + (long)src * p.ldm + 4 * quad;
      const float* yr = Y + (long)(kk / p.f2) * 256 + 4 * quad;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rm[i] = *reinterpret_cast<const f32x4*>(mr + 64 * i);
        ry[i] = *reinterpret_cast<const f32x4*>(yr + 64 * i);
      }
    };
    auto write_stage = [&](int st) {
      float* as = smem + st * EG_STAGE + row * EG_LDA + 4 * quad;
      float* bs = smem + st * EG_STAGE + EG_BK * EG_LDA + row * EG_LDB + 4 * quad;
      *reinterpret_cast<f32x4*>(as) = ra[0];
      *reinterpret_cast<f32x4*>(as + 64) = ra[1];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 b;
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = rm[i][e] > 0.f ? rw * ry[i][e] : 0.f;
        *reinterpret_cast<f32x4*>(bs + 64 * i) = b;
      }
    };
    load_raw(k_lo);
    write_stage(0);
    __syncthreads();
    int st = 0;
    for (int k0 = k_lo; k0 < k_hi; k0 += EG_BK) {
      const bool more = k0 + EG_BK < k_hi;
      if (more) load_raw(k0 + EG_BK);                 // in flight under the MFMAs below
      const float* __restrict__ as = smem + st * EG_STAGE + 32 * wave + li;
      const float* __restrict__ bs = smem + st * EG_STAGE + EG_BK * EG_LDA + col_n;
      if constexpr (MODE == 0) {
        float a[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) a[t] = as[(2 * t + h) * EG_LDA];          // MFMA t consumes k = 2 t + (lane >> 5)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float b[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) b[t] = bs[(2 * t + h) * EG_LDB + 32 * j];
#pragma unroll
          for (int t = 0; t < 8; ++t) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc[j], 0, 0, 0);
        }
      } else if constexpr (MODE == 3) {
        bf16x8 ah, am, al;                                                     // x = h + m + l exactly (8 + 8 + 8 bits)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float x = as[(8 * h + t) * EG_LDA];
          const __bf16 hb = (__bf16)x;
          const float r1 = x - (float)hb;
          const __bf16 mb = (__bf16)r1;
          ah[t] = hb; am[t] = mb; al[t] = (__bf16)(r1 - (float)mb);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          bf16x8 bh, bm, bl;
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const float x = bs[(8 * h + t) * EG_LDB + 32 * j];
            const __bf16 hb = (__bf16)x;
            const float r1 = x - (float)hb;
            const __bf16 mb = (__bf16)r1;
            bh[t] = hb; bm[t] = mb; bl[t] = (__bf16)(r1 - (float)mb);
          }
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);     // small terms first
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
        }
      } else {
        bf16x8 ah, al;                                                         // lane half h supplies k = 8 h .. 8 h + 7
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float x = as[(8 * h + t) * EG_LDA];
          const __bf16 hb = (__bf16)x;
          ah[t] = hb;
          if constexpr (MODE == 1) al[t] = (__bf16)(x - (float)hb);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          bf16x8 bh, bl;
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const float x = bs[(8 * h + t) * EG_LDB + 32 * j];
            const __bf16 hb = (__bf16)x;
            bh[t] = hb;
            if constexpr (MODE == 1) bl[t] = (__bf16)(x - (float)hb);
          }
          if constexpr (MODE == 1) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
          }
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
        }
      }
      if (more) write_stage(st ^ 1);
      __syncthreads();
      st ^= 1;
    }
  }
  // slab tile: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); rows = features, cols = hidden units
  const int rbase = m0 + 32 * wave + 4 * h;
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = rbase + (r & 3) + 8 * (r >> 2);
      if (m < p.d) out[(long)m * 256 + 32 * j + li] = acc[j][r];
    }
}

static int g_edge_enabled = 1;
extern "C" int gs_gemm_set_edge_kernel(int enable) {
  g_edge_enabled = enable != 0;
  return GS_OK;
}

// the generated-B descriptors gs_gcond_stage_real_grads builds: A = X gathered along K, split K, one batch per class
bool gs_edge_eligible(const gs_gemm_desc& p) {
  if (!g_edge_enabled || !p.bgen_mask || !p.bgen_idx || !p.bgen_w || !p.gatherA_k) return false;
  if (p.N != 256 || p.sAm != 1 || p.sBn != 1 || p.sBk != 256 || p.ldc != 256 || p.bgen_ldm < 256) return false;
  if ((p.M & 3) || (p.sAk & 3) || (p.bgen_ldm & 3) || p.kchunk <= 0 || (p.kchunk % EG_BK) || !p.Kb) return false;
  if (p.bgen_div < 1 || (p.sBb1 & 3)) return false;            // the kernel divides by f2 and reads Y rows with 16-byte loads
  if (p.gatherA_k != p.bgen_idx) return false;                 // one source id per edge feeds both the X row and the mask row
  if (p.alpha != 1.f || p.beta != 0.f || p.bias || p.relu || p.mask || p.colstat || p.offA || p.offB || p.offC || p.Mb) return false;
  if (p.gatherA_m || p.gatherB_k || p.sAb1 != 0 || p.sAb2 != 0 || p.sBb2 != 0) return false;
  if ((((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.bgen_mask) & 15) || p.prec < 0 || p.prec > 3) return false;
  return true;
}

int gs_edge_launch(const gs_gemm_desc& d, hipStream_t s) {
  gs_edge_args a;
  a.X = d.A; a.ldx = d.sAk; a.d = d.M;
  a.col = d.bgen_idx; a.w = d.bgen_w; a.sK = d.sGb1;
  a.H1 = d.bgen_mask; a.ldm = d.bgen_ldm;
  a.Y = d.B; a.sYc = d.sBb1; a.f2 = d.bgen_div;
  a.kcount = d.Kb; a.kchunk = d.kchunk;
  a.out = d.C; a.sOc = d.sCb1; a.sOs = d.sCb2;
  const dim3 grid(d.nb2, d.nb1, gs_cdiv(d.M, 128)), block(256);
  if (d.prec == 0) hipLaunchKernelGGL((gs_edge_gemm_kernel<0>), grid, block, 0, s, a);
  else if (d.prec == 1) hipLaunchKernelGGL((gs_edge_gemm_kernel<1>), grid, block, 0, s, a);
  else if (d.prec == 2) hipLaunchKernelGGL((gs_edge_gemm_kernel<2>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((gs_edge_gemm_kernel<3>), grid, block, 0, s, a);
  GS_LAUNCH_CHECK();
  return GS_OK;
}
