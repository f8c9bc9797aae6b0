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
// MODE 4 = MODE 3's arithmetic with the B operand split ONCE, when it is written to LDS (round 3): the stage holds A as fp32
// [16][132] and B as three bf16 planes (h, m, l) of [16 k][288] (row stride 576 B = 144 dwords = 16 mod 64: the hardware-transposed
// fragment reads of all 32 lanes of a half hit 64 distinct banks).  A B fragment = 2 x ds_read_b64_tr_b16 per plane (4 consecutive k of
// the lane's column each) instead of 8 ds_read_b32 + 8 three-way splits per tile and wave: the split of a B element is done once by
// the thread that generates it, not by each of the four waves that read it (the first form of mode 3 was VALU-bound: ~1 500 VALU
// cycles beside 1 536 MFMA cycles per 16-edge step).
#define EG4_LDB 288                                    // bf16 elements per k row of a plane
#define EG4_PLANE (EG_BK * EG4_LDB * 2)                // bytes
#define EG4_STAGE (EG_BK * EG_LDA * 4 + 3 * EG4_PLANE) // bytes per stage
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct gs_edge_args {
  const float* X; long ldx; int d;
  const int* col; const float* w; long sK;       // [class][sK] source node ids / edge weights
  const float* H1; long ldm;                     // mask rows: H1[src][n] > 0
  const unsigned* bits; long ldbits;             // BITS: the same predicate as sign bits, 8 words per source row (32 B instead of 1 KiB)
  int bits_edge;                                 // the words are per EDGE SLOT (row class * sK + k), written by the hop-2 gather
  const float* Y; long sYc; int f2;              // Y1 rows: Y[class * sYc + (k / f2) * 256 + n]
  const int* kcount; int kchunk;
  float* out; long sOc, sOs;                     // slabs: out[class * sOc + slab * sOs + m * 256 + n]
  const int2* erow; const float2* ew;            // ENT: per K entry two (row of Y, weight) pairs; `col` holds the entry's source (f2 / w unused)
  const int* plan; int nclass;                   // balanced slabs (gs_edge_plan_kernel): [nclass + 1] first slab of a class, then [nclass] its
                                                 // slab length; slab b of the launch writes out[b * sOs + ...] (null: nsplit slabs of kchunk per class)
};

// ---- balanced K slabs (round 5) ---------------------------------------------------------------------------------------------------
// The edge lists of the classes differ in length (n1[c] * f2 live slots: 7 400 .. 11 800 at ogbn-arxiv) and a launch of (class x
// fixed-length slab) workgroups left 22 % of the chip's 512 workgroup slots empty while the others ran 63 K steps (profiles/r04: the
// longest kernel of the iteration).  The plan cuts the chip's slots over the classes in proportion to their step counts, so that every
// workgroup runs about total_steps / slots steps: plan[c] = first slab of class c (plan[nclass] = slabs in use), plan[nclass + 1 + c] =
// slab length of class c (a multiple of the 16-edge step).  Also writes kcount[c] = n1[c] * f2 (gs_kcount_kernel's job).  One wave.
__global__ __launch_bounds__(64) void gs_edge_plan_kernel(const int* __restrict__ n1, int f2, int nclass, int target, int* __restrict__ kcount,
                                                          int* __restrict__ plan) {
  __shared__ int steps[64], nsl[64];
  const int c = threadIdx.x;
  int st = 0;
  if (c < nclass) {
    const int k = n1[c] * f2;
    kcount[c] = k;
    st = (k + EG_BK - 1) / EG_BK;
  }
  steps[c] = st;
  __syncthreads();
  long total = 0;
  for (int i = 0; i < nclass; ++i) total += steps[i];
  int mine = 0, chunk = EG_BK;
  if (c < nclass && st > 0) {
    mine = (int)(((long)target * st) / (total > 0 ? total : 1));
    if (mine < 1) mine = 1;
    const int per = (st + mine - 1) / mine;          // steps per slab
    mine = (st + per - 1) / per;                     // slabs that are not empty
    chunk = per * EG_BK;
  }
  nsl[c] = mine;
  __syncthreads();
  if (c < nclass) {
    int base = 0;
    for (int i = 0; i < c; ++i) base += nsl[i];
    plan[c] = base;
    plan[nclass + 1 + c] = chunk;
    if (c == nclass - 1) plan[nclass] = base + mine;
  }
}

// sum of a class's slabs in slab order (fixed order: reproducible), 16-byte accesses; a class without live edges gets zeros
__global__ void gs_reduce_slabs_plan_kernel(float* __restrict__ out, long out_stride, const float* __restrict__ slabs, long n4,
                                            long slab_stride, const int* __restrict__ plan) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int b = blockIdx.y;
  const int lo = plan[b], hi = plan[b + 1];
  const float4* sl = reinterpret_cast<const float4*>(slabs) + i;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = lo; k < hi; ++k) {
    const float4 v = sl[(long)k * (slab_stride >> 2)];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  reinterpret_cast<float4*>(out + (long)b * out_stride)[i] = s;
}


// One K step (16 rows of both LDS images of `stage`) of a workgroup's 128 x 256 tile: wave w owns rows [32 w, 32 w + 32) x 256 columns,
// 8 MFMA tiles.  Shared by the edge product and the PGE dW1 product below (same stage layout, same arithmetic).
template <int MODE>
__device__ __forceinline__ void eg_mfma_step(const float* __restrict__ stage, f32x16 (&acc)[8], int wave, int lane) {
  const int li = lane & 31, h = lane >> 5, col_n = li;
  const float* __restrict__ as = stage + 32 * wave + li;
  const float* __restrict__ bs = stage + EG_BK * EG_LDA + col_n;
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
  } else if constexpr (MODE == 4) {
    bf16x8 ah, am, al;                                                     // A: this wave's own 32 feature rows, split here
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float x = as[(8 * h + t) * EG_LDA];
      const __bf16 hb = (__bf16)x;
      const float r1 = x - (float)hb;
      const __bf16 mb = (__bf16)r1;
      ah[t] = hb; am[t] = mb; al[t] = (__bf16)(r1 - (float)mb);
    }
    // B fragment of tile j: k = 8 h .. 8 h + 7 of column 32 j + li = two transposed reads per plane: lane 16 g + 4 q + p supplies the
    // address of (row 8 h + 4 s + q, columns 32 j + 16 (g & 1) + 4 p ..) and receives rows 8 h + 4 s .. + 3 of column 32 j + li
    // (tools/micro/tr16_probe.hip prints the mapping).  EXEC is all ones here (block-uniform control flow).
    const int g1 = (lane >> 4) & 1, q4 = (lane >> 2) & 3, p4 = lane & 3;
    const char* bq = reinterpret_cast<const char*>(stage) + EG_BK * EG_LDA * 4 + (8 * h + q4) * (EG4_LDB * 2) +
                     (16 * g1 + 4 * p4) * 2;
    auto frag = [&](int j, int P) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(bq + P * EG4_PLANE + 64 * j));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(bq + P * EG4_PLANE + 64 * j + 4 * (EG4_LDB * 2)));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(bf16x8, v);
    };
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bf16x8 bh = frag(j, 0), bm = frag(j, 1), bl = frag(j, 2);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);     // small terms first (same order as mode 3)
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
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
}

// ENT (round 6): the K dimension runs over ENTRIES = (distinct source, up to two of its in-edges) instead of edge slots: the x row and the
// mask row of a source are read once per entry, the B row is [H1[src] > 0] * (w0 Y[row0] + w1 Y[row1]) — the source's in-edges summed
// in ascending slot order (fmaf(w1, y1, w0 y0)); a source with more than two in-edges takes several entries (the product is linear).
template <int MODE, bool BITS = false, bool ENT = false>
__global__ __launch_bounds__(256, 2) void gs_edge_gemm_kernel(gs_edge_args p) {
  constexpr int STAGE_F = MODE == 4 ? EG4_STAGE / 4 : EG_STAGE;      // floats per stage
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_F];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  int slab = blockIdx.x, c = blockIdx.y, kchunk = p.kchunk;
  const int m0 = blockIdx.z * 128;
  float* __restrict__ out;
  if (p.plan) {      // balanced slabs: blockIdx.x counts the launch's slabs, class by class
    if ((int)blockIdx.x >= p.plan[p.nclass]) return;
    c = 0;
    while (c + 1 < p.nclass && p.plan[c + 1] <= (int)blockIdx.x) ++c;      // (wave-uniform: scalar loads of <= 64 words)
    slab = (int)blockIdx.x - p.plan[c];
    kchunk = p.plan[p.nclass + 1 + c];
    out = p.out + (long)blockIdx.x * p.sOs;
  } else {
    out = p.out + (long)c * p.sOc + (long)slab * p.sOs;
  }
  const int kc = p.kcount[c];
  const int k_lo = slab * kchunk, k_hi = min(kc, k_lo + kchunk);
  f32x16 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  if (k_lo < k_hi) {      // (block-uniform; an empty slab still writes zeros below: the reduction reads every slab)
    const int* __restrict__ col = p.col + (long)c * p.sK;
    const float* __restrict__ wv = ENT ? nullptr : p.w + (long)c * p.sK;
    const float* __restrict__ Y = p.Y + (long)c * p.sYc;
    const int row = tid >> 4, quad = tid & 15;
    const bool m_ok0 = m0 + 4 * quad < p.d, m_ok1 = m0 + 64 + 4 * quad < p.d;      // d % 4 == 0 (checked by the launcher)
    f32x4 ra[2], rm[BITS ? 1 : 4], ry[4], ry2[ENT ? 4 : 1];
    float rw2 = 0.f;
    unsigned mw[BITS ? 4 : 1];                       // BITS: word 2 i + (quad >> 3) of the source row holds columns 4 quad + 64 i .. + 3
    const int bo = (4 * quad) & 31;
    float rw;
    auto load_raw = [&](int k0) {
      const int k = k0 + row;
      const int kk = min(k, k_hi - 1);
      const int src = col[kk];
      int yrow = 0, yrow2 = 0;
      if constexpr (ENT) {
        const int2 rr = p.erow[(long)c * p.sK + kk];
        const float2 ww = p.ew[(long)c * p.sK + kk];
        yrow = rr.x; yrow2 = rr.y;
        rw = k < k_hi ? ww.x : 0.f;
        rw2 = k < k_hi ? ww.y : 0.f;
      } else {
        rw = k < k_hi ? wv[kk] : 0.f;
        yrow = kk / p.f2;
      }
      const float* xr = p.X + (long)src * p.ldx + m0 + 4 * quad;
      ra[0] = m_ok0 ? *reinterpret_cast<const f32x4*>(xr) : f32x4{0.f, 0.f, 0.f, 0.f};
      ra[1] = m_ok1 ? *reinterpret_cast<const f32x4*>(xr + 64) : f32x4{0.f, 0.f, 0.f, 0.f};
      const float* yr = Y + (long)yrow * 256 + 4 * quad;
      if constexpr (BITS) {
        const unsigned* br = p.bits + (p.bits_edge ? (long)c * p.sK + kk : (long)src) * p.ldbits + (quad >> 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) mw[i] = br[2 * i];
      } else {
        const float* mr = p.H1 + (long)src * p.ldm + 4 * quad;
#pragma unroll
        for (int i = 0; i < 4; ++i) rm[i] = *reinterpret_cast<const f32x4*>(mr + 64 * i);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) ry[i] = *reinterpret_cast<const f32x4*>(yr + 64 * i);
      if constexpr (ENT) {
        const float* yr2 = Y + (long)yrow2 * 256 + 4 * quad;
#pragma unroll
        for (int i = 0; i < 4; ++i) ry2[i] = *reinterpret_cast<const f32x4*>(yr2 + 64 * i);
      }
    };
    auto bval = [&](int i, int e) -> float {
      if constexpr (ENT) return fmaf(rw2, ry2[i][e], rw * ry[i][e]);
      else return rw * ry[i][e];
    };
    auto on = [&](int i, int e) -> bool {
      if constexpr (BITS) return (mw[i] >> (bo + e)) & 1u;
      else return rm[i][e] > 0.f;
    };
    auto write_stage = [&](int st) {
      float* as = smem + st * STAGE_F + row * EG_LDA + 4 * quad;
      *reinterpret_cast<f32x4*>(as) = ra[0];
      *reinterpret_cast<f32x4*>(as + 64) = ra[1];
      if constexpr (MODE == 4) {
        char* bp = reinterpret_cast<char*>(smem + st * STAGE_F) + EG_BK * EG_LDA * 4 + row * (EG4_LDB * 2) + 8 * quad;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bf16x4 bh, bm, bl;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = on(i, e) ? bval(i, e) : 0.f;
            const __bf16 hb = (__bf16)x;
            const float r1 = x - (float)hb;
            const __bf16 mb = (__bf16)r1;
            bh[e] = hb; bm[e] = mb; bl[e] = (__bf16)(r1 - (float)mb);
          }
          *reinterpret_cast<bf16x4*>(bp + 128 * i) = bh;                       // columns 4 quad + 64 i .. + 3 of row k
          *reinterpret_cast<bf16x4*>(bp + EG4_PLANE + 128 * i) = bm;
          *reinterpret_cast<bf16x4*>(bp + 2 * EG4_PLANE + 128 * i) = bl;
        }
      } else {
        float* bs = smem + st * STAGE_F + EG_BK * EG_LDA + row * EG_LDB + 4 * quad;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4 b;
#pragma unroll
          for (int e = 0; e < 4; ++e) b[e] = on(i, e) ? bval(i, e) : 0.f;
          *reinterpret_cast<f32x4*>(bs + 64 * i) = b;
        }
      }
    };
    load_raw(k_lo);
    write_stage(0);
    __syncthreads();
    int st = 0;
    for (int k0 = k_lo; k0 < k_hi; k0 += EG_BK) {
      const bool more = k0 + EG_BK < k_hi;
      if (more) load_raw(k0 + EG_BK);                 // in flight under the MFMAs below
      eg_mfma_step<MODE>(smem + st * STAGE_F, acc, wave, lane);
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

static int g_edge_enabled = 1, g_edge_split_at_write = 1;
// 0 = general tiled kernel; 1 (default) = the dedicated edge kernel, exact split done once at LDS-write time + transposed fragment
// reads (round 3); 2 = the dedicated kernel with the split at fragment-read time (round 2's form of mode 3).  Same bits either way.
extern "C" int gs_gemm_set_edge_kernel(int enable) {
  g_edge_enabled = enable != 0;
  g_edge_split_at_write = enable != 2;
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

// plan / planned launch / planned reduction (engine-internal, csrc/gs_engine.hip): `plan` holds 2 nclass + 1 ints, `max_slabs` bounds
// the slabs the plan may use (target + nclass fit the slab buffer)
int gs_edge_plan(const int* n1, int f2, int nclass, int target, int* kcount, int* plan, hipStream_t s) {
  if (nclass < 1 || nclass > 64 || target < 1) GS_FAIL(GS_ERR_ARG, "gs_edge_plan: 1 <= nclass <= 64, target >= 1");
  hipLaunchKernelGGL(gs_edge_plan_kernel, dim3(1), dim3(64), 0, s, n1, f2, nclass, target, kcount, plan);
  GS_LAUNCH_CHECK();
  return GS_OK;
}
int gs_reduce_slabs_plan(float* out, long out_stride, const float* slabs, long n, long slab_stride, int nbatch, const int* plan,
                         hipStream_t s) {
  if (n <= 0 || nbatch <= 0) return GS_OK;
  hipLaunchKernelGGL(gs_reduce_slabs_plan_kernel, dim3(gs_cdiv(n >> 2, 256), nbatch), dim3(256), 0, s, out, out_stride, slabs, n >> 2,
                     slab_stride, plan);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

static int edge_launch(const gs_gemm_desc& d, const int* plan, int max_slabs, hipStream_t s, const void* erow = nullptr, const void* ew = nullptr);
// ENT form: d.bgen_idx = the entries' sources [class][sGb1], erow / ew = their (row, weight) pairs; planned slabs only
int gs_edge_launch_entries(const gs_gemm_desc& d, const void* erow, const void* ew, const int* plan, int max_slabs, hipStream_t s) {
  return edge_launch(d, plan, max_slabs, s, erow, ew);
}
int gs_edge_launch(const gs_gemm_desc& d, hipStream_t s) { return edge_launch(d, nullptr, 0, s); }
int gs_edge_launch_plan(const gs_gemm_desc& d, const int* plan, int max_slabs, hipStream_t s) { return edge_launch(d, plan, max_slabs, s); }

static int edge_launch(const gs_gemm_desc& d, const int* plan, int max_slabs, hipStream_t s, const void* erow, const void* ew) {
  gs_edge_args a;
  a.erow = (const int2*)erow; a.ew = (const float2*)ew;
  a.X = d.A; a.ldx = d.sAk; a.d = d.M;
  a.col = d.bgen_idx; a.w = d.bgen_w; a.sK = d.sGb1;
  a.H1 = d.bgen_mask; a.ldm = d.bgen_ldm;
  a.bits = d.bgen_bits; a.ldbits = d.bgen_ldbits; a.bits_edge = d.bgen_bits_edge;
  a.Y = d.B; a.sYc = d.sBb1; a.f2 = d.bgen_div;
  a.kcount = d.Kb; a.kchunk = d.kchunk;
  a.out = d.C; a.sOc = d.sCb1; a.sOs = d.sCb2;
  a.plan = plan; a.nclass = d.nb1;
  const dim3 grid(plan ? max_slabs : d.nb2, plan ? 1 : d.nb1, gs_cdiv(d.M, 128)), block(256);
  const bool bits = d.bgen_bits != nullptr && d.bgen_ldbits >= 8;
#define GS_EDGE_LAUNCH(MODE_)                                                                               \
  do {                                                                                                      \
    if (erow && bits) hipLaunchKernelGGL((gs_edge_gemm_kernel<MODE_, true, true>), grid, block, 0, s, a);   \
    else if (erow) hipLaunchKernelGGL((gs_edge_gemm_kernel<MODE_, false, true>), grid, block, 0, s, a);     \
    else if (bits) hipLaunchKernelGGL((gs_edge_gemm_kernel<MODE_, true>), grid, block, 0, s, a);            \
    else hipLaunchKernelGGL((gs_edge_gemm_kernel<MODE_, false>), grid, block, 0, s, a);                     \
  } while (0)
  if (d.prec == 0) GS_EDGE_LAUNCH(0);
  else if (d.prec == 1) GS_EDGE_LAUNCH(1);
  else if (d.prec == 2) GS_EDGE_LAUNCH(2);
  else if (g_edge_split_at_write) GS_EDGE_LAUNCH(4);
  else GS_EDGE_LAUNCH(3);
#undef GS_EDGE_LAUNCH
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// ---- PGE epochs: dW1 | db1 on GENERATED operands (parametrized_adj.py:57-64, the backward of layer 2) -----------------------------
//
//     dW1[o][j] = sum over pair rows p = (r, c) of  dE2[p][o] * A1[p][j],        db1[o] = sum_p dE2[p][o]
//
// K = n'^2 pair rows (206 116 at ogbn-arxiv), M = N = 256: 27 GFLOP — the same shape as the edge product above (a K-long sum of
// outer products of two generated 128- / 256-wide rows), so it runs on the same stage layout and MFMA step.  Neither operand
// exists in memory:
//   * dE2[p][o] is formed from E2[p][o] (the only HBM stream, read once per 128-column half) with the SAME arithmetic and the same
//     constants table as the fused-marginals product (gs_panel.hip, panel_transform_a<AGEN>), so both kernels see the same dE2;
//   * A1[(r, c)][j] = max(Us[c][j] + Vs[r][j], 0) with the BatchNorm-1 affine folded into the two n' x 256 terms
//     (gs_pge_scale_uv_kernel below: Us = s U + t, Vs = s V, s = g1 is1, t = s (b0 - mu1) + be1) — L2-resident, no constants.
// A K step = 16 consecutive c of ONE r (the last step of an r is masked: n' = 454 -> 29 steps, 2 % idle rows), so r and the first c
// are wave-uniform and the Vs row is read once per step.  Workgroup = (K slab, 128-column half of dE2); the slabs carry
// [256][256] of dW1 followed by [256] of db1 and are reduced in a fixed order by gs_reduce_slabs_f32.
// Round 2 materialised dE2 (211 MB written + read three times) and A1 (211 MB) for a general tiled product that fetched both
// twice (895 MB per launch).
struct gs_dw1_args {
  const float* E2; long lde;          // this rank's pair rows: row (r - r_lo) n' + c
  const float* de3;                   // [rows] gradient of the layer-3 output of the same rows
  const float* tab;                   // [256][8] rows of gs_panel_gen_row
  const float* Us; const float* Vs;   // [n'][256]; Vs already offset to this rank's first r
  int n, spr, stages, q;              // n', steps per r = ceil(n' / 16), nr * spr, steps per slab
  float* out; long sOs;               // slabs: out[slab * sOs + m * 256 + j], db1 at out[slab * sOs + 65536 + m]
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void gs_dw1_gen_kernel(gs_dw1_args p) {
  constexpr int STAGE_F = MODE == 4 ? EG4_STAGE / 4 : EG_STAGE;      // floats per stage
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_F];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  const int slab = blockIdx.x, m0 = blockIdx.z * 128;
  const int s_lo = slab * p.q, s_hi = min(p.stages, s_lo + p.q);
  float* __restrict__ out = p.out + (long)slab * p.sOs;
  f32x16 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // generation: A side (dE2 half): thread = 4 columns m0 + 4 cgA .. of rows 2 rpA, 2 rpA + 1 of the step (a half-wave reads 512
  // contiguous bytes of an E2 row); B side (A1): thread = 4 columns 4 lane .. of rows 4 wave .. + 3 (a wave reads a whole Us row)
  const int cgA = tid & 31, rpA = tid >> 5;
  f32x4 dsum = {0.f, 0.f, 0.f, 0.f};
  if (s_lo < s_hi) {      // (block-uniform)
    f32x4 k_is2, k_nm, k_g2, k_be2, k_c4, k_c5, k_c6;
    {
      const float* t = p.tab + (long)(m0 + 4 * cgA) * 8;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(t + 8 * e), hi = *reinterpret_cast<const f32x4*>(t + 8 * e + 4);
        k_is2[e] = lo[0]; k_nm[e] = lo[1]; k_g2[e] = lo[2]; k_be2[e] = lo[3];
        k_c4[e] = hi[0]; k_c5[e] = hi[1]; k_c6[e] = hi[2];
      }
    }
    // every row of step s + 1 is requested one MFMA block ahead (a second E2 register set with two blocks of lead measured the same:
    // the loop is not waiting for memory)
    int r = s_lo / p.spr, c0 = 16 * (s_lo - r * p.spr), nvalid = 0;
    f32x4 ra[2], ru[4], rv;
    float rd[2];
    auto load_raw = [&](int r_, int c0_) {
      nvalid = p.n - c0_;
      const long row0 = (long)r_ * p.n;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const long row = row0 + min(c0_ + 2 * rpA + i, p.n - 1);
        ra[i] = *reinterpret_cast<const f32x4*>(p.E2 + row * p.lde + m0 + 4 * cgA);
        rd[i] = p.de3[row];
      }
      rv = *reinterpret_cast<const f32x4*>(p.Vs + (long)r_ * 256 + 4 * lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        ru[i] = *reinterpret_cast<const f32x4*>(p.Us + (long)min(c0_ + 4 * wave + i, p.n - 1) * 256 + 4 * lane);
    };
    auto write_stage = [&](int st) {
      float* base = smem + st * STAGE_F;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 2 * rpA + i;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = fmaf(ra[i][e], k_is2[e], k_nm[e]);
          const float y = fmaf(k_g2[e], xh, k_be2[e]);
          const float sel = y > 0.f ? rd[i] * k_c4[e] : 0.f;
          const float val = fmaf(-xh, k_c6[e], sel - k_c5[e]);
          v[e] = row < nvalid ? val : 0.f;
          dsum[e] += v[e];
        }
        *reinterpret_cast<f32x4*>(base + row * EG_LDA + 4 * cgA) = v;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 4 * wave + i;
        f32x4 b;
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = fmaxf(ru[i][e] + rv[e], 0.f);
        if constexpr (MODE == 4) {
          char* bp = reinterpret_cast<char*>(base) + EG_BK * EG_LDA * 4 + row * (EG4_LDB * 2) + 8 * lane;
          bf16x4 bh, bm, bl;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = b[e];
            const __bf16 hb = (__bf16)x;
            const float r1 = x - (float)hb;
            const __bf16 mb = (__bf16)r1;
            bh[e] = hb; bm[e] = mb; bl[e] = (__bf16)(r1 - (float)mb);
          }
          *reinterpret_cast<bf16x4*>(bp) = bh;
          *reinterpret_cast<bf16x4*>(bp + EG4_PLANE) = bm;
          *reinterpret_cast<bf16x4*>(bp + 2 * EG4_PLANE) = bl;
        } else {
          *reinterpret_cast<f32x4*>(base + EG_BK * EG_LDA + row * EG_LDB + 4 * lane) = b;
        }
      }
    };
    load_raw(r, c0);
    write_stage(0);
    __syncthreads();
    int st = 0;
    for (int sidx = s_lo; sidx < s_hi; ++sidx) {
      const bool more = sidx + 1 < s_hi;
      if (more) {
        c0 += 16;
        if (c0 >= 16 * p.spr) { c0 = 0; ++r; }
        load_raw(r, c0);
      }
      eg_mfma_step<MODE>(smem + st * STAGE_F, acc, wave, lane);
      if (more) write_stage(st ^ 1);
      __syncthreads();
      st ^= 1;
    }
  }
  const int rbase = m0 + 32 * wave + 4 * h;
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(long)(rbase + (r & 3) + 8 * (r >> 2)) * 256 + 32 * j + li] = acc[j][r];
  // db1: the 8 row pairs' column sums, fixed order (the stages are free: the loop ended on a barrier)
#pragma unroll
  for (int e = 0; e < 4; ++e) smem[rpA * 128 + 4 * cgA + e] = dsum[e];
  __syncthreads();
  if (tid < 128) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += smem[g * 128 + tid];
    out[65536 + m0 + tid] = t;
  }
}

// Us = s U + t, Vs = s V per column (s = g1 is1, t = s (b0 - mu1) + be1 from the [Ph][8] rows {b0, mu1, is1, g1, be1, ...} the
// BatchNorm-1 statistics kernel leaves): relu(Us[c] + Vs[r]) = the layer-1 activation of pair (r, c) up to fp32 rounding
__global__ void gs_pge_scale_uv_kernel(const float* __restrict__ U, const float* __restrict__ V, const float* __restrict__ tab,
                                       long total, int Ph, float* __restrict__ Us, float* __restrict__ Vs) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float* t = tab + (i % Ph) * 8;
  const float sc = t[3] * t[2];
  Us[i] = fmaf(sc, U[i], fmaf(sc, t[0] - t[1], t[4]));
  Vs[i] = sc * V[i];
}

// engine-internal (gs_engine.hip): slabs must hold nsplit * (65536 + 256) floats, uv 2 n' 256 floats; *nsplit_out = slabs written.
// GS_ERR_UNSUPPORTED when the shape is not covered (the caller then materialises both operands).
// engine-internal: Us, Vs of the current (U, V, BatchNorm-1 statistics) — see gs_pge_scale_uv_kernel
int gs_pge_scale_uv(const float* U, const float* V, const float* a1_tab, long total, int Ph, float* Us, float* Vs, hipStream_t s) {
  hipLaunchKernelGGL(gs_pge_scale_uv_kernel, dim3(gs_cdiv(total, 256)), dim3(256), 0, s, U, V, a1_tab, total, Ph, Us, Vs);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// (uv_ready: [2][n][Ph] Us | Vs already computed by the forward — then `uv` is not written)
int gs_dw1_generated(const float* E2, const float* de3, const float* gen_tab, const float* U, const float* V, const float* a1_tab,
                     int n, int r_lo, int nr, int Ph, int prec, float* uv, float* slabs, long slabs_cap, int* nsplit_out,
                     const float* uv_ready, hipStream_t s) {
  if (!g_edge_enabled || Ph != 256 || n < 16 || nr < 1 || prec < 0 || prec > 3) return GS_ERR_UNSUPPORTED;
  if (((uintptr_t)E2 | (uintptr_t)uv | (uintptr_t)gen_tab) & 15) return GS_ERR_UNSUPPORTED;
  gs_dw1_args a;
  a.E2 = E2; a.lde = Ph; a.de3 = de3; a.tab = gen_tab;
  const float* uvp = uv_ready ? uv_ready : uv;
  a.Us = uvp; a.Vs = uvp + (long)n * Ph + (long)r_lo * Ph;
  a.n = n; a.spr = gs_cdiv(n, EG_BK); a.stages = nr * a.spr;
  // 512 workgroup slots (2 per CU) = 256 slabs x 2 halves; short sums (a rank's shard): at least 16 steps per slab
  int nsplit = a.stages >= 256 * 16 ? 256 : gs_cdiv(a.stages, 16);
  a.q = gs_cdiv(a.stages, nsplit);
  nsplit = gs_cdiv(a.stages, a.q);
  a.sOs = 65536 + 256;
  if ((long)nsplit * a.sOs > slabs_cap) return GS_ERR_UNSUPPORTED;
  a.out = slabs;
  const long total = (long)n * Ph;
  if (!uv_ready) hipLaunchKernelGGL(gs_pge_scale_uv_kernel, dim3(gs_cdiv(total, 256)), dim3(256), 0, s, U, V, a1_tab, total, Ph, uv, uv + total);
  const dim3 grid(nsplit, 1, 2), block(256);
  if (prec == 0) hipLaunchKernelGGL((gs_dw1_gen_kernel<0>), grid, block, 0, s, a);
  else if (prec == 1) hipLaunchKernelGGL((gs_dw1_gen_kernel<1>), grid, block, 0, s, a);
  else if (prec == 2) hipLaunchKernelGGL((gs_dw1_gen_kernel<2>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((gs_dw1_gen_kernel<4>), grid, block, 0, s, a);
  GS_LAUNCH_CHECK();
  *nsplit_out = nsplit;
  return GS_OK;
}
