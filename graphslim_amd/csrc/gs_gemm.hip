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
#include "gs_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GS_BK 16
#define GS_PAD 4

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gs_gemm_f32_kernel(gs_gemm_desc p) {
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
  constexpr int LA = BM * GS_BK / 256, LB = BN * GS_BK / 256;  // elements per thread per tile
  __shared__ float As[GS_BK][BM + GS_PAD];
  __shared__ float Bs[GS_BK][BN + GS_PAD];

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
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (m0 >= M) return;

  const float* __restrict__ A = p.A + (long)b1 * p.sAb1 + (p.kchunk > 0 ? 0 : (long)b2 * p.sAb2) + (p.offA ? p.offA[b1] : 0);
  const float* __restrict__ B = p.B + (long)b1 * p.sBb1 + (p.kchunk > 0 ? 0 : (long)b2 * p.sBb2) + (p.offB ? p.offB[b1] : 0);
  float* __restrict__ C = p.C + (long)b1 * p.sCb1 + (long)b2 * p.sCb2 + (p.offC ? p.offC[b1] : 0);
  const int* gAm = p.gatherA_m ? p.gatherA_m + (long)b1 * p.sGb1 : nullptr;
  const int* gAk = p.gatherA_k ? p.gatherA_k + (long)b1 * p.sGb1 : nullptr;
  const int* gBk = p.gatherB_k ? p.gatherB_k + (long)b1 * p.sGb1 : nullptr;

  // thread -> tile element mapping: contiguous lanes along the memory-contiguous dimension
  const bool a_kfast = (p.sAk == 1 && !gAk);   // A rows contiguous in k
  const bool b_kfast = (p.sBk == 1 && !gBk);   // B columns contiguous in k  (B given as [N][K])
  float ra[LA], rb[LB];

  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      int e = tid + i * 256, m, k;
      if (a_kfast) { k = e % GS_BK; m = e / GS_BK; } else { m = e % BM; k = e / BM; }
      int gm = m0 + m, gk = k0 + k;
      float v = 0.f;
      if (gm < M && gk < k_hi) {
        long im = gAm ? (long)gAm[gm] : (long)gm;
        long ik = gAk ? (long)gAk[gk] : (long)gk;
        v = A[im * p.sAm + ik * p.sAk];
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      int e = tid + i * 256, n, k;
      if (b_kfast) { k = e % GS_BK; n = e / GS_BK; } else { n = e % BN; k = e / BN; }
      int gn = n0 + n, gk = k0 + k;
      float v = 0.f;
      if (gn < p.N && gk < k_hi) {
        long ik = gBk ? (long)gBk[gk] : (long)gk;
        v = B[ik * p.sBk + (long)gn * p.sBn];
      }
      rb[i] = v;
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      int e = tid + i * 256, m, k;
      if (a_kfast) { k = e % GS_BK; m = e / GS_BK; } else { m = e % BM; k = e / BM; }
      As[k][m] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      int e = tid + i * 256, n, k;
      if (b_kfast) { k = e % GS_BK; n = e / GS_BK; } else { n = e % BN; k = e / BN; }
      Bs[k][n] = rb[i];
    }
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

  if (k_lo < k_hi) {
    load_tiles(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += GS_BK) {
      store_tiles();
      __syncthreads();
      if (k0 + GS_BK < k_hi) load_tiles(k0 + GS_BK);   // in flight under the MFMAs below
#pragma unroll
      for (int s = 0; s < GS_BK / 2; ++s) {
        float a[TM], bb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[2 * s + kh][wm0 + i * 32 + li];
#pragma unroll
        for (int j = 0; j < TN; ++j) bb[j] = Bs[2 * s + kh][wn0 + j * 32 + li];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
  }

  // epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        int col = n0 + wn0 + j * 32 + li;
        if (row < M && col < p.N) {
          float v = p.alpha * acc[i][j][r];
          if (p.bias) v += p.bias[(long)b1 * p.sbias_b1 + col];
          if (p.relu) v = fmaxf(v, 0.f);
          if (p.mask) v = (p.mask[(long)row * p.ldmask + col] > 0.f) ? v : 0.f;
          long ci = (long)row * p.ldc + col;
          if (p.beta != 0.f) v += p.beta * C[ci];
          C[ci] = v;
        }
      }
}

__global__ void gs_reduce_slabs_kernel(float* __restrict__ out, const float* __restrict__ slabs, long n, int ns,
                                       long slab_stride, float beta) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < ns; ++k) s += slabs[(long)k * slab_stride + i];   // fixed order: reproducible
  out[i] = (beta != 0.f ? beta * out[i] : 0.f) + s;
}

extern "C" int gs_reduce_slabs_f32(float* out, const float* slabs, long n, int n_slabs, long slab_stride, float beta,
                                   gs_stream_t stream) {
  if (n <= 0) return GS_OK;
  if (!out || !slabs || n_slabs < 1) GS_FAIL(GS_ERR_ARG, "gs_reduce_slabs_f32: bad argument");
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

extern "C" int gs_reduce_slabs_batched_f32(float* out, long out_stride, const float* slabs, long n, int n_slabs,
                                           long slab_stride, int nbatch, float beta, gs_stream_t stream) {
  if (n <= 0 || nbatch <= 0) return GS_OK;
  if (!out || !slabs || n_slabs < 1) GS_FAIL(GS_ERR_ARG, "gs_reduce_slabs_batched_f32: bad argument");
  hipLaunchKernelGGL(gs_reduce_slabs_batched_kernel, dim3(gs_cdiv(n, 256), nbatch), dim3(256), 0, (hipStream_t)stream, out,
                     out_stride, slabs, n, n_slabs, slab_stride, beta);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

template <int BM, int BN, int WM, int WN>
static int launch_gemm(const gs_gemm_desc& p, int nbatch, hipStream_t s) {
  dim3 grid(gs_cdiv(p.N, BN), gs_cdiv(p.M, BM), nbatch);
  hipLaunchKernelGGL((gs_gemm_f32_kernel<BM, BN, WM, WN>), grid, dim3(256), 0, s, p);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_gemm_f32(const gs_gemm_desc* d, gs_stream_t stream) {
  if (!d || !d->A || !d->B || !d->C) GS_FAIL(GS_ERR_ARG, "gs_gemm_f32: null operand");
  gs_gemm_desc p = *d;
  if (p.M <= 0 || p.N <= 0) return GS_OK;
  if (p.K < 0) GS_FAIL(GS_ERR_ARG, "gs_gemm_f32: negative K");
  if (p.nb1 < 1) p.nb1 = 1;
  if (p.nb2 < 1) p.nb2 = 1;
  const int nbatch = p.nb1 * p.nb2;
  if ((long)nbatch > 65535) GS_FAIL(GS_ERR_ARG, "gs_gemm_f32: too many batches (%d)", nbatch);
  hipStream_t s = (hipStream_t)stream;
  // tile choice: fill the 256 CUs; narrow outputs get the tall/flat tiles
  if (p.N <= 32) return launch_gemm<128, 32, 32, 32>(p, nbatch, s);
  if (p.M <= 32) return launch_gemm<32, 128, 32, 32>(p, nbatch, s);
  long tiles128 = (long)gs_cdiv(p.M, 128) * gs_cdiv(p.N, 128) * nbatch;
  if (p.M <= 64 || p.N <= 64 || tiles128 < 192) return launch_gemm<64, 64, 32, 32>(p, nbatch, s);
  return launch_gemm<128, 128, 64, 64>(p, nbatch, s);
}
