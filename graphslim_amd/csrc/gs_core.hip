// Stateless building blocks: error plumbing, CSR SpMM (wave-per-row gather), row gather,
// dense GCN normalisation fwd/bwd, fused Adam.
#include <stdarg.h>

#include "gs_common.h"

static thread_local char g_err[512] = "";

void gs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* gs_last_error(void) { return g_err; }
extern "C" int gs_version(void) { return 400; }   // 100 * round + revision: round 4 added the persistent 256-row exact-split kernels, the register-chained class products, the replicated-PGE layout; knobs moved to gslim_internal.h

// ------------------------------------------------------------------------------------------------
// CSR SpMM:  Y[i,:] = sum_e val[e] * X[col[e],:]
// One row per LPR-lane group (LPR = 64 for wide F, down to 2 for F <= 8), VEC floats per lane per
// access (16-B loads when the operands allow it).  Each group walks its row's edges four at a time so
// four gathered rows are in flight per lane; col/val loads are group-uniform (one transaction).
// HBM roofline: bytes = 4(M+1) + 8 nnz + 4 F (distinct source rows) + 4 F M.
// ------------------------------------------------------------------------------------------------
template <int VEC>
struct VecT;
template <>
struct VecT<1> {
  typedef float T;
};
template <>
struct VecT<4> {
  typedef float4 T;
};

__device__ __forceinline__ void fma_acc(float& a, float w, float x) { a = fmaf(w, x, a); }
__device__ __forceinline__ void fma_acc(float4& a, float w, float4 x) {
  a.x = fmaf(w, x.x, a.x);
  a.y = fmaf(w, x.y, a.y);
  a.z = fmaf(w, x.z, a.z);
  a.w = fmaf(w, x.w, a.w);
}
__device__ __forceinline__ void zero(float& a) { a = 0.f; }
__device__ __forceinline__ void zero(float4& a) { a = make_float4(0.f, 0.f, 0.f, 0.f); }

template <int VEC, int LPR>
__global__ __launch_bounds__(256) void gs_spmm_csr_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                          const float* __restrict__ val, int n_rows,
                                                          const float* __restrict__ X, long ldx, int F,
                                                          float* __restrict__ Y, long ldy) {
  typedef typename VecT<VEC>::T V;
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const long wave_g = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const long row = wave_g * RPW + sub;
  const bool live = row < n_rows;
  const int e0 = live ? rowptr[row] : 0, e1 = live ? rowptr[row + 1] : 0;
  // the widest row of the wave decides the trip count, so the shuffles below are executed by all 64 lanes
  int span = e1 - e0;
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) span = max(span, __shfl_xor(span, o, 64));
  const int mine = e1 - e0;
  for (int ch = 0; ch * LPR * VEC < F; ++ch) {        // same trip count in every lane: the shuffles need all 64 lanes
    const int c0 = (ch * LPR + l) * VEC;
    const bool colok = c0 < F;
    V acc;
    zero(acc);
    for (int base = 0; base < span; base += LPR) {
      // one coalesced load of up to LPR (col, val) pairs per row group; edges are then broadcast by shuffle, so the
      // gathers of a chunk do not wait on a second dependent index load
      const int my = e0 + base + l;
      const int cj = (my < e1) ? col[my] : 0;
      const float wj = (my < e1) ? val[my] : 0.f;
      const int cnt = min(LPR, span - base);
      int j = 0;
      for (; j + 4 <= cnt; j += 4) {
        const int s0 = sub * LPR + j;
        const int j0 = __shfl(cj, s0, 64), j1 = __shfl(cj, s0 + 1, 64), j2 = __shfl(cj, s0 + 2, 64), j3 = __shfl(cj, s0 + 3, 64);
        const float w0 = __shfl(wj, s0, 64), w1 = __shfl(wj, s0 + 1, 64), w2 = __shfl(wj, s0 + 2, 64), w3 = __shfl(wj, s0 + 3, 64);
        if (colok) {
          V x0 = *reinterpret_cast<const V*>(X + (long)j0 * ldx + c0);
          V x1 = *reinterpret_cast<const V*>(X + (long)j1 * ldx + c0);
          V x2 = *reinterpret_cast<const V*>(X + (long)j2 * ldx + c0);
          V x3 = *reinterpret_cast<const V*>(X + (long)j3 * ldx + c0);
          V z;
          zero(z);   // slots past this row's own edges read row 0 with weight 0: drop the value so 0*inf cannot appear
          fma_acc(acc, w0, base + j + 0 < mine ? x0 : z);
          fma_acc(acc, w1, base + j + 1 < mine ? x1 : z);
          fma_acc(acc, w2, base + j + 2 < mine ? x2 : z);
          fma_acc(acc, w3, base + j + 3 < mine ? x3 : z);
        }
      }
      for (; j < cnt; ++j) {
        const int jj = __shfl(cj, sub * LPR + j, 64);
        const float ww = __shfl(wj, sub * LPR + j, 64);
        if (colok) {
          V x = *reinterpret_cast<const V*>(X + (long)jj * ldx + c0);
          V z;
          zero(z);
          fma_acc(acc, ww, base + j < mine ? x : z);
        }
      }
    }
    if (live && colok) *reinterpret_cast<V*>(Y + row * ldy + c0) = acc;
  }
}

template <int VEC, int LPR>
static void launch_spmm(const int* rowptr, const int* col, const float* val, int n_rows, const float* X, long ldx, int F,
                        float* Y, long ldy, hipStream_t s) {
  constexpr int RPW = 64 / LPR;
  long waves = ((long)n_rows + RPW - 1) / RPW;
  int blocks = gs_cdiv(waves, 4);
  hipLaunchKernelGGL((gs_spmm_csr_kernel<VEC, LPR>), dim3(blocks), dim3(256), 0, s, rowptr, col, val, n_rows, X, ldx, F,
                     Y, ldy);
}

extern "C" int gs_spmm_csr_f32(const int* rowptr, const int* col, const float* val, int n_rows, const float* X, long ldx,
                               int F, float* Y, long ldy, gs_stream_t stream) {
  if (n_rows <= 0 || F <= 0) return GS_OK;
  if (!rowptr || !col || !val || !X || !Y) GS_FAIL(GS_ERR_ARG, "gs_spmm_csr_f32: null pointer");
  if (ldx < F || ldy < F) GS_FAIL(GS_ERR_ARG, "gs_spmm_csr_f32: leading dimension smaller than F");
  hipStream_t s = (hipStream_t)stream;
  bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (((uintptr_t)X | (uintptr_t)Y) % 16 == 0);
  if (vec) {
    int q = F / 4;
    if (q > 32) launch_spmm<4, 64>(rowptr, col, val, n_rows, X, ldx, F, Y, ldy, s);
    else if (q > 16) launch_spmm<4, 32>(rowptr, col, val, n_rows, X, ldx, F, Y, ldy, s);
    else if (q > 8) launch_spmm<4, 16>(rowptr, col, val, n_rows, X, ldx, F, Y, ldy, s);
    else if (q > 4) launch_spmm<4, 8>(rowptr, col, val, n_rows, X, ldx, F, Y, ldy, s);
    else if (q > 2) launch_spmm<4, 4>(rowptr, col, val, n_rows, X, ldx, F, Y, ldy, s);
    else launch_spmm<4, 2>(rowptr, col, val, n_rows, X, ldx, F, Y, ldy, s);
  } else {
    if (F > 32) launch_spmm<1, 64>(rowptr, col, val, n_rows, X, ldx, F, Y, ldy, s);
    else if (F > 16) launch_spmm<1, 32>(rowptr, col, val, n_rows, X, ldx, F, Y, ldy, s);
    else if (F > 8) launch_spmm<1, 16>(rowptr, col, val, n_rows, X, ldx, F, Y, ldy, s);
    else if (F > 4) launch_spmm<1, 8>(rowptr, col, val, n_rows, X, ldx, F, Y, ldy, s);
    else launch_spmm<1, 4>(rowptr, col, val, n_rows, X, ldx, F, Y, ldy, s);
  }
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// ------------------------------------------------------------------------------------------------
__global__ void gs_gather_rows_kernel(const float* __restrict__ X, long ldx, const int* __restrict__ idx, int n, int F,
                                      float* __restrict__ out, long ldo) {
  const int lane = threadIdx.x & 63;
  const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= n) return;
  const float* src = X + (long)idx[row] * ldx;
  float* dst = out + row * ldo;
  for (int c = lane; c < F; c += 64) dst[c] = src[c];
}

extern "C" int gs_gather_rows_f32(const float* X, long ldx, const int* idx, int n, int F, float* out, long ldo,
                                  gs_stream_t stream) {
  if (n <= 0 || F <= 0) return GS_OK;
  if (!X || !idx || !out) GS_FAIL(GS_ERR_ARG, "gs_gather_rows_f32: null pointer");
  hipLaunchKernelGGL(gs_gather_rows_kernel, dim3(gs_cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, X, ldx, idx, n, F, out,
                     ldo);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// ------------------------------------------------------------------------------------------------
// dense_gcn_norm (utils.py:429-439):  M = adj + I, deg = rowsum(M), r = deg^-1/2 (inf -> 0),
// A^ = diag(r) M diag(r).   Backward: T = dA^ * M;  dr_i = sum_j T_ij r_j + sum_j T_ji r_j;
// d deg_i = -1/2 dr_i r_i^3;  d adj_ij = r_i r_j dA^_ij + d deg_i.
// ------------------------------------------------------------------------------------------------
__global__ void gs_norm_deg_kernel(const float* __restrict__ adj, int n, float* __restrict__ r) {
  const int i = blockIdx.x;
  float s = 0.f;
  for (int j = threadIdx.x; j < n; j += blockDim.x) s += adj[(long)i * n + j];
  __shared__ float red[4];
  s = gs_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float deg = red[0] + red[1] + red[2] + red[3] + 1.0f;
    float ri = powf(deg, -0.5f);
    r[i] = isinf(ri) ? 0.f : ri;
  }
}

__global__ void gs_norm_scale_kernel(const float* __restrict__ adj, const float* __restrict__ r, int n,
                                     float* __restrict__ out) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * n) return;
  int i = idx / n, j = idx - (long)i * n;
  float m = adj[idx] + (i == j ? 1.f : 0.f);
  out[idx] = r[i] * m * r[j];
}

extern "C" int gs_dense_gcn_norm_fwd(const float* adj, int n, float* adj_norm, float* r_out, gs_stream_t stream) {
  if (n <= 0) return GS_OK;
  if (!adj || !adj_norm || !r_out) GS_FAIL(GS_ERR_ARG, "gs_dense_gcn_norm_fwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gs_norm_deg_kernel, dim3(n), dim3(256), 0, s, adj, n, r_out);
  hipLaunchKernelGGL(gs_norm_scale_kernel, dim3(gs_cdiv((long)n * n, 256)), dim3(256), 0, s, adj, r_out, n, adj_norm);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// a_i = sum_j T_ij r_j (row pass) and b_i = sum_j T_ji r_j (column pass), T = dA * (adj + I); one block per i
__global__ void gs_norm_bwd_dr_kernel(const float* __restrict__ adj, const float* __restrict__ r,
                                      const float* __restrict__ dA, int n, float* __restrict__ ddeg) {
  const int i = blockIdx.x;
  float s = 0.f;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    float mij = adj[(long)i * n + j] + (i == j ? 1.f : 0.f);
    float mji = adj[(long)j * n + i] + (i == j ? 1.f : 0.f);
    s += dA[(long)i * n + j] * mij * r[j] + dA[(long)j * n + i] * mji * r[j];
  }
  __shared__ float red[4];
  s = gs_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float dr = red[0] + red[1] + red[2] + red[3];
    float ri = r[i];
    ddeg[i] = -0.5f * dr * ri * ri * ri;
  }
}

// Large n (a block-diagonal batch of structures: MSGC, 7 264 rows at ogbn-arxiv): the kernel above walks column i with stride n
// (0.88 ms measured at n = 7 264).  Here the row part runs one workgroup per row and the column part as per-chunk partial sums
// with the columns across the lanes (coalesced), added up in chunk order: ddeg_i = -r_i^3 / 2 (row_i + sum_c part[c][i]).
constexpr int GS_NORM_CHUNKS = 32;

__global__ __launch_bounds__(256) void gs_norm_bwd_rows_kernel(const float* __restrict__ adj, const float* __restrict__ r,
                                                               const float* __restrict__ dA, int n, float* __restrict__ rowpart) {
  const int i = blockIdx.x;
  float s = 0.f;
  for (int j = threadIdx.x; j < n; j += blockDim.x)
    s += dA[(long)i * n + j] * (adj[(long)i * n + j] + (i == j ? 1.f : 0.f)) * r[j];
  __shared__ float red[4];
  s = gs_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) rowpart[i] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void gs_norm_bwd_cols_kernel(const float* __restrict__ adj, const float* __restrict__ r,
                                                               const float* __restrict__ dA, int n, float* __restrict__ part) {
  __shared__ float sh[4][64];
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int per = gs_cdiv(n, GS_NORM_CHUNKS);
  const int j0 = blockIdx.y * per, j1 = j0 + per < n ? j0 + per : n;
  float s = 0.f;
  if (i < n)
    for (int j = j0 + rl; j < j1; j += 4) s += dA[(long)j * n + i] * (adj[(long)j * n + i] + (i == j ? 1.f : 0.f)) * r[j];
  sh[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && i < n) {
    const int c = threadIdx.x & 63;
    part[(long)blockIdx.y * n + i] = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
  }
}

__global__ void gs_norm_bwd_ddeg_kernel(const float* __restrict__ r, const float* __restrict__ part, int n, float* __restrict__ ddeg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = ddeg[i];                       // the row part
  for (int c = 0; c < GS_NORM_CHUNKS; ++c) s += part[(long)c * n + i];
  const float ri = r[i];
  ddeg[i] = -0.5f * s * ri * ri * ri;
}

__global__ void gs_norm_bwd_out_kernel(const float* __restrict__ r, const float* __restrict__ dA,
                                       const float* __restrict__ ddeg, int n, float* __restrict__ dadj) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * n) return;
  int i = idx / n, j = idx - (long)i * n;
  dadj[idx] = r[i] * r[j] * dA[idx] + ddeg[i];
}

extern "C" int gs_dense_gcn_norm_bwd(const float* adj, const float* r, const float* d_adj_norm, int n, float* d_adj,
                                     gs_stream_t stream) {
  if (n <= 0) return GS_OK;
  if (!adj || !r || !d_adj_norm || !d_adj) GS_FAIL(GS_ERR_ARG, "gs_dense_gcn_norm_bwd: null pointer");
  if (n > 2048 && d_adj == d_adj_norm)
    GS_FAIL(GS_ERR_ARG, "gs_dense_gcn_norm_bwd: d_adj must not alias d_adj_norm for n > 2048 (d_adj stages the column partials)");
  hipStream_t s = (hipStream_t)stream;
  // d deg is staged in the first n floats of d_adj's last row region?  no: use d_adj itself as scratch is unsafe;
  // the [n] vector lives at the tail of r (caller allocates r with 2n floats: r[0..n) = deg^-1/2, r[n..2n) scratch).
  float* ddeg = const_cast<float*>(r) + n;
  if (n > 2048) {      // d_adj's head holds the column partials until ddeg is complete (d_adj is written last, below)
    hipLaunchKernelGGL(gs_norm_bwd_rows_kernel, dim3(n), dim3(256), 0, s, adj, r, d_adj_norm, n, ddeg);
    hipLaunchKernelGGL(gs_norm_bwd_cols_kernel, dim3(gs_cdiv(n, 64), GS_NORM_CHUNKS), dim3(256), 0, s, adj, r, d_adj_norm, n, d_adj);
    hipLaunchKernelGGL(gs_norm_bwd_ddeg_kernel, dim3(gs_cdiv(n, 256)), dim3(256), 0, s, r, (const float*)d_adj, n, ddeg);
  } else {
    hipLaunchKernelGGL(gs_norm_bwd_dr_kernel, dim3(n), dim3(256), 0, s, adj, r, d_adj_norm, n, ddeg);
  }
  hipLaunchKernelGGL(gs_norm_bwd_out_kernel, dim3(gs_cdiv((long)n * n, 256)), dim3(256), 0, s, r, d_adj_norm, ddeg, n,
                     d_adj);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// ------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam, single-tensor form): m.lerp_(g, 1-b1); v = b2 v + (1-b2) g^2;
// p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// ------------------------------------------------------------------------------------------------
__global__ void gs_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                               float* __restrict__ v, long n, float step_size, float inv_sqrt_bc2, float b1, float b2,
                               float eps, float wd) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float gi = g[i], pi = p[i];
  if (wd != 0.f) gi = fmaf(wd, pi, gi);
  float mi = m[i], vi = v[i];
  mi = mi + (1.f - b1) * (gi - mi);
  vi = b2 * vi + (1.f - b2) * gi * gi;
  float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
  p[i] = pi - step_size * (mi / denom);
  m[i] = mi;
  v[i] = vi;
}

extern "C" int gs_adam_step_f32(float* param, const float* grad, float* m, float* v, long n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int step, gs_stream_t stream) {
  if (n <= 0) return GS_OK;
  if (!param || !grad || !m || !v || step < 1) GS_FAIL(GS_ERR_ARG, "gs_adam_step_f32: bad argument");
  double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  float step_size = (float)((double)lr / bc1);
  float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  hipLaunchKernelGGL(gs_adam_kernel, dim3(gs_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n,
                     step_size, inv_sqrt_bc2, beta1, beta2, eps, weight_decay);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// ------------------------------------------------------------------------------------------------
// Pieces of the evaluation GCN (models/base.py:51-78,80-187, layers.py:36-51): bias + ReLU in place,
// log-softmax / NLL gradient, column sums, argmax accuracy count.
// ------------------------------------------------------------------------------------------------
__global__ void gs_bias_act_kernel(float* __restrict__ X, long ld, const float* __restrict__ bias, long n, int F, int relu) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * F) return;
  const long i = idx / F;
  const int k = (int)(idx - i * F);
  float v = X[i * ld + k] + (bias ? bias[k] : 0.f);
  X[i * ld + k] = relu ? fmaxf(v, 0.f) : v;
}

extern "C" int gs_bias_act_f32(float* X, long ld, const float* bias, long n, int F, int relu, gs_stream_t stream) {
  if (n <= 0 || F <= 0) return GS_OK;
  if (!X || ld < F) GS_FAIL(GS_ERR_ARG, "gs_bias_act_f32: bad argument");
  hipLaunchKernelGGL(gs_bias_act_kernel, dim3(gs_cdiv(n * F, 256)), dim3(256), 0, (hipStream_t)stream, X, ld, bias, n, F, relu);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// G[i,:] = (softmax(Z[i,:]) - onehot(labels[i])) / n ; loss_rows[i] = -log_softmax(Z[i])[label]   (F.nll_loss mean)
__global__ __launch_bounds__(256) void gs_softmax_nll_grad_kernel(const float* __restrict__ Z, long ldz,
                                                                  const int* __restrict__ labels, long n, int C,
                                                                  float* __restrict__ G, float* __restrict__ loss_rows) {
  const int lane = threadIdx.x & 63;
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n) return;
  const float* z = Z + i * ldz;
  const int y = labels[i];
  float mx = -INFINITY;
  for (int k = lane; k < C; k += 64) mx = fmaxf(mx, z[k]);
  mx = gs_wave_max(mx);
  float se = 0.f;
  for (int k = lane; k < C; k += 64) se += expf(z[k] - mx);
  se = gs_wave_sum(se);
  const float inv = 1.f / se, invn = 1.f / (float)n;
  for (int k = lane; k < C; k += 64) {
    float p = expf(z[k] - mx) * inv;
    if (G) G[i * C + k] = (p - (k == y ? 1.f : 0.f)) * invn;
  }
  if (lane == 0 && loss_rows) loss_rows[i] = -(z[y] - mx - logf(se));
}

extern "C" int gs_softmax_nll_grad_f32(const float* Z, long ldz, const int* labels, long n, int C, float* G,
                                       float* loss_rows, gs_stream_t stream) {
  if (n <= 0 || C <= 0) return GS_OK;
  if (!Z || !labels || ldz < C) GS_FAIL(GS_ERR_ARG, "gs_softmax_nll_grad_f32: bad argument");
  hipLaunchKernelGGL(gs_softmax_nll_grad_kernel, dim3(gs_cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, Z, ldz, labels, n, C, G,
                     loss_rows);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// out[k] = sum_i M[i*ld + k]  (fp64 accumulation; block = 32 columns x 8 row groups)
__global__ __launch_bounds__(256) void gs_colsum_kernel(const float* __restrict__ M, long ld, long rows, int cols,
                                                        float* __restrict__ out) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int k = blockIdx.x * 32 + tx;
  __shared__ double sh[8][32];
  double s = 0;
  if (k < cols)
    for (long i = ty; i < rows; i += 8) s += M[i * ld + k];
  sh[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && k < cols) {
    s = 0;
    for (int q = 0; q < 8; ++q) s += sh[q][tx];
    out[k] = (float)s;
  }
}

extern "C" int gs_colsum_f32(const float* M, long ld, long rows, int cols, float* out, gs_stream_t stream) {
  if (cols <= 0) return GS_OK;
  if (!M || !out || ld < cols) GS_FAIL(GS_ERR_ARG, "gs_colsum_f32: bad argument");
  hipLaunchKernelGGL(gs_colsum_kernel, dim3(gs_cdiv(cols, 32)), dim3(256), 0, (hipStream_t)stream, M, ld, rows, cols, out);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// count[0] += #{ i : argmax_k Z[idx[i], k] == labels[i] }   (graphslim/utils.py:516-538; first maximum wins like torch.max)
__global__ __launch_bounds__(256) void gs_count_correct_kernel(const float* __restrict__ Z, long ldz, int C,
                                                               const int* __restrict__ idx, const int* __restrict__ labels,
                                                               long n, int* __restrict__ count) {
  const int lane = threadIdx.x & 63;
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int ok = 0;
  if (i < n) {
    const float* z = Z + (long)(idx ? idx[i] : i) * ldz;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    for (int k = lane; k < C; k += 64) {
      float v = z[k];
      if (v > best) { best = v; arg = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ob = __shfl_xor(best, o, 64);
      int oa = __shfl_xor(arg, o, 64);
      if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    ok = (lane == 0 && arg == labels[i]) ? 1 : 0;
  }
  if (ok) atomicAdd(count, 1);     // integer atomics: order-independent
}

extern "C" int gs_count_correct(const float* Z, long ldz, int C, const int* idx, const int* labels, long n, int* count,
                                gs_stream_t stream) {
  if (n <= 0) return GS_OK;
  if (!Z || !labels || !count || C <= 0) GS_FAIL(GS_ERR_ARG, "gs_count_correct: bad argument");
  hipLaunchKernelGGL(gs_count_correct_kernel, dim3(gs_cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, Z, ldz, C, idx, labels, n,
                     count);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// out[i,:] = log_softmax(Z[i,:])   (F.log_softmax, models/base.py:78) — one wave per row
__global__ __launch_bounds__(256) void gs_log_softmax_kernel(const float* __restrict__ Z, long ldz, long n, int C,
                                                             float* __restrict__ out, long ldo) {
  const int lane = threadIdx.x & 63;
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n) return;
  const float* z = Z + i * ldz;
  float mx = -INFINITY;
  for (int k = lane; k < C; k += 64) mx = fmaxf(mx, z[k]);
  mx = gs_wave_max(mx);
  float se = 0.f;
  for (int k = lane; k < C; k += 64) se += expf(z[k] - mx);
  se = gs_wave_sum(se);
  const float lse = mx + logf(se);
  for (int k = lane; k < C; k += 64) out[i * ldo + k] = z[k] - lse;
}

extern "C" int gs_log_softmax_f32(const float* Z, long ldz, long n, int C, float* out, long ldo, gs_stream_t stream) {
  if (n <= 0 || C <= 0) return GS_OK;
  if (!Z || !out || ldz < C || ldo < C) GS_FAIL(GS_ERR_ARG, "gs_log_softmax_f32: bad argument");
  hipLaunchKernelGGL(gs_log_softmax_kernel, dim3(gs_cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, Z, ldz, n, C, out, ldo);
  GS_LAUNCH_CHECK();
  return GS_OK;
}
