// Edge-list structure generator of MSGC ("Multiple sparse graphs condensation", graphslim/condensation/msgc.py:31-39,224-236):
// an MLP with training-mode BatchNorm that scores every directed edge of `batch_adj` sparse structures over the same synthetic
// nodes, scattered into a dense adjacency and symmetrised.  The MLP's matrix products run through gs_gemm_f32; this file holds
// the gather / BatchNorm / score / scatter kernels around them, forward and backward, all HBM-bound row streams:
//   * one wave per edge row, lanes over the hidden units (coalesced 256-B row segments);
//   * column statistics in two deterministic stages (fp64 partials per row chunk, then a fixed-order sum) — no float atomics;
//   * scatter-adds onto nodes are segment sums over a host-built edge permutation (fixed order, reproducible).
#include <hip/hip_runtime.h>

#include "../../include/gslim.h"
#include "gs_common.h"

namespace {

constexpr int SG_MAX_CHUNKS = 256;    // row chunks of the two-stage column statistics (scratch: 2 * SG_MAX_CHUNKS * F doubles)

inline int sg_chunks(long E) {
  long c = gs_cdiv(E, 256);
  return (int)(c < 1 ? 1 : (c > SG_MAX_CHUNKS ? SG_MAX_CHUNKS : c));
}

// out[e][k] = U[rows[e]][k] + V[cols[e]][k] + bias[k]
__global__ __launch_bounds__(256) void sg_pair_sum_kernel(const float* __restrict__ U, const float* __restrict__ V, long ld,
                                                          const float* __restrict__ bias, const int* __restrict__ rows,
                                                          const int* __restrict__ cols, long E, int F, float* __restrict__ out,
                                                          long ldo) {
  const int lane = threadIdx.x & 63;
  const long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (e >= E) return;
  const float* u = U + (long)rows[e] * ld;
  const float* v = V + (long)cols[e] * ld;
  float* o = out + e * ldo;
  for (int k = lane; k < F; k += 64) o[k] = (u[k] + v[k]) + (bias ? bias[k] : 0.f);
}

// Stage 1 of the column statistics.  MODE 0: (sum x, sum x^2).  MODE 1: (sum dz, sum dz * xhat) with dz = dY * [Y > 0],
// xhat = (x - mu) * istd.  grid (cdiv(F, 64), chunks), block 256 = 64 columns x 4 row lanes.
template <int MODE>
__global__ __launch_bounds__(256) void sg_colstats_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ dY,
                                                          long lddy, const float* __restrict__ Y, long ldy,
                                                          const float* __restrict__ mu, const float* __restrict__ istd, long E, int F,
                                                          int chunks, double* __restrict__ part) {
  __shared__ double sh[2][4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const long per = gs_cdiv(E, (long)chunks);
  const long r0 = (long)blockIdx.y * per;
  const long r1 = r0 + per < E ? r0 + per : E;
  double a = 0.0, b = 0.0;
  if (col < F) {
    const float m = MODE == 1 ? mu[col] : 0.f, is = MODE == 1 ? istd[col] : 0.f;
    for (long r = r0 + rl; r < r1; r += 4) {
      const float x = X[r * ldx + col];
      if (MODE == 0) {
        a += (double)x;
        b += (double)x * (double)x;
      } else {
        const float dz = Y[r * ldy + col] > 0.f ? dY[r * lddy + col] : 0.f;
        a += (double)dz;
        b += (double)dz * (double)((x - m) * is);
      }
    }
  }
  sh[0][rl][threadIdx.x & 63] = a;
  sh[1][rl][threadIdx.x & 63] = b;
  __syncthreads();
  if (rl == 0 && col < F) {
    const int c = threadIdx.x & 63;
    part[((long)blockIdx.y * 2 + 0) * F + col] = ((sh[0][0][c] + sh[0][1][c]) + sh[0][2][c]) + sh[0][3][c];
    part[((long)blockIdx.y * 2 + 1) * F + col] = ((sh[1][0][c] + sh[1][1][c]) + sh[1][2][c]) + sh[1][3][c];
  }
}

// Stage 2: the chunk partials of column `col` summed by 4 lanes over contiguous quarters of the chunks, combined in lane order
// (block = 64 columns x 4 lanes).  Returns the two totals to lane 0 of every column.
__device__ __forceinline__ void sg_sum_parts(const double* __restrict__ part, int chunks, int F, int col, double* s_out, double* q_out) {
  __shared__ double sh[2][4][64];
  const int c = threadIdx.x & 63, l = threadIdx.x >> 6;
  const int per = (chunks + 3) >> 2;
  int c0 = l * per, c1 = c0 + per;
  if (c1 > chunks) c1 = chunks;
  double s = 0.0, q = 0.0;
  if (col < F)
    for (int k = c0; k < c1; ++k) { s += part[((long)k * 2 + 0) * F + col]; q += part[((long)k * 2 + 1) * F + col]; }
  sh[0][l][c] = s;
  sh[1][l][c] = q;
  __syncthreads();
  *s_out = ((sh[0][0][c] + sh[0][1][c]) + sh[0][2][c]) + sh[0][3][c];
  *q_out = ((sh[1][0][c] + sh[1][1][c]) + sh[1][2][c]) + sh[1][3][c];
}

// MODE 0: mean and 1/sqrt(biased variance + eps) (nn.BatchNorm1d in training mode normalises with the biased variance).
__global__ __launch_bounds__(256) void sg_bn_finish_kernel(const double* __restrict__ part, int chunks, long E, int F, float eps,
                                                           float* __restrict__ mu, float* __restrict__ istd) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  double s, q;
  sg_sum_parts(part, chunks, F, col, &s, &q);
  if (col >= F || threadIdx.x >= 64) return;
  const double m = s / (double)E;
  double var = q / (double)E - m * m;
  if (var < 0.0) var = 0.0;
  mu[col] = (float)m;
  istd[col] = (float)(1.0 / sqrt(var + (double)eps));
}

// MODE 1: the two sums as floats: dbe = sum dz, dg = sum dz * xhat
__global__ __launch_bounds__(256) void sg_bwd_finish_kernel(const double* __restrict__ part, int chunks, int F, float* __restrict__ dbe,
                                                            float* __restrict__ dg) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  double s, q;
  sg_sum_parts(part, chunks, F, col, &s, &q);
  if (col >= F || threadIdx.x >= 64) return;
  dbe[col] = (float)s;
  dg[col] = (float)q;
}

// plain column sum: out[k] = sum x
__global__ __launch_bounds__(256) void sg_sum_finish_kernel(const double* __restrict__ part, int chunks, int F, float* __restrict__ out) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  double s, q;
  sg_sum_parts(part, chunks, F, col, &s, &q);
  if (col >= F || threadIdx.x >= 64) return;
  out[col] = (float)s;
}

// Y = relu(g * (x - mu) * istd + be)
__global__ __launch_bounds__(256) void sg_bn_relu_apply_kernel(const float* __restrict__ X, long ldx, long E, int F,
                                                               const float* __restrict__ mu, const float* __restrict__ istd,
                                                               const float* __restrict__ g, const float* __restrict__ be,
                                                               float* __restrict__ Y, long ldy) {
  const int lane = threadIdx.x & 63;
  const long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (e >= E) return;
  for (int k = lane; k < F; k += 64) {
    const float xh = (X[e * ldx + k] - mu[k]) * istd[k];
    Y[e * ldy + k] = fmaxf(fmaf(g[k], xh, be[k]), 0.f);
  }
}

// dX = g * istd * (dz - mean(dz) - xhat * mean(dz * xhat));  dX may alias dY
__global__ __launch_bounds__(256) void sg_bn_relu_bwd_apply_kernel(const float* dY, long lddy, const float* __restrict__ Y, long ldy,
                                                                   const float* __restrict__ X, long ldx, long E, int F,
                                                                   const float* __restrict__ mu, const float* __restrict__ istd,
                                                                   const float* __restrict__ g, const float* __restrict__ dbe,
                                                                   const float* __restrict__ dg, float* dX, long lddx) {
  const int lane = threadIdx.x & 63;
  const long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (e >= E) return;
  const float inv = 1.f / (float)E;
  for (int k = lane; k < F; k += 64) {
    const float xh = (X[e * ldx + k] - mu[k]) * istd[k];
    const float dz = Y[e * ldy + k] > 0.f ? dY[e * lddy + k] : 0.f;
    dX[e * lddx + k] = g[k] * istd[k] * (dz - dbe[k] * inv - xh * (dg[k] * inv));
  }
}

// s[e] = sigmoid(A[e] . w + b)
__global__ __launch_bounds__(256) void sg_score_kernel(const float* __restrict__ A, long ld, const float* __restrict__ w,
                                                       const float* __restrict__ b, long E, int F, float* __restrict__ s) {
  const int lane = threadIdx.x & 63;
  const long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (e >= E) return;
  float acc = 0.f;
  for (int k = lane; k < F; k += 64) acc = fmaf(A[e * ld + k], w[k], acc);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) s[e] = 1.f / (1.f + expf(-(acc + b[0])));
}

// M[pr[e]][pc[e]] = s[e]   (index assignment: duplicate (r, c) pairs carry identical scores — same inputs, same statistics)
__global__ void sg_scatter_kernel(const float* __restrict__ s, const int* __restrict__ pr, const int* __restrict__ pc, long E, int n,
                                  float* __restrict__ M) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < E) M[(long)pr[e] * n + pc[e]] = s[e];
}

// adj = (M + M^T) / 2
__global__ void sg_symmetrise_kernel(const float* __restrict__ M, int n, float* __restrict__ adj) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * n) return;
  const long r = idx / n, c = idx % n;
  adj[idx] = (M[c * n + r] + M[idx]) / 2.f;
}

// dlogit[e] = (dA[r][c] + dA[c][r]) / 2 * s (1 - s);  dA2[e][k] = dlogit[e] * w[k]
__global__ __launch_bounds__(256) void sg_score_bwd_kernel(const float* __restrict__ dA, const int* __restrict__ pr,
                                                           const int* __restrict__ pc, const float* __restrict__ s,
                                                           const float* __restrict__ w, long E, int n, int F,
                                                           float* __restrict__ dlogit, float* __restrict__ dA2, long ld) {
  const int lane = threadIdx.x & 63;
  const long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (e >= E) return;
  const long r = pr[e], c = pc[e];
  const float se = s[e];
  const float dl = (dA[r * n + c] + dA[c * n + r]) / 2.f * (se * (1.f - se));
  if (lane == 0) dlogit[e] = dl;
  for (int k = lane; k < F; k += 64) dA2[e * ld + k] = dl * w[k];
}

// out[i][k] = sum_{j in [ptr[i], ptr[i+1])} Xe[perm[j]][k].  One workgroup per (node, 64-column chunk); the segment is cut into
// four contiguous quarters, one per wave, each summed in order, and the four partial sums are added in wave order: a fixed
// summation tree, reproducible from run to run.
__global__ __launch_bounds__(256) void sg_segment_sum_kernel(const float* __restrict__ Xe, long ld, const int* __restrict__ ptr,
                                                             const int* __restrict__ perm, int n, int F, float* __restrict__ out,
                                                             long ldo) {
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x;
  const int k = blockIdx.y * 64 + lane;
  const int j0 = ptr[i], j1 = ptr[i + 1];
  const int len = j1 - j0, q = (len + 3) >> 2;
  int a = j0 + w * q, b = a + q;
  if (a > j1) a = j1;
  if (b > j1) b = j1;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  if (k < F) {
    int j = a;
    for (; j + 4 <= b; j += 4) {       // four independent loads in flight; added in index order below
      const float x0 = Xe[(long)perm[j] * ld + k], x1 = Xe[(long)perm[j + 1] * ld + k];
      const float x2 = Xe[(long)perm[j + 2] * ld + k], x3 = Xe[(long)perm[j + 3] * ld + k];
      acc0 += x0; acc1 += x1; acc2 += x2; acc3 += x3;
    }
    for (; j < b; ++j) acc0 += Xe[(long)perm[j] * ld + k];
  }
  sh[w][lane] = (acc0 + acc1) + (acc2 + acc3);
  __syncthreads();
  if (w == 0 && k < F) out[(long)i * ldo + k] = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
}


// ---- IGNR pieces (SGDD, graphslim/models/ignr.py:151-177) ---------------------------------------------------------------
// z = a * b  (ignr.py:167 `x = x * c`)
__global__ __launch_bounds__(256) void sg_mul_kernel(const float* __restrict__ a, const float* __restrict__ b, long total,
                                                     float* __restrict__ z) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) z[i] = a[i] * b[i];
}

// backward of z = a * b: da = dz * b;  db = (acc ? db : 0) + dz * a.   da may alias dz.
__global__ __launch_bounds__(256) void sg_mul_bwd_kernel(const float* dz, const float* __restrict__ a, const float* __restrict__ b,
                                                         long total, float* da, float* __restrict__ db, int acc) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float g = dz[i];
  const float vb = g * a[i];
  db[i] = acc ? db[i] + vb : vb;
  da[i] = g * b[i];
}

// e3[e] = wa * (A[e] . ua + ba[0]) + wb * (B[e] . ub + bb[0])   (ignr.py:161-170: the two Linear(nfeat, 1) heads mixed by ep_ratio)
__global__ __launch_bounds__(256) void sg_mix_heads_kernel(const float* __restrict__ A, const float* __restrict__ ua,
                                                           const float* __restrict__ ba, float wa, const float* __restrict__ B,
                                                           const float* __restrict__ ub, const float* __restrict__ bb, float wb,
                                                           long E, int F, float* __restrict__ e3) {
  const int lane = threadIdx.x & 63;
  const long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (e >= E) return;
  float sa = 0.f, sb = 0.f;
  for (int k = lane; k < F; k += 64) { sa = fmaf(A[e * F + k], ua[k], sa); sb = fmaf(B[e * F + k], ub[k], sb); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
  if (lane == 0) e3[e] = wa * (sa + ba[0]) + wb * (sb + bb[0]);
}

// dA[e][k] = wa * de3[e] * ua[k];  dB[e][k] = wb * de3[e] * ub[k];  sa[e] = wa * de3[e];  sb[e] = wb * de3[e] (the heads' own
// upstream gradients: d ua = A^T sa, d ba = sum sa, ...)
__global__ __launch_bounds__(256) void sg_mix_heads_bwd_kernel(const float* __restrict__ de3, const float* __restrict__ ua, float wa,
                                                               const float* __restrict__ ub, float wb, long E, int F,
                                                               float* __restrict__ dA, float* __restrict__ dB, float* __restrict__ sa,
                                                               float* __restrict__ sb) {
  const int lane = threadIdx.x & 63;
  const long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (e >= E) return;
  const float g = de3[e];
  if (lane == 0) { sa[e] = wa * g; sb[e] = wb * g; }
  for (int k = lane; k < F; k += 64) { dA[e * F + k] = wa * g * ua[k]; dB[e * F + k] = wb * g * ub[k]; }
}

// adj = sigmoid((e3 + e3^T) / 2) with a zero diagonal (ignr.py:174-177)
__global__ void sg_sym_sigmoid_kernel(const float* __restrict__ e3, int n, float* __restrict__ adj) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * n) return;
  const long i = idx / n, j = idx % n;
  adj[idx] = i == j ? 0.f : 1.f / (1.f + expf(-(e3[idx] + e3[j * n + i]) / 2.f));
}

// its backward: de3[i][j] = (dA[i][j] + dA[j][i]) / 2 * a (1 - a) off the diagonal (a is symmetric), 0 on it
__global__ void sg_sym_sigmoid_bwd_kernel(const float* __restrict__ dA, const float* __restrict__ adj, int n, float* __restrict__ de3) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * n) return;
  const long i = idx / n, j = idx % n;
  const float a = adj[idx];
  de3[idx] = i == j ? 0.f : (dA[idx] + dA[j * n + i]) / 2.f * (a * (1.f - a));
}

// nn.BatchNorm1d bookkeeping: running <- (1 - momentum) running + momentum (batch mean, UNBIASED batch variance)
__global__ void sg_bn_running_kernel(const float* __restrict__ mu, const float* __restrict__ istd, long E, float eps, float momentum,
                                     int F, float* __restrict__ rm, float* __restrict__ rv) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= F) return;
  const float is = istd[k];
  float var = 1.f / (is * is) - eps;               // the biased batch variance the forward normalised with
  if (var < 0.f) var = 0.f;
  if (E > 1) var = var * ((float)E / (float)(E - 1));
  rm[k] = (1.f - momentum) * rm[k] + momentum * mu[k];
  rv[k] = (1.f - momentum) * rv[k] + momentum * var;
}

__global__ void sg_bn_eval_consts_kernel(const float* __restrict__ rm, const float* __restrict__ rv, float eps, int F,
                                         float* __restrict__ mu, float* __restrict__ istd) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= F) return;
  mu[k] = rm[k];
  istd[k] = 1.f / sqrtf(rv[k] + eps);
}


// ---- optimal-transport regulariser of SGDD (graphslim/models/ignr.py:30-41,184-209) in the eigenbasis of the symmetric structure ----
// adj = Q diag(lam) Q^T.  torch.svd gives U = Q, S = |lam|, V = Q sign(lam), so U f(S) V^T = Q diag(sign(lam) f(|lam|)) Q^T:
//   Ly_inv_rt = Q diag(g) Q^T,  g_i = sign(lam_i) |lam_i|^-1/2;   Ly_inv = Q diag(1 / lam_i) Q^T
// with the entries whose singular value EQUALS the smallest one zeroed when that one is below 0.009 (ignr.py:32-38).
// One workgroup: out[0..n) = g, [n..2n) = g' (derivative, 0 where zeroed), [2n..3n) = d(1/lam)/dlam = -1/lam^2 (0 where zeroed),
// [3n] = trace(Ly_inv).
__global__ __launch_bounds__(256) void sg_ot_spectrum_kernel(const float* __restrict__ lam, int n, float sv_eps, float* __restrict__ out) {
  __shared__ float red[256];
  float mn = INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) mn = fminf(mn, fabsf(lam[i]));
  red[threadIdx.x] = mn;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = fminf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  const float smin = red[0];
  __syncthreads();
  float tr = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float l = lam[i], sa = fabsf(l);
    const bool keep = !(smin < sv_eps) || sa > smin;
    const float sg = l < 0.f ? -1.f : 1.f;
    const float inv = keep ? 1.f / sa : 0.f;
    out[i] = sg * sqrtf(inv);
    out[n + i] = keep ? -0.5f * inv * sqrtf(inv) : 0.f;      // d/dlam [sign |lam|^-1/2] = -1/2 |lam|^-3/2
    out[2 * n + i] = keep ? -inv * inv : 0.f;                // d/dlam [1 / lam]
    tr += sg * inv;
  }
  red[threadIdx.x] = tr;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[3 * n] = red[0];
}

// value = | trace * n - 2 sum sqrt(max(mu, 2e-20)) |  (ignr.py:205-208); acc[0] += value;  dmu_k = scale * sign * (-1 / sqrt(mu_k))
// (0 where clamped);  out[n] = scale * sign * n = d / d trace.   One workgroup.
__global__ __launch_bounds__(256) void sg_ot_value_kernel(const float* __restrict__ mu, const float* __restrict__ trace, int n, float scale,
                                                          float* __restrict__ dmu, float* __restrict__ value, double* __restrict__ acc) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += sqrtf(fmaxf(mu[i], 2e-20f));
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float inner = trace[0] * (float)n - 2.f * red[0];
  const float sg = inner < 0.f ? -1.f : (inner > 0.f ? 1.f : 0.f);
  __syncthreads();
  // Structurally zero eigenvalues (a zeroed singular value of the structure, or fewer corner rows than nodes, make the product
  // singular) come out of an fp32 decomposition as +-rounding noise.  Upstream's clamp(min=2e-20) gives them zero gradient when
  // the noise is negative and -1/sqrt(noise) when it is positive; here everything below 1e-6 of the largest eigenvalue counts as
  // clamped, which is the exact-arithmetic gradient.
  float mx = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, mu[i]);
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  const float thr = fmaxf(2e-20f, 1e-6f * red[0]);
  for (int i = threadIdx.x; i < n; i += 256) dmu[i] = mu[i] > thr ? -scale * sg / sqrtf(mu[i]) : 0.f;
  if (threadIdx.x == 0) {
    dmu[n] = scale * sg * (float)n;
    value[0] = fabsf(inner);
    if (acc) acc[0] += (double)fabsf(inner);
  }
}

// x <- clamp(x, -bound, bound), NaN -> 0
__global__ void sg_clamp_finite_kernel(float* __restrict__ x, long n, float bound) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  x[i] = v != v ? 0.f : fminf(fmaxf(v, -bound), bound);
}

// B[i][k] = A[i][k] * v[k]
__global__ void sg_scale_cols_kernel(const float* __restrict__ A, const float* __restrict__ v, int rows, int cols, float* __restrict__ B) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (long)rows * cols) B[idx] = A[idx] * v[idx % cols];
}

// K[i][j] <- F(lam_i, lam_j) K[i][j] + [i == j] dtr[0] hp[i]:   F = (g_i - g_j) / (lam_i - lam_j), or g'(lam_i) on the diagonal and
// for numerically equal eigenvalues (Daleckii-Krein: the derivative of the matrix function Q diag(g(lam)) Q^T in the eigenbasis)
__global__ void sg_ot_divdiff_kernel(float* __restrict__ K, const float* __restrict__ lam, const float* __restrict__ g,
                                     const float* __restrict__ gp, const float* __restrict__ hp, const float* __restrict__ dtr, int n) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * n) return;
  const int i = idx / n, j = idx % n;
  const float li = lam[i], lj = lam[j];
  const float den = li - lj;
  float F;
  if (i == j || fabsf(den) <= 1e-6f * fmaxf(fabsf(li), fabsf(lj))) F = 0.5f * (gp[i] + gp[j]);
  else F = (g[i] - g[j]) / den;
  K[idx] = F * K[idx] + (i == j ? dtr[0] * hp[i] : 0.f);
}

}  // namespace

extern "C" int gs_edge_pair_sum_f32(const float* U, const float* V, long ld, const float* bias, const int* rows, const int* cols,
                                    long E, int F, float* out, long ldo, gs_stream_t stream) {
  if (E <= 0) return GS_OK;
  if (!U || !V || !rows || !cols || !out || F <= 0 || ld < F || ldo < F) GS_FAIL(GS_ERR_ARG, "gs_edge_pair_sum_f32: bad argument");
  hipLaunchKernelGGL(sg_pair_sum_kernel, dim3(gs_cdiv(E, 4)), dim3(256), 0, (hipStream_t)stream, U, V, ld, bias, rows, cols, E, F, out,
                     ldo);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" size_t gs_bn_scratch_bytes(int F) { return sizeof(double) * 2 * (size_t)SG_MAX_CHUNKS * (size_t)(F > 0 ? F : 0); }

extern "C" int gs_bn_relu_fwd_f32(const float* X, long ldx, long E, int F, const float* g, const float* be, float eps, float* Y,
                                  long ldy, float* mu, float* istd, void* scratch, gs_stream_t stream) {
  if (!X || !g || !be || !Y || !mu || !istd || !scratch || E <= 0 || F <= 0 || ldx < F || ldy < F)
    GS_FAIL(GS_ERR_ARG, "gs_bn_relu_fwd_f32: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const int chunks = sg_chunks(E);
  double* part = (double*)scratch;
  hipLaunchKernelGGL((sg_colstats_kernel<0>), dim3(gs_cdiv(F, 64), chunks), dim3(256), 0, s, X, ldx, (const float*)nullptr, 0L,
                     (const float*)nullptr, 0L, (const float*)nullptr, (const float*)nullptr, E, F, chunks, part);
  hipLaunchKernelGGL(sg_bn_finish_kernel, dim3(gs_cdiv(F, 64)), dim3(256), 0, s, part, chunks, E, F, eps, mu, istd);
  hipLaunchKernelGGL(sg_bn_relu_apply_kernel, dim3(gs_cdiv(E, 4)), dim3(256), 0, s, X, ldx, E, F, mu, istd, g, be, Y, ldy);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_bn_relu_bwd_f32(const float* dY, long lddy, const float* Y, long ldy, const float* X, long ldx, long E, int F,
                                  const float* mu, const float* istd, const float* g, float* dX, long lddx, float* dg, float* dbe,
                                  void* scratch, gs_stream_t stream) {
  if (!dY || !Y || !X || !mu || !istd || !g || !dX || !dg || !dbe || !scratch || E <= 0 || F <= 0 || lddy < F || ldy < F || ldx < F ||
      lddx < F)
    GS_FAIL(GS_ERR_ARG, "gs_bn_relu_bwd_f32: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const int chunks = sg_chunks(E);
  double* part = (double*)scratch;
  hipLaunchKernelGGL((sg_colstats_kernel<1>), dim3(gs_cdiv(F, 64), chunks), dim3(256), 0, s, X, ldx, dY, lddy, Y, ldy, mu, istd, E, F,
                     chunks, part);
  hipLaunchKernelGGL(sg_bwd_finish_kernel, dim3(gs_cdiv(F, 64)), dim3(256), 0, s, part, chunks, F, dbe, dg);
  hipLaunchKernelGGL(sg_bn_relu_bwd_apply_kernel, dim3(gs_cdiv(E, 4)), dim3(256), 0, s, dY, lddy, Y, ldy, X, ldx, E, F, mu, istd, g,
                     dbe, dg, dX, lddx);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_edge_score_f32(const float* A, long ld, const float* w, const float* b, long E, int F, float* s,
                                 gs_stream_t stream) {
  if (E <= 0) return GS_OK;
  if (!A || !w || !b || !s || F <= 0 || ld < F) GS_FAIL(GS_ERR_ARG, "gs_edge_score_f32: bad argument");
  hipLaunchKernelGGL(sg_score_kernel, dim3(gs_cdiv(E, 4)), dim3(256), 0, (hipStream_t)stream, A, ld, w, b, E, F, s);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_edge_scatter_sym_f32(const float* s, const int* pr, const int* pc, long E, int n, float* M, float* adj,
                                       gs_stream_t stream) {
  if (!M || !adj || n <= 0 || (E > 0 && (!s || !pr || !pc))) GS_FAIL(GS_ERR_ARG, "gs_edge_scatter_sym_f32: bad argument");
  hipStream_t st = (hipStream_t)stream;
  GS_HIP(hipMemsetAsync(M, 0, sizeof(float) * (size_t)n * n, st));
  if (E > 0) hipLaunchKernelGGL(sg_scatter_kernel, dim3(gs_cdiv(E, 256)), dim3(256), 0, st, s, pr, pc, E, n, M);
  hipLaunchKernelGGL(sg_symmetrise_kernel, dim3(gs_cdiv((long)n * n, 256)), dim3(256), 0, st, M, n, adj);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_edge_score_bwd_f32(const float* dA, const int* pr, const int* pc, const float* s, const float* w, long E, int n,
                                     int F, float* dlogit, float* dA2, long ld, gs_stream_t stream) {
  if (E <= 0) return GS_OK;
  if (!dA || !pr || !pc || !s || !w || !dlogit || !dA2 || n <= 0 || F <= 0 || ld < F)
    GS_FAIL(GS_ERR_ARG, "gs_edge_score_bwd_f32: bad argument");
  hipLaunchKernelGGL(sg_score_bwd_kernel, dim3(gs_cdiv(E, 4)), dim3(256), 0, (hipStream_t)stream, dA, pr, pc, s, w, E, n, F, dlogit,
                     dA2, ld);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_segment_sum_rows_f32(const float* Xe, long ld, const int* ptr, const int* perm, int n, int F, float* out, long ldo,
                                       gs_stream_t stream) {
  if (n <= 0) return GS_OK;
  if (!Xe || !ptr || !perm || !out || F <= 0 || ld < F || ldo < F) GS_FAIL(GS_ERR_ARG, "gs_segment_sum_rows_f32: bad argument");
  hipLaunchKernelGGL(sg_segment_sum_kernel, dim3(n, gs_cdiv(F, 64)), dim3(256), 0, (hipStream_t)stream, Xe, ld, ptr, perm, n, F, out,
                     ldo);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_colsum_tall_f32(const float* X, long ld, long E, int F, float* out, void* scratch, gs_stream_t stream) {
  if (!X || !out || !scratch || E <= 0 || F <= 0 || ld < F) GS_FAIL(GS_ERR_ARG, "gs_colsum_tall_f32: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const int chunks = sg_chunks(E);
  double* part = (double*)scratch;
  hipLaunchKernelGGL((sg_colstats_kernel<0>), dim3(gs_cdiv(F, 64), chunks), dim3(256), 0, s, X, ld, (const float*)nullptr, 0L,
                     (const float*)nullptr, 0L, (const float*)nullptr, (const float*)nullptr, E, F, chunks, part);
  hipLaunchKernelGGL(sg_sum_finish_kernel, dim3(gs_cdiv(F, 64)), dim3(256), 0, s, part, chunks, F, out);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_mul_f32(const float* a, const float* b, long total, float* z, gs_stream_t stream) {
  if (total <= 0) return GS_OK;
  if (!a || !b || !z) GS_FAIL(GS_ERR_ARG, "gs_mul_f32: null argument");
  hipLaunchKernelGGL(sg_mul_kernel, dim3(gs_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, a, b, total, z);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_mul_bwd_f32(const float* dz, const float* a, const float* b, long total, float* da, float* db, int accumulate_db,
                              gs_stream_t stream) {
  if (total <= 0) return GS_OK;
  if (!dz || !a || !b || !da || !db) GS_FAIL(GS_ERR_ARG, "gs_mul_bwd_f32: null argument");
  hipLaunchKernelGGL(sg_mul_bwd_kernel, dim3(gs_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dz, a, b, total, da, db,
                     accumulate_db);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_mix_heads_f32(const float* A, const float* ua, const float* ba, float wa, const float* B, const float* ub,
                                const float* bb, float wb, long E, int F, float* e3, gs_stream_t stream) {
  if (E <= 0) return GS_OK;
  if (!A || !ua || !ba || !B || !ub || !bb || !e3 || F <= 0) GS_FAIL(GS_ERR_ARG, "gs_mix_heads_f32: bad argument");
  hipLaunchKernelGGL(sg_mix_heads_kernel, dim3(gs_cdiv(E, 4)), dim3(256), 0, (hipStream_t)stream, A, ua, ba, wa, B, ub, bb, wb, E, F, e3);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_mix_heads_bwd_f32(const float* de3, const float* ua, float wa, const float* ub, float wb, long E, int F, float* dA,
                                    float* dB, float* sa, float* sb, gs_stream_t stream) {
  if (E <= 0) return GS_OK;
  if (!de3 || !ua || !ub || !dA || !dB || !sa || !sb || F <= 0) GS_FAIL(GS_ERR_ARG, "gs_mix_heads_bwd_f32: bad argument");
  hipLaunchKernelGGL(sg_mix_heads_bwd_kernel, dim3(gs_cdiv(E, 4)), dim3(256), 0, (hipStream_t)stream, de3, ua, wa, ub, wb, E, F, dA, dB,
                     sa, sb);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_sym_sigmoid_f32(const float* e3, int n, float* adj, gs_stream_t stream) {
  if (!e3 || !adj || n <= 0) GS_FAIL(GS_ERR_ARG, "gs_sym_sigmoid_f32: bad argument");
  hipLaunchKernelGGL(sg_sym_sigmoid_kernel, dim3(gs_cdiv((long)n * n, 256)), dim3(256), 0, (hipStream_t)stream, e3, n, adj);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_sym_sigmoid_bwd_f32(const float* dA, const float* adj, int n, float* de3, gs_stream_t stream) {
  if (!dA || !adj || !de3 || n <= 0) GS_FAIL(GS_ERR_ARG, "gs_sym_sigmoid_bwd_f32: bad argument");
  hipLaunchKernelGGL(sg_sym_sigmoid_bwd_kernel, dim3(gs_cdiv((long)n * n, 256)), dim3(256), 0, (hipStream_t)stream, dA, adj, n, de3);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_bn_running_update_f32(const float* mu, const float* istd, long E, int F, float eps, float momentum, float* running_mean,
                                        float* running_var, gs_stream_t stream) {
  if (!mu || !istd || !running_mean || !running_var || F <= 0 || E <= 0) GS_FAIL(GS_ERR_ARG, "gs_bn_running_update_f32: bad argument");
  hipLaunchKernelGGL(sg_bn_running_kernel, dim3(gs_cdiv(F, 64)), dim3(64), 0, (hipStream_t)stream, mu, istd, E, eps, momentum, F,
                     running_mean, running_var);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_bn_relu_eval_f32(const float* X, long ldx, long E, int F, const float* g, const float* be, float eps,
                                   const float* running_mean, const float* running_var, float* Y, long ldy, float* mu, float* istd,
                                   gs_stream_t stream) {
  if (!X || !g || !be || !running_mean || !running_var || !Y || !mu || !istd || E <= 0 || F <= 0 || ldx < F || ldy < F)
    GS_FAIL(GS_ERR_ARG, "gs_bn_relu_eval_f32: bad argument");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sg_bn_eval_consts_kernel, dim3(gs_cdiv(F, 64)), dim3(64), 0, s, running_mean, running_var, eps, F, mu, istd);
  hipLaunchKernelGGL(sg_bn_relu_apply_kernel, dim3(gs_cdiv(E, 4)), dim3(256), 0, s, X, ldx, E, F, mu, istd, g, be, Y, ldy);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_ot_spectrum_f32(const float* lam, int n, float sv_eps, float* out, gs_stream_t stream) {
  if (!lam || !out || n <= 0) GS_FAIL(GS_ERR_ARG, "gs_ot_spectrum_f32: bad argument");
  hipLaunchKernelGGL(sg_ot_spectrum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lam, n, sv_eps, out);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_ot_value_f32(const float* mu, const float* trace, int n, float scale, float* dmu, float* value, double* acc,
                               gs_stream_t stream) {
  if (!mu || !trace || !dmu || !value || n <= 0) GS_FAIL(GS_ERR_ARG, "gs_ot_value_f32: bad argument");
  hipLaunchKernelGGL(sg_ot_value_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mu, trace, n, scale, dmu, value, acc);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_scale_cols_f32(const float* A, const float* v, int rows, int cols, float* B, gs_stream_t stream) {
  if (!A || !v || !B || rows <= 0 || cols <= 0) GS_FAIL(GS_ERR_ARG, "gs_scale_cols_f32: bad argument");
  hipLaunchKernelGGL(sg_scale_cols_kernel, dim3(gs_cdiv((long)rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, A, v, rows, cols, B);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_ot_divdiff_f32(float* K, const float* lam, const float* g, const float* gp, const float* hp, const float* dtr, int n,
                                 gs_stream_t stream) {
  if (!K || !lam || !g || !gp || !hp || !dtr || n <= 0) GS_FAIL(GS_ERR_ARG, "gs_ot_divdiff_f32: bad argument");
  hipLaunchKernelGGL(sg_ot_divdiff_kernel, dim3(gs_cdiv((long)n * n, 256)), dim3(256), 0, (hipStream_t)stream, K, lam, g, gp, hp, dtr, n);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_clamp_finite_f32(float* x, long n, float bound, gs_stream_t stream) {
  if (n <= 0) return GS_OK;
  if (!x || !(bound > 0.f)) GS_FAIL(GS_ERR_ARG, "gs_clamp_finite_f32: bad argument");
  hipLaunchKernelGGL(sg_clamp_finite_kernel, dim3(gs_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, n, bound);
  GS_LAUNCH_CHECK();
  return GS_OK;
}
