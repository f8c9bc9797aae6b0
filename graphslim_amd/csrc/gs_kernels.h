// Custom (non-GEMM) kernels of the GCond engine.  Included once, by gs_engine.hip.
// Every kernel cites the reference line whose arithmetic it carries.
#pragma once
#include "gs_common.h"

// ------------------------------------------------------------------------------------------------
// Dropout masks of the condensation model (models/sgc.py:42: F.dropout behind the hidden transform, training mode inside train_class
// and the inner steps).  torch's generator cannot be replayed here, so the masks are a counter-based function of
// (seed, iteration, kind, index, row, column) — the same Bernoulli(1 - p) entries scaled by 1 / (1 - p):
//   key  = mix64(mix64(seed) + (iteration << 24) + (kind << 20) + index)      kind 0 = real batch of class `index`, 1 = synthetic
//                                                                             graph for class `index`, 2 = inner step `index`
//   word = mix64(key + (row << 8) + (column >> 2))                            row = the node's id in the graph (real) / the synthetic row
//   kept = ((word >> 16 (column & 3)) & 0xffff) < thr,  thr = round((1 - p) 65536)
// (splitmix64's finaliser; hidden <= 1024).  graphslim_amd/engine.py dropout_keep restates it for the parity tests.
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ unsigned long long gs_mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// key0 = mix64(seed) + (iteration << 24) + (kind << 20): the kernels add the index
__host__ __device__ __forceinline__ unsigned long long gs_drop_key0(unsigned long long seed, unsigned long long iter, int kind) {
  return gs_mix64(seed) + (iter << 24) + ((unsigned long long)kind << 20);
}
// multipliers of columns col4 .. col4 + 3 (col4 % 4 == 0) of `row`
__device__ __forceinline__ float4 gs_drop_mult4(unsigned long long key, long row, int col4, unsigned thr, float q) {
  const unsigned long long w = gs_mix64(key + ((unsigned long long)row << 8) + (unsigned)(col4 >> 2));
  float4 m;
  m.x = ((unsigned)(w) & 0xffffu) < thr ? q : 0.f;
  m.y = ((unsigned)(w >> 16) & 0xffffu) < thr ? q : 0.f;
  m.z = ((unsigned)(w >> 32) & 0xffffu) < thr ? q : 0.f;
  m.w = ((unsigned)(w >> 48) & 0xffffu) < thr ? q : 0.f;
  return m;
}
__device__ __forceinline__ float gs_drop_mult1(unsigned long long key, long row, int col, unsigned thr, float q) {
  const unsigned long long w = gs_mix64(key + ((unsigned long long)row << 8) + (unsigned)(col >> 2));
  return ((unsigned)(w >> (16 * (col & 3))) & 0xffffu) < thr ? q : 0.f;
}

// out[c][i][:] = H[i][:] * multiplier(kind, class0 + c, row i)     (one masked copy of the shared hidden activations per class; Cl = 1 with
// out == H: the inner step's mask in place).  h % 4 == 0.
__global__ __launch_bounds__(256) void gs_drop_rows_kernel(const float* __restrict__ H, float* __restrict__ out, int n, int h, int Cl,
                                                           unsigned long long key0, int index0, unsigned thr, float q) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (c, row, col4)
  const int h4 = h >> 2;
  if (i >= (long)Cl * n * h4) return;
  const int c = (int)(i / ((long)n * h4));
  const long r = i - (long)c * n * h4;
  const int row = (int)(r / h4), col4 = 4 * (int)(r - (long)row * h4);
  const float4 m = gs_drop_mult4(gs_mix64(key0 + (unsigned)(index0 + c)), row, col4, thr, q);
  const float4 v = *reinterpret_cast<const float4*>(H + (long)row * h + col4);
  *reinterpret_cast<float4*>(out + ((long)c * n + row) * h + col4) = make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w);
}
// buf[c][i][:] *= multiplier(kind, class0 + c, row i)     (per-class activations masked in place: the hidden layers of the GCN
// condensation model under dropout, models/base.py:59; `kind` carries the layer, gs_gcnk.inc)
__global__ __launch_bounds__(256) void gs_drop_rows_self_kernel(float* __restrict__ buf, int n, int h, int Cl, unsigned long long key0,
                                                                int index0, unsigned thr, float q) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (c, row, col4)
  const int h4 = h >> 2;
  if (i >= (long)Cl * n * h4) return;
  const int c = (int)(i / ((long)n * h4));
  const long r = i - (long)c * n * h4;
  const int row = (int)(r / h4), col4 = 4 * (int)(r - (long)row * h4);
  const float4 m = gs_drop_mult4(gs_mix64(key0 + (unsigned)(index0 + c)), row, col4, thr, q);
  float4* p = reinterpret_cast<float4*>(buf + ((long)c * n + row) * h + col4);
  const float4 v = *p;
  *p = make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w);
}
// buf[c][l][:] *= multiplier(kind, class0 + c, node ids[c][l]) for the live rows l < live[c] of class c's frontier (capacity `cap` rows):
// the real batch's hidden activations, keyed by the GLOBAL node id of the row like the SGC path's gathered rows
__global__ __launch_bounds__(256) void gs_drop_rows_ids_kernel(float* __restrict__ buf, const int* __restrict__ ids,
                                                               const int* __restrict__ live, int cap, int h, int Cl,
                                                               unsigned long long key0, int index0, unsigned thr, float q) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (c, row, col4)
  const int h4 = h >> 2;
  if (i >= (long)Cl * cap * h4) return;
  const int c = (int)(i / ((long)cap * h4));
  const long r = i - (long)c * cap * h4;
  const int row = (int)(r / h4), col4 = 4 * (int)(r - (long)row * h4);
  if (row >= live[c]) return;
  const float4 m = gs_drop_mult4(gs_mix64(key0 + (unsigned)(index0 + c)), ids[(long)c * cap + row], col4, thr, q);
  float4* p = reinterpret_cast<float4*>(buf + ((long)c * cap + row) * h + col4);
  const float4 v = *p;
  *p = make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w);
}
// buf = [Hc > 0] * q * buf   (the backward of relu + dropout: Hc = relu(.) * multiplier, so Hc > 0 <=> passed by both)
__global__ __launch_bounds__(256) void gs_drop_mask_scale_kernel(float* __restrict__ buf, const float* __restrict__ Hc, long total, float q) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) buf[i] = Hc[i] > 0.f ? buf[i] * q : 0.f;
}
// out[i] = sum_c [Hc[c][i] > 0] * q * tmp[c][i]      (classes in order: a fixed summation order)
__global__ __launch_bounds__(256) void gs_drop_mask_reduce_kernel(const float* __restrict__ tmp, const float* __restrict__ Hc, long nh, int Cl,
                                                                  float q, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nh) return;
  float s = 0.f;
  for (int c = 0; c < Cl; ++c) s += Hc[(long)c * nh + i] > 0.f ? tmp[(long)c * nh + i] * q : 0.f;
  out[i] = s;
}

// ------------------------------------------------------------------------------------------------
// ELL block aggregation with gather (sampled-block SpMM, models/sgc.py:46-47 on the real batch).
// out[(c,row), :] = sum_j val[(c,row,j)] * src[(base_c + col[(c,row,j)]) * ld_src + :]
// rs_out[(c,row)]  = sum_j val * (rs_in ? rs_in[base_c + col] : 1)
// rows >= nrows[c] are written as zeros so that fixed-capacity batched GEMMs downstream see exact
// zeros in the padding.  One wave per row, VEC floats per lane per access, f gathers in flight.
// ------------------------------------------------------------------------------------------------
// sign bits of one gathered 256-wide row held as 4 consecutive columns per lane: the lane's nibble goes to bit 4 (lane & 7) of word
// lane >> 3, the eight lanes of a word are OR-ed with three shuffles, lane 8 w stores word w (32 contiguous bytes per row)
__device__ __forceinline__ void gs_ell_put_bits(unsigned* __restrict__ dst, const float4& x, int lane) {
  unsigned v = ((x.x > 0.f ? 1u : 0u) | (x.y > 0.f ? 2u : 0u) | (x.z > 0.f ? 4u : 0u) | (x.w > 0.f ? 8u : 0u)) << (4 * (lane & 7));
  v |= __shfl_xor(v, 1, 64);
  v |= __shfl_xor(v, 2, 64);
  v |= __shfl_xor(v, 4, 64);
  if ((lane & 7) == 0) dst[lane >> 3] = v;
}

template <int VEC, bool DROP = false>
__global__ __launch_bounds__(256) void gs_ell_spmm_kernel(const int* __restrict__ col, const float* __restrict__ val,
                                                          const int* __restrict__ nrows, int cap_rows, int f,
                                                          long col_base_stride, const float* __restrict__ src, long ld_src,
                                                          const float* __restrict__ rs_in, int F, float* __restrict__ out,
                                                          long ld_out, float* __restrict__ rs_out, int nclass,
                                                          unsigned* __restrict__ ebits = nullptr, int xcd_map = 0,
                                                          unsigned long long dkey0 = 0, int dclass0 = 0, unsigned dthr = 0, float dq = 1.f,
                                                          int pad_zero = 1) {
  // pad_zero = 0: rows >= nrows[c] are left unwritten — the output is only ever read through the next gather's live indices (the
  // outermost and middle hops of the real side: a third of the capacity rows are padding, round 5).
  // DROP (VEC = 4 only): every gathered source row is multiplied by the dropout mask of (class dclass0 + c, source row, column) before it
  // is summed — the hop-2 gather of H1 = relu(X W1 + b1) under models/sgc.py:42 (see gs_drop_mult4).
  // ebits (VEC = 4, F = 256, optional): per edge slot (grow * f + j) the sign bits of the gathered source row, bit (n & 31) of word
  // n >> 5 = [src[col_j][n] > 0] — 8 words = 32 B.  The hop-2 gather has every source row of H1 = relu(X W1 + b1) in registers; the
  // edge product then reads its ReLU mask from these 32 B per edge instead of gathering the 1-KiB fp32 row again.
  // (VEC = 4 needs ld_src, ld_out multiples of 4 that cover F rounded up to 4 — the engine pads its row strides, so that widths
  // like 602 (reddit) or 1 433 (cora) take the 16-byte path too: the columns [F, round4(F)) carry zeros from X's padding)
  const int lane = threadIdx.x & 63;
  long grow = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;   // (c, row)
  int c, row;
  if (xcd_map) {
    // class -> XCD affinity (round 5; speed only, never correctness): workgroups are dealt round-robin over the 8 XCDs, so workgroup b
    // lands on the XCD of b % 8 — the rows of class c all go to workgroups with b % 8 = c % 8 (cap_rows % 4 == 0: a workgroup's 4 waves
    // are 4 rows of one class).  A class's gathered source rows repeat 1.6x inside the class (ogbn-arxiv: 8 700 gathers, 5 400 distinct)
    // and its output rows are the next hop's sources: both now meet in ONE XCD's L2 instead of being spread over eight.
    const int b = blockIdx.x, x = b & 7, i = b >> 3, rq = cap_rows >> 2;
    const int slot = i / rq;
    c = x + 8 * slot;
    row = 4 * (i - slot * rq) + (threadIdx.x >> 6);
    if (c >= nclass) return;
    grow = (long)c * cap_rows + row;
  } else {
    if (grow >= (long)nclass * cap_rows) return;
    c = (int)(grow / cap_rows);
    row = (int)(grow - (long)c * cap_rows);
  }
  float* o = out + grow * ld_out;
  if (row >= nrows[c]) {
    if (!pad_zero) return;
    for (int c0 = lane * VEC; c0 < F; c0 += 64 * VEC)
      for (int q = 0; q < VEC; ++q)
        if (c0 + q < (VEC == 4 ? (int)ld_out : F)) o[c0 + q] = 0.f;
    if (lane == 0 && rs_out) rs_out[grow] = 0.f;
    return;
  }
  const long ebase = grow * f, cb = (long)c * col_base_stride;
  if (lane == 0 && rs_out) {
    float s = 0.f;
    for (int j = 0; j < f; ++j) s += val[ebase + j] * (rs_in ? rs_in[cb + col[ebase + j]] : 1.f);
    rs_out[grow] = s;
  }
  // the row's (col, val) pairs are fetched once by lanes 0..f-1 and read back with v_readlane (wave-uniform j; works when
  // only some lanes run the column loop, F < 256); the gathers go out four at a time before their FMAs (same j order as a
  // plain loop: identical sums)
  const int ecol = (lane < f) ? col[ebase + lane] : 0;
  const int evalb = (lane < f) ? __float_as_int(val[ebase + lane]) : 0;
  const unsigned long long dkey = DROP ? gs_mix64(dkey0 + (unsigned)(dclass0 + c)) : 0ull;
  for (int c0 = lane * VEC; c0 < F; c0 += 64 * VEC) {
    if (VEC == 4) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      // up to eight gathers in flight before their FMAs (round 6: fan-outs of 5 … 8 took a second, dependent round of loads behind the
      // first four); the predicates are wave-uniform, the FMA order is j = 0, 1, .. as before: identical sums
      for (int j = 0; j < f; j += 8) {
        float w[8];
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (j + u < f) {
            w[u] = __int_as_float(__builtin_amdgcn_readlane(evalb, j + u));
            x[u] = *reinterpret_cast<const float4*>(src + (cb + __builtin_amdgcn_readlane(ecol, j + u)) * ld_src + c0);
            if (DROP) {
              const float4 m = gs_drop_mult4(dkey, cb + __builtin_amdgcn_readlane(ecol, j + u), c0, dthr, dq);
              x[u].x *= m.x; x[u].y *= m.y; x[u].z *= m.z; x[u].w *= m.w;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (j + u < f) {
            acc.x = fmaf(w[u], x[u].x, acc.x);
            acc.y = fmaf(w[u], x[u].y, acc.y);
            acc.z = fmaf(w[u], x[u].z, acc.z);
            acc.w = fmaf(w[u], x[u].w, acc.w);
            if (ebits) gs_ell_put_bits(ebits + (ebase + j + u) * 8, x[u], lane);
          }
        }
      }
      *reinterpret_cast<float4*>(o + c0) = acc;
    } else {
      float acc = 0.f;
      for (int j = 0; j < f; ++j) acc = fmaf(val[ebase + j], src[(cb + col[ebase + j]) * ld_src + c0], acc);
      o[c0] = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Real side: Z = S Wl (from the GEMM) + rs * bl;  G = (softmax(Z) - onehot(c)) / B_c, in place.
// gcond_base.py:220-221 (log_softmax + nll_loss mean over the B seeds) differentiated wrt Z.
// Padding rows (b >= n_seed[c]) -> 0.  One wave per row; C <= 64*8.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gs_real_softmax_grad_kernel(float* __restrict__ Z, const float* __restrict__ rs,
                                                                   const float* __restrict__ bl,
                                                                   const int* __restrict__ n_seed, int bcap, int C,
                                                                   int class0, int nclass_local) {
  const int lane = threadIdx.x & 63;
  const long grow = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (grow >= (long)nclass_local * bcap) return;
  const int c = (int)(grow / bcap), b = (int)(grow - (long)c * bcap);
  float* z = Z + grow * C;
  const int B = n_seed[c];
  if (b >= B) {
    for (int k = lane; k < C; k += 64) z[k] = 0.f;
    return;
  }
  const float r = rs[grow];
  float mx = -INFINITY;
  for (int k = lane; k < C; k += 64) mx = fmaxf(mx, z[k] + r * bl[k]);
  mx = gs_wave_max(mx);
  float se = 0.f;
  for (int k = lane; k < C; k += 64) se += expf(z[k] + r * bl[k] - mx);
  se = gs_wave_sum(se);
  const float inv = 1.f / se, invB = 1.f / (float)B;
  for (int k = lane; k < C; k += 64) {
    float p = expf(z[k] + r * bl[k] - mx) * inv;
    z[k] = (p - (k == class0 + c ? 1.f : 0.f)) * invB;
  }
}

// Synthetic side: P = softmax(S Wl + rs bl) kept, G = (P - onehot(label)) / n_label.
// gcond_base.py:226-230 (loss on the rows of class c only, mean over n_c).
__global__ __launch_bounds__(256) void gs_syn_softmax_grad_kernel(const float* __restrict__ Z, const float* __restrict__ rs,
                                                                  const float* __restrict__ bl,
                                                                  const int* __restrict__ labels,
                                                                  const int* __restrict__ syn_lo,
                                                                  const int* __restrict__ syn_hi, int n, int C,
                                                                  float* __restrict__ P, float* __restrict__ G,
                                                                  int divide_by_n /* inner loop: mean over all n */) {
  const int lane = threadIdx.x & 63;
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n) return;
  const float* z = Z + i * C;
  const int y = labels[i];
  const float r = rs[i];
  float mx = -INFINITY;
  for (int k = lane; k < C; k += 64) mx = fmaxf(mx, z[k] + r * bl[k]);
  mx = gs_wave_max(mx);
  float se = 0.f;
  for (int k = lane; k < C; k += 64) se += expf(z[k] + r * bl[k] - mx);
  se = gs_wave_sum(se);
  const float inv = 1.f / se;
  const float invn = 1.f / (float)(divide_by_n ? n : (syn_hi[y] - syn_lo[y]));
  for (int k = lane; k < C; k += 64) {
    float p = expf(z[k] + r * bl[k] - mx) * inv;
    if (P) P[i * C + k] = p;
    G[i * C + k] = (p - (k == y ? 1.f : 0.f)) * invn;
  }
}

// V_i = (P_i * U_i - P_i (P_i . U_i)) / n_c with U_i = Ug_i + rs_i * Dbl_c   (softmax backward of
// d loss / d G routed through G = (P - Y)/n_c).   drs_i = G_i . Dbl_c + V_i . bl.
__global__ __launch_bounds__(256) void gs_syn_v_kernel(const float* __restrict__ P, const float* __restrict__ G,
                                                       const float* __restrict__ Ug, const float* __restrict__ rs,
                                                       const float* __restrict__ D, long P_total, long off_bl,
                                                       const float* __restrict__ bl, const int* __restrict__ labels,
                                                       const int* __restrict__ syn_lo, const int* __restrict__ syn_hi, int n,
                                                       int C, int c_lo, int c_hi, float* __restrict__ V,
                                                       float* __restrict__ drs, const float* __restrict__ loss_c = nullptr,
                                                       int n_loss = 0, double* __restrict__ loss_sum = nullptr,
                                                       float* __restrict__ loss_last = nullptr, long* __restrict__ steps = nullptr,
                                                       const double* __restrict__ parts = nullptr, int n_parts = 0,
                                                       float* __restrict__ loss_out = nullptr) {
  // parts (optional): [C][n_parts] shares of the class losses left by gs_dist_fused_kernel; loss_out = the loss_c buffer to finish
  // loss_c (optional): the iteration's loss bookkeeping (gs_loss_accum_kernel's arithmetic) by the first thread of the launch — the
  // class losses are final before this kernel (unsharded engines inside gs_gcond_outer_step: one launch less at the end of the stage)
  if (parts && blockIdx.x == 0 && threadIdx.x == 0) {      // the classes' losses from the distance kernel's shares (workgroup order)
    for (int c = c_lo; c < c_hi; ++c) {
      double tot = 0.0;
      for (int q = 0; q < n_parts; ++q) tot += parts[(long)c * n_parts + q];
      const int ncl = syn_hi[c] - syn_lo[c];
      loss_out[c] = ncl == 0 ? 0.f : (float)ncl / (float)n * (float)tot;
    }
  }
  if (loss_c && blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.f;
    for (int c = 0; c < n_loss; ++c) s += parts ? loss_out[c] : loss_c[c];
    *loss_last = s;
    *loss_sum += (double)s;
    *steps += 1;
  }
  const int lane = threadIdx.x & 63;
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n) return;
  const int y = labels[i];
  if (y < c_lo || y >= c_hi) {     // rows of classes owned by another rank: no contribution from this rank
    for (int k = lane; k < C; k += 64) V[i * C + k] = 0.f;
    if (lane == 0) drs[i] = 0.f;
    return;
  }
  const float* Db = D + (long)y * P_total + off_bl;
  const float r = rs[i];
  const float invn = 1.f / (float)(syn_hi[y] - syn_lo[y]);
  float pu = 0.f, gd = 0.f;
  for (int k = lane; k < C; k += 64) {
    float u = Ug[i * C + k] + r * Db[k];
    pu += P[i * C + k] * u;
    gd += G[i * C + k] * Db[k];
  }
  pu = gs_wave_sum(pu);
  gd = gs_wave_sum(gd);
  float vb = 0.f;
  for (int k = lane; k < C; k += 64) {
    float p = P[i * C + k];
    float u = Ug[i * C + k] + r * Db[k];
    float v = (p * u - p * pu) * invn;
    V[i * C + k] = v;
    vb += v * bl[k];
  }
  vb = gs_wave_sum(vb);
  if (lane == 0) drs[i] = gd + vb;
}

// M[i, :] += v[i]  (the rank-one  drs 1^T  term of dQ);  rowsum: out[i] = sum_j M[i, j]
__global__ void gs_add_rowvec_kernel(float* __restrict__ M, const float* __restrict__ v, int rows, int cols) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)rows * cols) return;
  M[idx] += v[idx / cols];
}
__global__ __launch_bounds__(256) void gs_rowsum_kernel(const float* __restrict__ M, int rows, int cols, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= rows) return;
  float s = 0.f;
  for (int j = lane; j < cols; j += 64) s += M[i * cols + j];
  s = gs_wave_sum(s);
  if (lane == 0) out[i] = s;
}

// ------------------------------------------------------------------------------------------------
// match_loss statistics (condensation/utils.py:12-106).  Parameter p of class c is viewed as a
// [rows x cols] matrix (a bias is [len x 1]); per column: dot = <r,s>, nr2 = |r|^2, ns2 = |s|^2.
// grid = (nparams, C); block = 256 = CB column lanes x (256/CB) row groups.
// ------------------------------------------------------------------------------------------------
struct gs_pdesc {
  int nparams;
  int rows[8], cols[8];
  long off[8];      // offset inside the flat parameter vector
  int col_off[8];   // offset of this parameter's columns inside the per-class stats row
  int is_bias[8];
  int total_cols;
};

typedef float gs_f32x4u __attribute__((ext_vector_type(4), aligned(4)));

__global__ __launch_bounds__(256) void gs_dist_stats_kernel(const float* __restrict__ gw_real, const float* __restrict__ gw_syn,
                                                            long P_total, gs_pdesc pd, float* __restrict__ stats /*[C][total_cols][3]*/) {
  const int p = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
  const int rows = pd.rows[p], cols = pd.cols[p];
  const float* R = gw_real + (long)c * P_total + pd.off[p];
  const float* S = gw_syn + (long)c * P_total + pd.off[p];
  __shared__ float sh[3][256 * 4];
  if ((cols & 3) == 0 && cols >= 16) {
    // wide parameter (W1: d x h, Wl: h x C): a thread owns 4 adjacent columns (16-byte loads) and every (256 / (cols / 4))-th
    // row; the row groups are summed through LDS in a fixed order.  (The first form walked all `rows` with one 4-byte load per
    // step and thread: 37-51 us on the critical path right after the stream join.)
    const int Q = cols >> 2;
    int CB = 1;
    while (CB < Q && CB < 256) CB <<= 1;
    const int tx = tid % CB, ty = tid / CB, nty = 256 / CB;
    for (int q0 = 0; q0 < Q; q0 += CB) {
      const int q = q0 + tx;
      float dot[4] = {0.f, 0.f, 0.f, 0.f}, nr2[4] = {0.f, 0.f, 0.f, 0.f}, ns2[4] = {0.f, 0.f, 0.f, 0.f};
      if (q < Q)
#pragma unroll 4
        for (int i = ty; i < rows; i += nty) {
          const gs_f32x4u r = *reinterpret_cast<const gs_f32x4u*>(R + (long)i * cols + 4 * q);
          const gs_f32x4u s_ = *reinterpret_cast<const gs_f32x4u*>(S + (long)i * cols + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dot[e] = fmaf(r[e], s_[e], dot[e]);
            nr2[e] = fmaf(r[e], r[e], nr2[e]);
            ns2[e] = fmaf(s_[e], s_[e], ns2[e]);
          }
        }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sh[0][tid * 4 + e] = dot[e];
        sh[1][tid * 4 + e] = nr2[e];
        sh[2][tid * 4 + e] = ns2[e];
      }
      __syncthreads();
      if (ty == 0 && q < Q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = dot[e], b = nr2[e], d_ = ns2[e];
          for (int g = 1; g < nty; ++g) {
            a += sh[0][(g * CB + tx) * 4 + e];
            b += sh[1][(g * CB + tx) * 4 + e];
            d_ += sh[2][(g * CB + tx) * 4 + e];
          }
          float* st = stats + ((long)c * pd.total_cols + pd.col_off[p] + 4 * q + e) * 3;
          st[0] = a;
          st[1] = b;
          st[2] = d_;
        }
      }
      __syncthreads();
    }
    return;
  }
  int CB = 1;
  while (CB < cols && CB < 256) CB <<= 1;
  const int tx = tid % CB, ty = tid / CB, nty = 256 / CB;
  for (int k0 = 0; k0 < cols; k0 += CB) {
    const int k = k0 + tx;
    float dot = 0.f, nr2 = 0.f, ns2 = 0.f;
    if (k < cols)
      for (int i = ty; i < rows; i += nty) {
        float r = R[(long)i * cols + k], s = S[(long)i * cols + k];
        dot = fmaf(r, s, dot);
        nr2 = fmaf(r, r, nr2);
        ns2 = fmaf(s, s, ns2);
      }
    sh[0][tid] = dot;
    sh[1][tid] = nr2;
    sh[2][tid] = ns2;
    __syncthreads();
    if (ty == 0 && k < cols) {
      for (int q = 1; q < nty; ++q) {
        dot += sh[0][q * CB + tx];
        nr2 += sh[1][q * CB + tx];
        ns2 += sh[2][q * CB + tx];
      }
      float* st = stats + ((long)c * pd.total_cols + pd.col_off[p] + k) * 3;
      st[0] = dot;
      st[1] = nr2;
      st[2] = ns2;
    }
    __syncthreads();
  }
}

// per class: loss_c and per-column (alpha, beta) with D = alpha*r + beta*s, both scaled by coeff_c.
// metric 0 'ours' (utils.py:33-39,86-104), 1 'mse' (:41-49), 2 'cos' (:51-60).
// colscale2 (optional, with_bn): s_j^2 per column of parameter 0 (W1).  The engine then works with the folded weight W1' = W1 diag(s)
// (s = bn.weight / sqrt(1 + eps)), whose gradient columns are 1 / s_j times the true ones: the statistics of the TRUE gradients are
// s_j^2 times those of the folded ones (the 1e-6 on the norm product makes the cosine distance not exactly scale-free), and
// d loss / d gw_syn' = s_j * d loss / d gw_syn = s_j^2 (a r' + b s').
__global__ __launch_bounds__(256) void gs_dist_finalize_kernel(const float* __restrict__ stats, gs_pdesc pd, int metric,
                                                               const int* __restrict__ syn_lo, const int* __restrict__ syn_hi,
                                                               int n_syn, float* __restrict__ ab /*[C][total_cols][2]*/,
                                                               float* __restrict__ loss_c /*[C]*/,
                                                               const float* __restrict__ colscale2 = nullptr) {
  const int c = blockIdx.x, tid = threadIdx.x;
  const int nc = syn_hi[c] - syn_lo[c];
  const float coeff = (float)nc / (float)n_syn;     // gcond_base.py:237
  const float* st = stats + (long)c * pd.total_cols * 3;
  float* o = ab + (long)c * pd.total_cols * 2;
  __shared__ double red[3][256];
  double l = 0.0, sd = 0.0, sr = 0.0, ss = 0.0;
  for (int p = 0; p < pd.nparams; ++p)
    for (int k = tid; k < pd.cols[p]; k += 256) {
      const int ci = pd.col_off[p] + k;
      float dot = st[ci * 3], nr2 = st[ci * 3 + 1], ns2 = st[ci * 3 + 2];
      const float sc2 = (colscale2 && p == 0) ? colscale2[k] : 1.f;
      dot *= sc2; nr2 *= sc2; ns2 *= sc2;
      if (metric == 0) {
        float a = 0.f, b = 0.f;
        if (!pd.is_bias[p]) {
          float nr = sqrtf(nr2), ns = sqrtf(ns2);
          float q = nr * ns + 1e-6f;
          l += 1.0 - (double)(dot / q);
          a = -1.f / q;
          b = ns > 0.f ? dot * nr / (ns * q * q) : 0.f;
        }
        o[ci * 2] = a * coeff * sc2;
        o[ci * 2 + 1] = b * coeff * sc2;
      } else if (metric == 1) {
        l += (double)ns2 - 2.0 * (double)dot + (double)nr2;
        o[ci * 2] = -2.f * coeff;
        o[ci * 2 + 1] = 2.f * coeff;
      } else {
        sd += dot;
        sr += nr2;
        ss += ns2;
      }
    }
  if (nc == 0) {   // class without synthetic nodes: no term (the reference never builds one, gcond_base.py:104-107)
    for (int k = tid; k < pd.total_cols; k += 256) { o[k * 2] = 0.f; o[k * 2 + 1] = 0.f; }
    if (tid == 0) loss_c[c] = 0.f;
    return;
  }
  red[0][tid] = metric == 2 ? sd : l;
  red[1][tid] = sr;
  red[2][tid] = ss;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      red[0][tid] += red[0][tid + s];
      red[1][tid] += red[1][tid + s];
      red[2][tid] += red[2][tid + s];
    }
    __syncthreads();
  }
  if (metric == 2) {
    float dot = (float)red[0][0], nr = sqrtf((float)red[1][0]), ns = sqrtf((float)red[2][0]);
    float q = nr * ns + 1e-6f;
    float a = -1.f / q, b = ns > 0.f ? dot * nr / (ns * q * q) : 0.f;
    for (int k = tid; k < pd.total_cols; k += 256) { o[k * 2] = a * coeff; o[k * 2 + 1] = b * coeff; }
    if (tid == 0) loss_c[c] = coeff * (1.f - dot / q);
  } else if (tid == 0) {
    loss_c[c] = coeff * (float)red[0][0];
  }
}

// D[c][p][i][k] = alpha[c][col] * r + beta[c][col] * s
__global__ void gs_dist_apply_kernel(const float* __restrict__ gw_real, const float* __restrict__ gw_syn, long P_total,
                                     gs_pdesc pd, const float* __restrict__ ab, float* __restrict__ D, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)C * P_total) return;
  const int c = (int)(idx / P_total);
  const long e = idx - (long)c * P_total;
  int p = 0;
  while (p + 1 < pd.nparams && e >= pd.off[p + 1]) ++p;
  const int k = (int)((e - pd.off[p]) % pd.cols[p]);
  const float* o = ab + ((long)c * pd.total_cols + pd.col_off[p] + k) * 2;
  D[idx] = o[0] * gw_real[idx] + o[1] * gw_syn[idx];
}

// The three launches above as ONE.  grid (classes, G): for 'ours' / 'mse' every column's (alpha, beta) depends on that column's own
// statistics only, so a class is shared by G = GS_DIST_SPLIT workgroups of 1024 threads — workgroups 0 .. G-2 take column ranges of
// parameter 0 (W1, 3/4 of the vector), workgroup G-1 every other parameter — each running statistics -> (alpha, beta) -> D on ITS
// columns (W1's share stays in registers between the two passes) and leaving its share of the class's loss in `part`: gs_syn_v_kernel
// adds the G shares in workgroup order (fixed order, no float atomics) and writes loss_c.  'cos' couples all columns: G = 1.  A class's ~170 KB go through G CUs instead of one (one CU: 24 us for 40 classes).
// Same formulas as the three-launch form; the column sums are grouped differently (1024 / CB row groups), so values agree with it to
// fp32 re-association.  Requires total_cols <= GS_DIST_FUSED_COLS.
#define GS_DIST_FUSED_COLS 1024
#define GS_DIST_SPLIT 4
__global__ __launch_bounds__(1024) void gs_dist_fused_kernel(const float* __restrict__ gw_real, const float* __restrict__ gw_syn,
                                                             long P_total, gs_pdesc pd, int metric, const int* __restrict__ syn_lo,
                                                             const int* __restrict__ syn_hi, int n_syn, float* __restrict__ ab,
                                                             float* __restrict__ loss_c, float* __restrict__ D,
                                                             const float* __restrict__ colscale2, double* __restrict__ part) {
  const int c = blockIdx.x, g = blockIdx.y, G = gridDim.y, tid = threadIdx.x;
  const float* R0 = gw_real + (long)c * P_total;
  const float* S0 = gw_syn + (long)c * P_total;
  __shared__ float sh[3][1024 * 4];
  __shared__ float st[GS_DIST_FUSED_COLS][3];
  __shared__ float abl[GS_DIST_FUSED_COLS][2];
  __shared__ double red[3][1024];
  const int nc = syn_hi[c] - syn_lo[c];
  const float coeff = (float)nc / (float)n_syn;
  // this workgroup's columns of parameter p: [k_lo, k_hi) (multiples of 4 inside a wide parameter)
  auto range = [&](int p, int& k_lo, int& k_hi) {
    const int cols = pd.cols[p];
    k_lo = 0; k_hi = cols;
    if (G == 1) return;
    if (p == 0 && pd.nparams > 1) {
      if (g == G - 1) { k_hi = 0; return; }
      const int Q = (cols + 3) >> 2, per = (Q + G - 2) / (G - 1);
      k_lo = min(cols, 4 * per * g); k_hi = min(cols, 4 * per * (g + 1));
    } else if (g != G - 1) {
      k_hi = 0;
    }
  };
  // ---- column statistics into LDS
  bool keep = false;                   // (block-uniform) parameter 0's share of this workgroup stays in keep_r / keep_s
  gs_f32x4u keep_r[4], keep_s[4];
  for (int p = 0; p < pd.nparams; ++p) {
    const int rows = pd.rows[p], cols = pd.cols[p];
    int k_lo, k_hi;
    range(p, k_lo, k_hi);
    if (k_hi <= k_lo) continue;      // (block-uniform)
    const float* R = R0 + pd.off[p];
    const float* S = S0 + pd.off[p];
    if (metric == 0 && pd.is_bias[p]) {      // 'ours' skips one-dimensional parameters (utils.py:86-88): (alpha, beta) = 0 below
      for (int k = k_lo + tid; k < k_hi; k += 1024) { st[pd.col_off[p] + k][0] = 0.f; st[pd.col_off[p] + k][1] = 0.f; st[pd.col_off[p] + k][2] = 0.f; }
      continue;
    }
    if ((cols & 3) == 0 && cols >= 16) {
      const int q_lo = k_lo >> 2, Q = (k_hi - k_lo) >> 2;
      int CB = 1;
      while (CB < Q && CB < 1024) CB <<= 1;
      const int tx = tid % CB, ty = tid / CB, nty = 1024 / CB;
      if (p == 0 && G > 1 && Q <= CB && rows <= 4 * nty) {
        // this workgroup's whole share of W1 fits its registers (<= 4 rows x 4 columns per thread, both vectors): D is formed from
        // them below instead of from a second pass over memory
        keep = true;
        float dot[4] = {0.f, 0.f, 0.f, 0.f}, nr2[4] = {0.f, 0.f, 0.f, 0.f}, ns2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = ty + j * nty;
          if (tx < Q && i < rows) {
            keep_r[j] = *reinterpret_cast<const gs_f32x4u*>(R + (long)i * cols + 4 * (q_lo + tx));
            keep_s[j] = *reinterpret_cast<const gs_f32x4u*>(S + (long)i * cols + 4 * (q_lo + tx));
          } else {
            keep_r[j] = gs_f32x4u{0.f, 0.f, 0.f, 0.f};
            keep_s[j] = gs_f32x4u{0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)      // (row order ty, ty + nty, ...: the order of the loop below)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dot[e] = fmaf(keep_r[j][e], keep_s[j][e], dot[e]);
            nr2[e] = fmaf(keep_r[j][e], keep_r[j][e], nr2[e]);
            ns2[e] = fmaf(keep_s[j][e], keep_s[j][e], ns2[e]);
          }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sh[0][tid * 4 + e] = dot[e];
          sh[1][tid * 4 + e] = nr2[e];
          sh[2][tid * 4 + e] = ns2[e];
        }
        __syncthreads();
        for (int u = tid; u < CB * 12; u += 1024) {
          const int qq = u / 12, w = u - qq * 12, which = w >> 2, e = w & 3;
          if (qq < Q) {
            float a = 0.f;
            for (int gq = 0; gq < nty; ++gq) a += sh[which][(gq * CB + qq) * 4 + e];
            st[pd.col_off[p] + 4 * (q_lo + qq) + e][which] = a;
          }
        }
        __syncthreads();
        continue;
      }
      for (int q0 = 0; q0 < Q; q0 += CB) {
        const int q = q0 + tx;
        float dot[4] = {0.f, 0.f, 0.f, 0.f}, nr2[4] = {0.f, 0.f, 0.f, 0.f}, ns2[4] = {0.f, 0.f, 0.f, 0.f};
        if (q < Q)
#pragma unroll 4
          for (int i = ty; i < rows; i += nty) {
            const gs_f32x4u r = *reinterpret_cast<const gs_f32x4u*>(R + (long)i * cols + 4 * (q_lo + q));
            const gs_f32x4u s_ = *reinterpret_cast<const gs_f32x4u*>(S + (long)i * cols + 4 * (q_lo + q));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              dot[e] = fmaf(r[e], s_[e], dot[e]);
              nr2[e] = fmaf(r[e], r[e], nr2[e]);
              ns2[e] = fmaf(s_[e], s_[e], ns2[e]);
            }
          }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sh[0][tid * 4 + e] = dot[e];
          sh[1][tid * 4 + e] = nr2[e];
          sh[2][tid * 4 + e] = ns2[e];
        }
        __syncthreads();
        // 12 sums per column quad, each over the nty row groups in order: one thread per (quad, statistic, element)
        for (int u = tid; u < CB * 12; u += 1024) {
          const int qq = u / 12, w = u - qq * 12, which = w >> 2, e = w & 3;
          if (q0 + qq < Q) {
            float a = 0.f;
            for (int gq = 0; gq < nty; ++gq) a += sh[which][(gq * CB + qq) * 4 + e];
            st[pd.col_off[p] + 4 * (q_lo + q0 + qq) + e][which] = a;
          }
        }
        __syncthreads();
      }
    } else {
      // narrow parameter (a bias is [len x 1]): every thread takes strided rows of one column, tree sum over the block in a fixed order
      for (int k = k_lo; k < k_hi; ++k) {
        float dot = 0.f, nr2 = 0.f, ns2 = 0.f;
        for (int i = tid; i < rows; i += 1024) {
          const float r = R[(long)i * cols + k], s = S[(long)i * cols + k];
          dot = fmaf(r, s, dot);
          nr2 = fmaf(r, r, nr2);
          ns2 = fmaf(s, s, ns2);
        }
        sh[0][tid] = dot; sh[1][tid] = nr2; sh[2][tid] = ns2;
        __syncthreads();
        for (int w = 512; w > 0; w >>= 1) {
          if (tid < w) { sh[0][tid] += sh[0][tid + w]; sh[1][tid] += sh[1][tid + w]; sh[2][tid] += sh[2][tid + w]; }
          __syncthreads();
        }
        if (tid < 3) st[pd.col_off[p] + k][tid] = sh[tid][0];
        __syncthreads();
      }
    }
  }
  __syncthreads();
  // ---- (alpha, beta) per column and this workgroup's share of the class's loss (the arithmetic of gs_dist_finalize_kernel)
  double l = 0.0, sd = 0.0, sr = 0.0, ss = 0.0;
  for (int p = 0; p < pd.nparams; ++p) {
    int k_lo, k_hi;
    range(p, k_lo, k_hi);
    for (int k = k_lo + tid; k < k_hi; k += 1024) {
      const int ci = pd.col_off[p] + k;
      float dot = st[ci][0], nr2 = st[ci][1], ns2 = st[ci][2];
      const float sc2 = (colscale2 && p == 0) ? colscale2[k] : 1.f;
      dot *= sc2; nr2 *= sc2; ns2 *= sc2;
      float a = 0.f, b = 0.f;
      if (metric == 0) {
        if (!pd.is_bias[p]) {
          const float nr = sqrtf(nr2), ns = sqrtf(ns2);
          const float q = nr * ns + 1e-6f;
          l += 1.0 - (double)(dot / q);
          a = -1.f / q;
          b = ns > 0.f ? dot * nr / (ns * q * q) : 0.f;
        }
        a *= coeff * sc2; b *= coeff * sc2;
      } else if (metric == 1) {
        l += (double)ns2 - 2.0 * (double)dot + (double)nr2;
        a = -2.f * coeff; b = 2.f * coeff;
      } else {
        sd += dot; sr += nr2; ss += ns2;
      }
      if (nc == 0) { a = 0.f; b = 0.f; }      // class without synthetic nodes: no term
      abl[ci][0] = a; abl[ci][1] = b;
    }
  }
  red[0][tid] = metric == 2 ? sd : l;
  red[1][tid] = sr;
  red[2][tid] = ss;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if (tid < w) { red[0][tid] += red[0][tid + w]; red[1][tid] += red[1][tid + w]; red[2][tid] += red[2][tid + w]; }
    __syncthreads();
  }
  if (metric == 2) {      // (G == 1: every column is this workgroup's)
    const float dot = (float)red[0][0], nr = sqrtf((float)red[1][0]), ns = sqrtf((float)red[2][0]);
    const float q = nr * ns + 1e-6f;
    const float a = nc == 0 ? 0.f : -1.f / q * coeff, b = nc == 0 ? 0.f : (ns > 0.f ? dot * nr / (ns * q * q) : 0.f) * coeff;
    for (int k = tid; k < pd.total_cols; k += 1024) { abl[k][0] = a; abl[k][1] = b; }
    if (tid == 0) loss_c[c] = nc == 0 ? 0.f : coeff * (1.f - dot / q);
    __syncthreads();
  } else if (G == 1) {
    if (tid == 0) loss_c[c] = nc == 0 ? 0.f : coeff * (float)red[0][0];
  } else {
    // this workgroup's share of the class's loss; gs_syn_v_kernel (the next kernel of the stage that has a free thread) adds the G
    // shares in workgroup order and writes loss_c[c] — no float atomics, no in-kernel rendezvous (fence + counter: 10 000 cycles)
    if (tid == 0) part[(long)c * G + g] = red[0][0];
  }
  // ---- this workgroup's columns of (alpha, beta) and of D = alpha r + beta s
  float* o = ab + (long)c * pd.total_cols * 2;
  float* Dc = D + (long)c * P_total;
  for (int p = 0; p < pd.nparams; ++p) {
    const int rows = pd.rows[p], cols = pd.cols[p], co = pd.col_off[p];
    int k_lo, k_hi;
    range(p, k_lo, k_hi);
    if (k_hi <= k_lo) continue;
    for (int k = k_lo + tid; k < k_hi; k += 1024) { o[(co + k) * 2] = abl[co + k][0]; o[(co + k) * 2 + 1] = abl[co + k][1]; }
    const long off = pd.off[p];
    if (p == 0 && keep) {
      const int q_lo = k_lo >> 2, Q = (k_hi - k_lo) >> 2;
      int CB = 1;
      while (CB < Q && CB < 1024) CB <<= 1;
      const int tx = tid % CB, ty = tid / CB, nty = 1024 / CB;
      if (tx < Q) {
        const int k = 4 * (q_lo + tx);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = ty + j * nty;
          if (i < rows) {
            gs_f32x4u v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = abl[co + k + e][0] * keep_r[j][e] + abl[co + k + e][1] * keep_s[j][e];
            *reinterpret_cast<gs_f32x4u*>(Dc + off + (long)i * cols + k) = v;
          }
        }
      }
      continue;
    }
    if ((cols & 3) == 0) {
      // four 16-byte pieces per thread and pass, all eight loads requested before the first is used
      const int q_lo = k_lo >> 2, Q = (k_hi - k_lo) >> 2;
      const long n4 = (long)rows * Q;
      for (long b4 = tid; b4 < n4; b4 += 4096) {
        gs_f32x4u r[4], s_[4];
        long at[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long i4 = min(b4 + (long)u * 1024, n4 - 1);
          const long row = i4 / Q;
          at[u] = off + row * cols + 4 * (q_lo + (int)(i4 - row * Q));
          r[u] = *reinterpret_cast<const gs_f32x4u*>(R0 + at[u]);
          s_[u] = *reinterpret_cast<const gs_f32x4u*>(S0 + at[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (b4 + (long)u * 1024 < n4) {
            const int k = (int)((at[u] - off) % cols);
            gs_f32x4u v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = abl[co + k + e][0] * r[u][e] + abl[co + k + e][1] * s_[u][e];
            *reinterpret_cast<gs_f32x4u*>(Dc + at[u]) = v;
          }
        }
      }
    } else {
      const int W = k_hi - k_lo;
      for (long i = tid; i < (long)rows * W; i += 1024) {
        const long row = i / W;
        const int k = k_lo + (int)(i - row * W);
        const long at = off + row * cols + k;
        Dc[at] = abl[co + k][0] * R0[at] + abl[co + k][1] * S0[at];
      }
    }
  }
}

// with_bn (gcond_base.py:261-285 check_bn, models/sgc.py:27-33): the BatchNorm1d between the two SGC transforms always runs in
// eval mode with its initial running statistics (0, 1), i.e. y = x * s + beta with s = bn.weight / sqrt(1 + eps).  It is folded
// into the first transform: W1' = W1 diag(s), b1' = b1 * s + beta — every stage then runs on theta' = [W1', b1', W2, b2] as if
// there were no BatchNorm.  theta = [W1 (d x h), b1 (h), W2 (h x C), b2 (C), bn.weight (h), bn.bias (h)].
__global__ void gs_bn_fold_kernel(const float* __restrict__ theta, int d, int h, long oW1, long ob1, long og, long obeta, long P_base,
                                  float* __restrict__ theta_eff, float* __restrict__ s2) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P_base) return;
  const float rs = 1.f / sqrtf(1.f + 1e-5f);
  float v = theta[idx];
  if (idx >= oW1 && idx < oW1 + (long)d * h) {
    const int j = (int)((idx - oW1) % h);
    v *= theta[og + j] * rs;
  } else if (idx >= ob1 && idx < ob1 + h) {
    const int j = (int)(idx - ob1);
    const float sj = theta[og + j] * rs;
    v = v * sj + theta[obeta + j];
    s2[j] = sj * sj;
  }
  theta_eff[idx] = v;
}

// with_bn under dis_metric 'mse' / 'cos' (round 5; condensation/utils.py:41-60: both metrics run over EVERY parameter, the 1-D BatchNorm
// weight / bias included): the per-class gradients of the FOLDED model [W1', b1', W2, b2] (what the real side and the synthetic side
// compute) are mapped to the gradients of the true parameters [W1, b1, W2, b2, bn.weight, bn.bias] before the distance,
//     tW1[:, j] = gW1'[:, j] s_j ;  tb1_j = gb1'_j s_j ;  t bn.bias_j = gb1'_j ;  t bn.weight_j = (sum_i gW1'_ij W1_ij + gb1'_j b1_j) rs,
// with s_j = bn.weight_j rs, rs = 1 / sqrt(1 + eps) — a LINEAR map J(theta) of the folded gradient, the same for the real and the
// synthetic side.  The distance then yields D_t = d loss / d t_syn, and gs_bn_fold_class_D_kernel applies J^T:
//     D_W1'[i][j] = D_tW1[i][j] s_j + W1[i][j] D_tg[j] rs ;  D_b1'[j] = D_tb1[j] s_j + b1[j] D_tg[j] rs + D_tbeta[j]
// so that the second-order chain behind it runs on the folded model unchanged.  Grid (ceil(h / 32), classes), 256 threads.
__global__ __launch_bounds__(256) void gs_bn_unfold_class_grads_kernel(const float* __restrict__ g, long sg, const float* __restrict__ theta,
                                                                       float* __restrict__ t, long st, int d, int h, long oW1, long ob1,
                                                                       long og, long obeta, long P_base) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + tx;
  const float rs = 1.f / sqrtf(1.f + 1e-5f);
  const float* gc = g + (long)blockIdx.y * sg;
  float* tc = t + (long)blockIdx.y * st;
  __shared__ double sh[8][32];
  double acc = 0.0;
  const float sj = j < h ? theta[og + j] * rs : 0.f;
  if (j < h)
    for (int i = ty; i < d; i += 8) {
      const float gw = gc[oW1 + (long)i * h + j];
      tc[oW1 + (long)i * h + j] = gw * sj;
      acc += (double)gw * (double)theta[oW1 + (long)i * h + j];
    }
  sh[ty][tx] = acc;
  __syncthreads();
  if (j < h && ty == 0) {
    double tot = 0.0;
    for (int q = 0; q < 8; ++q) tot += sh[q][tx];
    const float gb = gc[ob1 + j];
    tc[ob1 + j] = gb * sj;
    tc[obeta + j] = gb;
    tc[og + j] = (float)((tot + (double)gb * (double)theta[ob1 + j]) * (double)rs);
  }
  // the other parameters (W2, b2: everything of the folded model outside W1 | b1) are the same in both models
  if (blockIdx.x == 0)
    for (long k = threadIdx.x; k < P_base; k += 256)
      if (!(k >= oW1 && k < oW1 + (long)d * h) && !(k >= ob1 && k < ob1 + h)) tc[k] = gc[k];
}

__global__ __launch_bounds__(256) void gs_bn_fold_class_D_kernel(float* __restrict__ D, long sD, const float* __restrict__ theta, int d, int h,
                                                                 long oW1, long ob1, long og, long obeta) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + tx;
  if (j >= h) return;
  const float rs = 1.f / sqrtf(1.f + 1e-5f);
  float* Dc = D + (long)blockIdx.y * sD;
  const float sj = theta[og + j] * rs, dg = Dc[og + j] * rs;
  for (int i = ty; i < d; i += 8) Dc[oW1 + (long)i * h + j] = fmaf(Dc[oW1 + (long)i * h + j], sj, theta[oW1 + (long)i * h + j] * dg);
  if (ty == 0) Dc[ob1 + j] = fmaf(Dc[ob1 + j], sj, fmaf(theta[ob1 + j], dg, Dc[obeta + j]));
}

// inner-step gradients of the folded model -> gradients of the true parameters (in place on g; g has room for bn.weight / bn.bias):
//   gW1[:, j] = gW1'[:, j] s_j ; gb1_j = gb1'_j s_j ; g bn.bias_j = gb1'_j ;
//   g bn.weight_j = sum_i dpre'_ij pre_ij / sqrt(1 + eps),  pre = X W1 + b1 with the TRUE W1, b1
// (dH1 = dpre' is already masked by the ReLU).  pre is re-formed from X and theta: Adam trains bn.weight, so the shortcut
// (H1 - beta) / bn.weight of the first build divides by a parameter that may pass through zero.  One block per 32 hidden columns.
__global__ __launch_bounds__(256) void gs_bn_unfold_grads_kernel(float* __restrict__ g, const float* __restrict__ theta,
                                                                 const float* __restrict__ dH1, const float* __restrict__ X, long ldx,
                                                                 int n, int d, int h, long oW1, long ob1, long og, long obeta) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + tx;
  const float rs = 1.f / sqrtf(1.f + 1e-5f);
  __shared__ double sh[8][32];
  double acc = 0.0;
  const float gam = j < h ? theta[og + j] : 1.f;
  if (j < h) {
    const float b1 = theta[ob1 + j];
    for (int i = ty; i < n; i += 8) {
      const float dh = dH1[(long)i * h + j];
      if (dh == 0.f) continue;
      float pre = b1;
      for (int k = 0; k < d; ++k) pre = fmaf(X[(long)i * ldx + k], theta[oW1 + (long)k * h + j], pre);
      acc += (double)dh * (double)pre;
    }
  }
  sh[ty][tx] = acc;
  __syncthreads();
  if (j >= h) return;
  const float sj = gam * rs;
  for (int i = ty; i < d; i += 8) g[oW1 + (long)i * h + j] *= sj;
  if (ty == 0) {
    double t = 0.0;
    for (int q = 0; q < 8; ++q) t += sh[q][tx];
    const float gb = g[ob1 + j];
    g[obeta + j] = gb;
    g[og + j] = (float)(t * (double)rs);
    g[ob1 + j] = gb * sj;
  }
}

// full-neighbour mode: Sr[c][b][:] = S_full[nid[c][b]][:F], rsr[c][b] = rs_full[nid[c][b]] for b < n_seed[c], zeros beyond
// (one wave per (class, seed slot))
__global__ __launch_bounds__(256) void gs_gather_seed_rows_kernel(const float* __restrict__ S_full, long ldS,
                                                                  const float* __restrict__ rs_full, const int* __restrict__ nid,
                                                                  int cap1, const int* __restrict__ n_seed, int bcap, int F,
                                                                  float* __restrict__ Sr, long ldSr, float* __restrict__ rsr, int C) {
  const int lane = threadIdx.x & 63;
  const long w = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (w >= (long)C * bcap) return;
  const int c = (int)(w / bcap), b = (int)(w - (long)c * bcap);
  float* out = Sr + w * ldSr;
  if (b >= n_seed[c]) {
    for (int k = lane; k < F; k += 64) out[k] = 0.f;
    if (lane == 0) rsr[w] = 0.f;
    return;
  }
  const long v = nid[(long)c * cap1 + b];
  const float* src = S_full + v * ldS;
  for (int k = lane; k < F; k += 64) out[k] = src[k];
  if (lane == 0) rsr[w] = rs_full[v];
}

// loss_total += sum_c loss_c (fixed order);  also keeps the last value in loss_last
__global__ void gs_loss_accum_kernel(const float* __restrict__ loss_c, int C, double* __restrict__ loss_sum,
                                     float* __restrict__ loss_last, long* __restrict__ steps) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += loss_c[c];
    *loss_last = s;
    *loss_sum += (double)s;
    *steps += 1;
  }
}

// ------------------------------------------------------------------------------------------------
// Real side, T = 2 backward pieces (autograd of models/sgc.py:37-47 on the sampled blocks).
// Y1[(c,l), :] = sum_{slot in in-edges of l} val1[slot] * Tm[(c, slot / f1), :]     (A1^T Tm via CSC)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gs_csc_spmm_t_kernel(const int* __restrict__ tptr, const int* __restrict__ tedge,
                                                            const float* __restrict__ val1, const int* __restrict__ n1,
                                                            int cap1, int bcap, int f1, const float* __restrict__ Tm, int F,
                                                            float* __restrict__ Y1, int nclass) {
  const int lane = threadIdx.x & 63;
  const long grow = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (grow >= (long)nclass * cap1) return;
  const int c = (int)(grow / cap1), l = (int)(grow - (long)c * cap1);
  float* o = Y1 + grow * F;
  if (l >= n1[c]) {
    for (int k = lane; k < F; k += 64) o[k] = 0.f;
    return;
  }
  const int* tp = tptr + (long)c * (cap1 + 1);
  const int* te = tedge + (long)c * bcap * f1;
  const int a = tp[l], b = tp[l + 1];
  for (int k = lane; k < F; k += 64) {
    float acc = 0.f;
    for (int i = a; i < b; ++i) {
      int slot = te[i];
      acc = fmaf(val1[(long)c * bcap * f1 + slot], Tm[((long)c * bcap + slot / f1) * F + k], acc);
    }
    o[k] = acc;
  }
}

// F = 256: one wave per frontier row, each lane owns 4 consecutive columns (one 16-byte load per in-edge, the edge list is
// walked once instead of once per 64-column slice).
__global__ __launch_bounds__(256) void gs_csc_spmm_t_v4_kernel(const int* __restrict__ tptr, const int* __restrict__ tedge,
                                                               const float* __restrict__ val1, const int* __restrict__ n1,
                                                               int cap1, int bcap, int f1, const float* __restrict__ Tm,
                                                               float* __restrict__ Y1, int nclass) {
  constexpr int F = 256;
  const int lane = threadIdx.x & 63;
  const long grow = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (grow >= (long)nclass * cap1) return;
  const int c = (int)(grow / cap1), l = (int)(grow - (long)c * cap1);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (l >= n1[c]) return;      // (padding rows of Y are never read: the edge product / the next block's CSC walk live edges only)
  {
    const int* tp = tptr + (long)c * (cap1 + 1);
    const int* te = tedge + (long)c * bcap * f1;
    const float* vv = val1 + (long)c * bcap * f1;
    const float* tm = Tm + (long)c * bcap * F + 4 * lane;
    const int a = tp[l], b = tp[l + 1];
    int i = a;
    for (; i + 1 < b; i += 2) {                 // two edges in flight
      const int s0 = te[i], s1 = te[i + 1];
      const float w0 = vv[s0], w1 = vv[s1];
      const float4 t0 = *reinterpret_cast<const float4*>(tm + (long)(s0 / f1) * F);
      const float4 t1 = *reinterpret_cast<const float4*>(tm + (long)(s1 / f1) * F);
      acc.x = fmaf(w0, t0.x, acc.x); acc.y = fmaf(w0, t0.y, acc.y); acc.z = fmaf(w0, t0.z, acc.z); acc.w = fmaf(w0, t0.w, acc.w);
      acc.x = fmaf(w1, t1.x, acc.x); acc.y = fmaf(w1, t1.y, acc.y); acc.z = fmaf(w1, t1.z, acc.z); acc.w = fmaf(w1, t1.w, acc.w);
    }
    if (i < b) {
      const int s0 = te[i];
      const float w0 = vv[s0];
      const float4 t0 = *reinterpret_cast<const float4*>(tm + (long)(s0 / f1) * F);
      acc.x = fmaf(w0, t0.x, acc.x); acc.y = fmaf(w0, t0.y, acc.y); acc.z = fmaf(w0, t0.z, acc.z); acc.w = fmaf(w0, t0.w, acc.w);
    }
  }
  *reinterpret_cast<float4*>(Y1 + grow * F + 4 * lane) = acc;
}

// Bmat[(c,l,j), :] = val0 * [H1full[col0, :] > 0] * Y1[(c,l), :]    (d loss / d pre-activation, per edge)
__global__ __launch_bounds__(256) void gs_make_bmat_kernel(const int* __restrict__ col0, const float* __restrict__ val0,
                                                           const int* __restrict__ n1, int cap1, int f2,
                                                           const float* __restrict__ H1full, long ldh,
                                                           const float* __restrict__ Y1, int F, float* __restrict__ Bmat,
                                                           int* __restrict__ kcount, int nclass, unsigned long long dkey0 = 0,
                                                           int dclass0 = 0, unsigned dthr = 0, float dq = 1.f) {
  const int lane = threadIdx.x & 63;
  const long ge = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;     // (c, l, j)
  if (ge >= (long)nclass * cap1 * f2) return;
  const long grow = ge / f2;
  const int c = (int)(grow / cap1), l = (int)(grow - (long)c * cap1);
  if (ge == (long)c * cap1 * f2 && lane == 0) kcount[c] = n1[c] * f2;
  if (l >= n1[c]) return;     // beyond kcount: never read
  const float w = val0[ge];
  const float* h = H1full + (long)col0[ge] * ldh;
  const float* y = Y1 + grow * F;
  float* o = Bmat + ge * F;
  if (dthr) {      // dropout (models/sgc.py:42): the source row's mask of this class multiplies the ReLU indicator
    const unsigned long long dkey = gs_mix64(dkey0 + (unsigned)(dclass0 + c));
    const long srow = col0[ge];
    for (int k = lane; k < F; k += 64) o[k] = (h[k] > 0.f) ? w * y[k] * gs_drop_mult1(dkey, srow, k, dthr, dq) : 0.f;
    return;
  }
  for (int k = lane; k < F; k += 64) o[k] = (h[k] > 0.f) ? w * y[k] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// PGE (models/parametrized_adj.py:40-71) with the layer-1 decomposition E1[r,c] = U[c] + V[r] + b0.
// ------------------------------------------------------------------------------------------------
// BN-1 batch statistics from U, V alone: mean = mean(U)+mean(V)+b0, var = var(U)+var(V) (biased), fp64.
// block = 32 columns x 32 row groups (1024 threads); two passes (mean, then centred squares) with an LDS reduction each.
// The matrices are tiny (n' x Ph) and this sits on the critical path of both PGE passes: the row loop is 8-way unrolled with
// clamped, unconditional loads so that a thread keeps 16 loads in flight instead of walking a dependent chain.
#define GS_BN1_RG 32
__global__ __launch_bounds__(1024) void gs_pge_bn1_stats_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                                const float* __restrict__ b0, int n, int Ph,
                                                                float* __restrict__ mu1, float* __restrict__ is1,
                                                                float* __restrict__ sumU, float* __restrict__ sumV,
                                                                const float* __restrict__ g1, const float* __restrict__ be1,
                                                                float* __restrict__ a1_tab, float* __restrict__ Us = nullptr,
                                                                float* __restrict__ Vs = nullptr, const float* __restrict__ W1 = nullptr,
                                                                __bf16* __restrict__ w1_split = nullptr) {
  // a1_tab (optional): [Ph][8] = {b0, mu1, is1, g1, be1, 0, 0, 0} per column — the constants of the generated layer-1
  // activations (gs_panel.hip, gs_panel_a1), written here so that no separate table launch is needed.
  // Us, Vs (optional, round 4): the workgroup also writes the folded layer-1 terms of its 32 columns (the arithmetic of
  // gs_pge_scale_uv_kernel, bitwise: Us = s U + t, Vs = s V, s = g1 is1, t = s (b0 - mu1) + be1) — one launch less per PGE pass.
  // w1_split (optional): workgroups beyond the ceil(Ph / 32) statistics workgroups write the exact three-way bf16 split image of the
  // layer-2 weight W1 [Ph][Ph] (independent work folded into this launch: the layer-2 product then starts without its own split launch)
  const int nstat = (Ph + 31) / 32;
  if ((int)blockIdx.x >= nstat) {
    const long idx = (long)(blockIdx.x - nstat) * 1024 + threadIdx.x;
    if (w1_split && idx < (long)Ph * Ph) gs_split3_store(W1[idx], (int)(idx / Ph), (int)(idx % Ph), Ph, w1_split);
    return;
  }
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int k = min(blockIdx.x * 32 + tx, Ph - 1);
  __shared__ double sh[2][GS_BN1_RG][32];
  __shared__ float fold[2][32];
  double su = 0, sv = 0;
  for (int i0 = ty; i0 < n; i0 += 8 * GS_BN1_RG) {
    float u[8], v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = min(i0 + j * GS_BN1_RG, n - 1);
      u[j] = U[(long)i * Ph + k];
      v[j] = V[(long)i * Ph + k];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (i0 + j * GS_BN1_RG < n) { su += u[j]; sv += v[j]; }
  }
  sh[0][ty][tx] = su;
  sh[1][ty][tx] = sv;
  __syncthreads();
  su = sv = 0;
  for (int q = 0; q < GS_BN1_RG; ++q) { su += sh[0][q][tx]; sv += sh[1][q][tx]; }
  __syncthreads();
  const double mu = su / n, mv = sv / n;
  double vu = 0, vv = 0;
  for (int i0 = ty; i0 < n; i0 += 8 * GS_BN1_RG) {
    float u[8], v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = min(i0 + j * GS_BN1_RG, n - 1);
      u[j] = U[(long)i * Ph + k];
      v[j] = V[(long)i * Ph + k];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (i0 + j * GS_BN1_RG < n) {
        const double a = u[j] - mu, b = v[j] - mv;
        vu += a * a;
        vv += b * b;
      }
  }
  sh[0][ty][tx] = vu;
  sh[1][ty][tx] = vv;
  __syncthreads();
  if (ty == 0 && blockIdx.x * 32 + tx < Ph) {
    vu = vv = 0;
    for (int q = 0; q < GS_BN1_RG; ++q) { vu += sh[0][q][tx]; vv += sh[1][q][tx]; }
    mu1[k] = (float)(mu + mv + (double)b0[k]);
    is1[k] = (float)(1.0 / sqrt(vu / n + vv / n + 1e-5));
    sumU[k] = (float)su;
    sumV[k] = (float)sv;
    if (a1_tab) {
      float* o = a1_tab + 8 * k;
      o[0] = b0[k]; o[1] = mu1[k]; o[2] = is1[k]; o[3] = g1[k]; o[4] = be1[k]; o[5] = o[6] = o[7] = 0.f;
    }
    const float m1 = (float)(mu + mv + (double)b0[k]), i1 = (float)(1.0 / sqrt(vu / n + vv / n + 1e-5));      // = mu1[k], is1[k]
    const float sc = g1[k] * i1;
    fold[0][tx] = sc;
    fold[1][tx] = fmaf(sc, b0[k] - m1, be1[k]);
  }
  if (Us) {      // (kernel argument: uniform)
    __syncthreads();
    if (blockIdx.x * 32 + tx < Ph) {
      const float sc = fold[0][tx], t = fold[1][tx];
      for (int i = ty; i < n; i += GS_BN1_RG) {
        Us[(long)i * Ph + k] = fmaf(sc, U[(long)i * Ph + k], t);
        Vs[(long)i * Ph + k] = sc * V[(long)i * Ph + k];
      }
    }
  }
}

// A1[(r,c), k] = relu(g1 * (U[c,k] + V[r,k] + b0 - mu1) * is1 + be1); Ph % 4 == 0: one thread per 4 consecutive k
__global__ void gs_pge_make_a1_v4_kernel(const float* __restrict__ U, const float* __restrict__ V, const float* __restrict__ b0,
                                         const float* __restrict__ mu1, const float* __restrict__ is1,
                                         const float* __restrict__ g1, const float* __restrict__ be1, int n, int Ph, int r_lo,
                                         int nr, float* __restrict__ A1) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int Q = Ph >> 2;
  if (idx >= (long)nr * n * Q) return;
  const int k = (int)(idx % Q) * 4;
  const long pair = idx / Q;
  const int rl = (int)(pair / n), c = (int)(pair - (long)rl * n), r = r_lo + rl;
  const float4 u = *reinterpret_cast<const float4*>(U + (long)c * Ph + k);
  const float4 v = *reinterpret_cast<const float4*>(V + (long)r * Ph + k);
  const float4 b = *reinterpret_cast<const float4*>(b0 + k), m = *reinterpret_cast<const float4*>(mu1 + k);
  const float4 is = *reinterpret_cast<const float4*>(is1 + k), g = *reinterpret_cast<const float4*>(g1 + k);
  const float4 be = *reinterpret_cast<const float4*>(be1 + k);
  float4 o;
  o.x = fmaxf(fmaf(g.x, (u.x + v.x + b.x - m.x) * is.x, be.x), 0.f);
  o.y = fmaxf(fmaf(g.y, (u.y + v.y + b.y - m.y) * is.y, be.y), 0.f);
  o.z = fmaxf(fmaf(g.z, (u.z + v.z + b.z - m.z) * is.z, be.z), 0.f);
  o.w = fmaxf(fmaf(g.w, (u.w + v.w + b.w - m.w) * is.w, be.w), 0.f);
  *reinterpret_cast<float4*>(A1 + pair * Ph + k) = o;
}

// Streaming form for Ph = 4 * Q, Q in {32, 64}: a thread keeps the BN-1 affine, b0 and V[r] of its 4 columns in registers and
// sweeps 8 pair columns c of one pair row r, so one output float4 costs one L2 read (U[c]) instead of seven.
template <int Q>
__global__ __launch_bounds__(256) void gs_pge_make_a1_v5_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                                const float* __restrict__ b0, const float* __restrict__ mu1,
                                                                const float* __restrict__ is1, const float* __restrict__ g1,
                                                                const float* __restrict__ be1, int n, int r_lo, int nr,
                                                                float* __restrict__ A1) {
  constexpr int Ph = 4 * Q, CS = 256 / Q, SWEEP = 8;      // CS pair columns per pass, SWEEP passes per block
  const int k = (threadIdx.x % Q) * 4, cs = threadIdx.x / Q;
  const int rl = blockIdx.y, r = r_lo + rl;
  const int c0 = blockIdx.x * (CS * SWEEP);
  const float4 v = *reinterpret_cast<const float4*>(V + (long)r * Ph + k);
  const float4 b = *reinterpret_cast<const float4*>(b0 + k), m = *reinterpret_cast<const float4*>(mu1 + k);
  const float4 is = *reinterpret_cast<const float4*>(is1 + k), g = *reinterpret_cast<const float4*>(g1 + k);
  const float4 be = *reinterpret_cast<const float4*>(be1 + k);
  float4 u[SWEEP];
#pragma unroll
  for (int i = 0; i < SWEEP; ++i) u[i] = *reinterpret_cast<const float4*>(U + (long)min(c0 + i * CS + cs, n - 1) * Ph + k);
#pragma unroll
  for (int i = 0; i < SWEEP; ++i) {
    const int c = c0 + i * CS + cs;
    float4 o;
    o.x = fmaxf(fmaf(g.x, (u[i].x + v.x + b.x - m.x) * is.x, be.x), 0.f);
    o.y = fmaxf(fmaf(g.y, (u[i].y + v.y + b.y - m.y) * is.y, be.y), 0.f);
    o.z = fmaxf(fmaf(g.z, (u[i].z + v.z + b.z - m.z) * is.z, be.z), 0.f);
    o.w = fmaxf(fmaf(g.w, (u[i].w + v.w + b.w - m.w) * is.w, be.w), 0.f);
    if (c < n) *reinterpret_cast<float4*>(A1 + ((long)rl * n + c) * Ph + k) = o;
  }
}

// column sums of up to two per-element functions over a tall matrix, fp64 partials:
// part[blk][0][k] = sum_rows f0, part[blk][1][k] = sum_rows f1.   mode 0: (x, x*x)  [BN-2 statistics]
__global__ __launch_bounds__(256) void gs_col_stats_kernel(const float* __restrict__ E, long rows, int Ph, int rows_per_block,
                                                           double* __restrict__ part) {
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  for (int k = threadIdx.x; k < Ph; k += 256) {
    double s = 0, s2 = 0;
    for (long r = r0; r < r1; ++r) {
      double x = E[r * Ph + k];
      s += x;
      s2 += x * x;
    }
    part[((long)blockIdx.x * 2 + 0) * Ph + k] = s;
    part[((long)blockIdx.x * 2 + 1) * Ph + k] = s2;
  }
}
// mean / inverse std of BN-2 from the (all-reduced) column sums:  sums = [sum x | sum x^2], fp64
// sum of `slices` partial sums laid out [slice][nq][Ph], in the order gs_sum_parts_kernel adds them (8 strided groups, then
// the groups in order) — so a consumer that folds the last reduction stage into itself gives the bits of the two-launch form
__device__ __forceinline__ double gs_sum_slices(const double* __restrict__ p, int slices, int nq, int Ph, int q, int k) {
  if (slices <= 1) return p[(long)q * Ph + k];
  double v[32];          // (reduce_parts leaves at most 32 slices) all loads in flight at once, then the fixed-order adds
#pragma unroll
  for (int b = 0; b < 32; ++b) v[b] = b < slices ? p[((long)b * nq + q) * Ph + k] : 0.0;
  double t = 0;
#pragma unroll
  for (int g = 0; g < 8; ++g) t += ((v[g] + v[g + 8]) + v[g + 16]) + v[g + 24];
  return t;
}

__global__ void gs_bn2_from_sums_kernel(const double* __restrict__ sums, int slices, long rows, int Ph, float* __restrict__ mu2,
                                        float* __restrict__ is2) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= Ph) return;
  double m = gs_sum_slices(sums, slices, 2, Ph, 0, k) / rows, v = gs_sum_slices(sums, slices, 2, Ph, 1, k) / rows - m * m;
  if (v < 0) v = 0;
  mu2[k] = (float)m;
  is2[k] = (float)(1.0 / sqrt(v + 1e-5));
}

// gs_sum_parts_f32_kernel (both stages) + gs_bn2_from_sums_kernel in ONE launch: the BatchNorm-2 batch statistics straight from the
// layer-2 product's per-tile column sums part[blk][2][Ph] (fp32), with the SAME grouping of the fp64 sums (S slices of
// ceil(nblk / S) blocks, 8 interleaved row groups per slice, slices combined as in gs_sum_slices) — bitwise the three-launch values.
// grid ceil(Ph / 32), block 1024 = 32 columns x 32 slices; S = 32 (nblk > 64) or 1.
__global__ __launch_bounds__(1024) void gs_bn2_stats_fold_kernel(const float* __restrict__ part, int nblk, int S, long rows, int Ph,
                                                                 float* __restrict__ mu2, float* __restrict__ is2) {
  const int tx = threadIdx.x & 31, z = threadIdx.x >> 5;
  const int k = min(blockIdx.x * 32 + tx, Ph - 1);
  const int per = (nblk + S - 1) / S, b0 = z * per, b1 = min(nblk, b0 + per);
  __shared__ double sh[2][32][32];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    double s = 0;
    if (z < S) {
      double s8[8];
#pragma unroll
      for (int ty = 0; ty < 8; ++ty) {
        double a = 0;
        for (int b = b0 + ty; b < b1; b += 8) a += (double)part[((long)b * 2 + q) * Ph + k];
        s8[ty] = a;
      }
#pragma unroll
      for (int g = 0; g < 8; ++g) s += s8[g];
    }
    sh[q][z][tx] = s;
  }
  __syncthreads();
  if (z != 0 || blockIdx.x * 32 + tx >= Ph) return;
  double tot[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (S <= 1) { tot[q] = sh[q][0][tx]; continue; }
    double t = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += ((sh[q][g][tx] + sh[q][g + 8][tx]) + sh[q][g + 16][tx]) + sh[q][g + 24][tx];
    tot[q] = t;
  }
  const double m = tot[0] / rows;
  double v = tot[1] / rows - m * m;
  if (v < 0) v = 0;
  mu2[k] = (float)m;
  is2[k] = (float)(1.0 / sqrt(v + 1e-5));
}

// e3[row] = sum_k relu(g2*(E2-mu2)*is2+be2) * w2[k] + b2      (one wave per pair)
__global__ __launch_bounds__(256) void gs_pge_layer3_kernel(const float* __restrict__ E2, const float* __restrict__ mu2,
                                                            const float* __restrict__ is2, const float* __restrict__ g2,
                                                            const float* __restrict__ be2, const float* __restrict__ w2,
                                                            const float* __restrict__ b2, long rows, int Ph,
                                                            float* __restrict__ e3) {
  const int lane = threadIdx.x & 63;
  const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= rows) return;
  float s = 0.f;
  for (int k = lane; k < Ph; k += 64) {
    float y = fmaf(g2[k], (E2[row * Ph + k] - mu2[k]) * is2[k], be2[k]);
    s = fmaf(fmaxf(y, 0.f), w2[k], s);
  }
  s = gs_wave_sum(s);
  if (lane == 0) e3[row] = s + b2[0];
}

// Same, streaming form for Ph = 4 * LPR (LPR = 64 or 32 lanes per row): each lane owns 4 consecutive columns, the BN-2 affine
// and w2 of those columns stay in registers, and a wave sweeps 8 passes of 64 / LPR rows with all its 16-byte loads in
// flight at once (the wave-per-row form re-read five per-column constants per element and ran at 2.4 TB/s).
template <int LPR>
__global__ __launch_bounds__(256) void gs_pge_layer3_v4_kernel(const float* __restrict__ E2, const float* __restrict__ mu2,
                                                               const float* __restrict__ is2, const float* __restrict__ g2,
                                                               const float* __restrict__ be2, const float* __restrict__ w2,
                                                               const float* __restrict__ b2, long rows, float* __restrict__ e3) {
  constexpr int RPW = 64 / LPR, PASSES = 8, Ph = 4 * LPR;
  const int lane = threadIdx.x & 63, sub = lane / LPR, cl = lane % LPR;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long base = wave * (RPW * PASSES);
  if (base >= rows) return;
  const float4 g = *reinterpret_cast<const float4*>(g2 + 4 * cl), m = *reinterpret_cast<const float4*>(mu2 + 4 * cl);
  const float4 is = *reinterpret_cast<const float4*>(is2 + 4 * cl), be = *reinterpret_cast<const float4*>(be2 + 4 * cl);
  const float4 w = *reinterpret_cast<const float4*>(w2 + 4 * cl);
  const float4 a = make_float4(g.x * is.x, g.y * is.y, g.z * is.z, g.w * is.w);
  // y = g*(E - mu)*is + be, evaluated as in the wave-per-row kernel (same rounding of the BN affine)
  float4 ev[PASSES];
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const long row = min(base + ps * RPW + sub, rows - 1);
    ev[ps] = *reinterpret_cast<const float4*>(E2 + row * Ph + 4 * cl);
  }
  const float bias = b2[0];
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const float4 e = ev[ps];
    float s = fmaxf(fmaf(g.x, (e.x - m.x) * is.x, be.x), 0.f) * w.x;
    s = fmaf(fmaxf(fmaf(g.y, (e.y - m.y) * is.y, be.y), 0.f), w.y, s);
    s = fmaf(fmaxf(fmaf(g.z, (e.z - m.z) * is.z, be.z), 0.f), w.z, s);
    s = fmaf(fmaxf(fmaf(g.w, (e.w - m.w) * is.w, be.w), 0.f), w.w, s);
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const long row = base + ps * RPW + sub;
    if (cl == 0 && row < rows) e3[row] = s + bias;
  }
  (void)a;
}

// adj[r,c] = sigmoid((e3[r,c] + e3[c,r]) / 2), zero diagonal     (parametrized_adj.py:66-70)
__global__ void gs_pge_adj_kernel(const float* __restrict__ e3, int n, float* __restrict__ adj) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * n) return;
  const int r = (int)(idx / n), c = (int)(idx - (long)r * n);
  float x = 0.5f * (e3[idx] + e3[(long)c * n + r]);
  adj[idx] = (r == c) ? 0.f : 1.f / (1.f + expf(-x));
}

// gs_pge_adj_kernel + the degree pass of dense_gcn_norm (gs_norm_deg_kernel, gs_core.hip) in one launch: block i writes row i of
// the structure and r[i] = (1 + sum_j adj[i][j])^-1/2 — the same per-thread strided partial sums, wave sums and 4-term total as the
// separate degree kernel, so r is bitwise the two-launch value (one dependent launch less on both PGE passes)
__global__ __launch_bounds__(256) void gs_pge_adj_deg_kernel(const float* __restrict__ e3, int n, float* __restrict__ adj,
                                                             float* __restrict__ r) {
  const int i = blockIdx.x;
  float s = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) {
    const float x = 0.5f * (e3[(long)i * n + j] + e3[(long)j * n + i]);
    const float a = (i == j) ? 0.f : 1.f / (1.f + expf(-x));
    adj[(long)i * n + j] = a;
    s += a;
  }
  __shared__ float red[4];
  s = gs_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float deg = red[0] + red[1] + red[2] + red[3] + 1.0f;
    const float ri = powf(deg, -0.5f);
    r[i] = isinf(ri) ? 0.f : ri;
  }
}

// A^ = diag(r) (adj + I) diag(r): the scaling pass of dense_gcn_norm (gs_norm_scale_kernel of gs_core.hip, same arithmetic)
__global__ void gs_pge_norm_scale_kernel(const float* __restrict__ adj, const float* __restrict__ r, int n, float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * n) return;
  const int i = (int)(idx / n), j = (int)(idx - (long)i * n);
  const float m = adj[idx] + (i == j ? 1.f : 0.f);
  out[idx] = r[i] * m * r[j];
}

// de3[r,c] = (g(r,c) + g(c,r)) / 2,  g = d_adj * a (1 - a) off the diagonal, 0 on it
__global__ void gs_pge_adj_bwd_kernel(const float* __restrict__ d_adj, const float* __restrict__ adj, int n,
                                      float* __restrict__ de3) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * n) return;
  const int r = (int)(idx / n), c = (int)(idx - (long)r * n);
  if (r == c) { de3[idx] = 0.f; return; }
  const long t = (long)c * n + r;
  float a = adj[idx], b = adj[t];
  de3[idx] = 0.5f * (d_adj[idx] * a * (1.f - a) + d_adj[t] * b * (1.f - b));
}

// layer-3 / BN-2 backward statistics (fp64 partials per block of rows):
// part[blk][0][k] = sum de3*A2 (-> dw2), [1] = sum dy2 (-> d beta2), [2] = sum dy2*xh2 (-> d gamma2),
// dy2 = de3 * w2[k] * [y2 > 0]
__global__ __launch_bounds__(256) void gs_pge_l3_bwd_stats_kernel(const float* __restrict__ E2, const float* __restrict__ de3,
                                                                  const float* __restrict__ mu2, const float* __restrict__ is2,
                                                                  const float* __restrict__ g2, const float* __restrict__ be2,
                                                                  const float* __restrict__ w2, long rows, int Ph,
                                                                  int rows_per_block, double* __restrict__ part) {
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  for (int k = threadIdx.x; k < Ph; k += 256) {
    double sw = 0, sb = 0, sg = 0;
    const float m = mu2[k], is = is2[k], g = g2[k], be = be2[k], w = w2[k];
    for (long r = r0; r < r1; ++r) {
      float xh = (E2[r * Ph + k] - m) * is;
      float y = fmaf(g, xh, be);
      float d = de3[r];
      if (y > 0.f) {
        sw += (double)(d * y);
        float dy = d * w;
        sb += dy;
        sg += (double)(dy * xh);
      }
    }
    part[((long)blockIdx.x * 3 + 0) * Ph + k] = sw;
    part[((long)blockIdx.x * 3 + 1) * Ph + k] = sb;
    part[((long)blockIdx.x * 3 + 2) * Ph + k] = sg;
  }
}
// Same partials, streaming form for Ph = 4 * QG (QG = 64 or 32 column groups): a thread owns 4 columns and every RG-th row of
// the block's rows (RG = 256 / QG), loads 8 rows of E2 as float4 before touching them, and selects instead of branching;
// the RG row groups meet in LDS.  Same fp64 accumulation; rows are visited in a different order than the scalar kernel.
template <int QG>
__global__ __launch_bounds__(256) void gs_pge_l3_bwd_stats_v4_kernel(const float* __restrict__ E2, const float* __restrict__ de3,
                                                                     const float* __restrict__ mu2, const float* __restrict__ is2,
                                                                     const float* __restrict__ g2, const float* __restrict__ be2,
                                                                     const float* __restrict__ w2, long rows, int rows_per_block,
                                                                     double* __restrict__ part) {
  constexpr int Ph = 4 * QG, RG = 256 / QG, UN = 8;
  const int cg = threadIdx.x % QG, rg = threadIdx.x / QG, k = 4 * cg;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  const float4 m = *reinterpret_cast<const float4*>(mu2 + k), is = *reinterpret_cast<const float4*>(is2 + k);
  const float4 g = *reinterpret_cast<const float4*>(g2 + k), be = *reinterpret_cast<const float4*>(be2 + k);
  const float4 w = *reinterpret_cast<const float4*>(w2 + k);
  const float mm[4] = {m.x, m.y, m.z, m.w}, ii[4] = {is.x, is.y, is.z, is.w}, gg[4] = {g.x, g.y, g.z, g.w};
  const float bb[4] = {be.x, be.y, be.z, be.w}, ww[4] = {w.x, w.y, w.z, w.w};
  double sw[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0}, sg[4] = {0, 0, 0, 0};
  for (long rb = r0 + rg; rb < r1; rb += (long)UN * RG) {
    float4 e[UN];
    float d[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long r = min(rb + (long)u * RG, r1 - 1);
      e[u] = *reinterpret_cast<const float4*>(E2 + r * Ph + k);
      d[u] = de3[r];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const bool in = rb + (long)u * RG < r1;
      const float ev[4] = {e[u].x, e[u].y, e[u].z, e[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (ev[j] - mm[j]) * ii[j];
        const float y = fmaf(gg[j], xh, bb[j]);
        const bool on = in && y > 0.f;
        const float dy = d[u] * ww[j];
        sw[j] += on ? (double)(d[u] * y) : 0.0;
        sb[j] += on ? (double)dy : 0.0;
        sg[j] += on ? (double)(dy * xh) : 0.0;
      }
    }
  }
  __shared__ double sh[3][RG][Ph];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sh[0][rg][k + j] = sw[j];
    sh[1][rg][k + j] = sb[j];
    sh[2][rg][k + j] = sg[j];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 3 * Ph; t += 256) {
    const int q = t / Ph, kk = t % Ph;
    double a = 0;
#pragma unroll
    for (int r = 0; r < RG; ++r) a += sh[q][r][kk];
    part[((long)blockIdx.x * 3 + q) * Ph + kk] = a;
  }
}

// sums[q][k] = sum over partial blocks (fp64).  grid = (ceil(Ph/32), nq); block = 32 columns x 8 groups
__global__ __launch_bounds__(256) void gs_sum_parts_kernel(const double* __restrict__ part, int nblk, int nq, int Ph,
    double* __restrict__ sums) {
  // grid = (ceil(Ph/32), nq, slices): slice z sums its share of the partial blocks into sums[z][q][k]
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int k = blockIdx.x * 32 + tx, q = blockIdx.y;
  const int per = (nblk + gridDim.z - 1) / gridDim.z, b0 = blockIdx.z * per, b1 = min(nblk, b0 + per);
  __shared__ double sh[8][32];
  double s = 0;
  if (k < Ph)
    for (int b = b0 + ty; b < b1; b += 8) s += part[((long)b * nq + q) * Ph + k];
  sh[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && k < Ph) {
    s = 0;
    for (int g = 0; g < 8; ++g) s += sh[g][tx];
    sums[((long)blockIdx.z * nq + q) * Ph + k] = s;
  }
}
// same from fp32 partials laid out [blk][nq][Ph] (the GEMM epilogue's colstat)
__global__ __launch_bounds__(256) void gs_sum_parts_f32_kernel(const float* __restrict__ part, int nblk, int nq, int Ph,
    double* __restrict__ sums) {
  // grid = (ceil(Ph/32), nq, slices): slice z sums its share of the partial blocks into sums[z][q][k]
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int k = blockIdx.x * 32 + tx, q = blockIdx.y;
  const int per = (nblk + gridDim.z - 1) / gridDim.z, b0 = blockIdx.z * per, b1 = min(nblk, b0 + per);
  __shared__ double sh[8][32];
  double s = 0;
  if (k < Ph)
    for (int b = b0 + ty; b < b1; b += 8) s += (double)part[((long)b * nq + q) * Ph + k];
  sh[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && k < Ph) {
    s = 0;
    for (int g = 0; g < 8; ++g) s += sh[g][tx];
    sums[((long)blockIdx.z * nq + q) * Ph + k] = s;
  }
}
// out_q[k] = (float) sums[q][k]
__global__ void gs_sums_to_float_kernel(const double* __restrict__ sums, int nq, int Ph, float* __restrict__ out0,
                                        float* __restrict__ out1, float* __restrict__ out2) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= Ph) return;
  if (out0) out0[k] = (float)sums[k];
  if (nq > 1 && out1) out1[k] = (float)sums[Ph + k];
  if (nq > 2 && out2) out2[k] = (float)sums[2 * Ph + k];
}

// dE2[row,k] = g2*is2 * (dy2 - dbe2/N - xh2*dg2/N),  dy2 = de3[row] * w2[k] * [y2 > 0]   (BatchNorm-2 + layer-3 backward)
__global__ void gs_pge_bn2_bwd_v4_kernel(const float* __restrict__ E2, const float* __restrict__ de3,
                                         const float* __restrict__ mu2, const float* __restrict__ is2,
                                         const float* __restrict__ g2, const float* __restrict__ be2,
                                         const float* __restrict__ w2, const float* __restrict__ dbe2,
                                         const float* __restrict__ dg2, long rows, long total_rows, int Ph,
                                         float* __restrict__ dE2) {
  // thread = (row group, 4 consecutive columns); the 7 per-column constants are loaded once (16-B loads) and reused
  // over ROWS_PER_THREAD rows
  constexpr int RPT = 8;
  const int Q = Ph >> 2;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = (int)(gid % Q) * 4;
  const long r0 = (gid / Q) * RPT;
  if (r0 >= rows) return;
  const float4 m = *reinterpret_cast<const float4*>(mu2 + k), is = *reinterpret_cast<const float4*>(is2 + k);
  const float4 g = *reinterpret_cast<const float4*>(g2 + k), be = *reinterpret_cast<const float4*>(be2 + k);
  const float4 w = *reinterpret_cast<const float4*>(w2 + k), db = *reinterpret_cast<const float4*>(dbe2 + k);
  const float4 dg = *reinterpret_cast<const float4*>(dg2 + k);
  const float invN = 1.f / (float)total_rows;
#pragma unroll
  for (int t = 0; t < RPT; ++t) {
    const long r = r0 + t;
    if (r < rows) {
      const float4 e = *reinterpret_cast<const float4*>(E2 + r * Ph + k);
      const float d = de3[r];
      float4 o;
      float xh, y;
      xh = (e.x - m.x) * is.x; y = fmaf(g.x, xh, be.x); o.x = g.x * is.x * ((y > 0.f ? d * w.x : 0.f) - db.x * invN - xh * dg.x * invN);
      xh = (e.y - m.y) * is.y; y = fmaf(g.y, xh, be.y); o.y = g.y * is.y * ((y > 0.f ? d * w.y : 0.f) - db.y * invN - xh * dg.y * invN);
      xh = (e.z - m.z) * is.z; y = fmaf(g.z, xh, be.z); o.z = g.z * is.z * ((y > 0.f ? d * w.z : 0.f) - db.z * invN - xh * dg.z * invN);
      xh = (e.w - m.w) * is.w; y = fmaf(g.w, xh, be.w); o.w = g.w * is.w * ((y > 0.f ? d * w.w : 0.f) - db.w * invN - xh * dg.w * invN);
      *reinterpret_cast<float4*>(dE2 + r * Ph + k) = o;
    }
  }
}

// marginal sums of dY1 [(r,c), k]:  colsum_r[r,k] = sum_c,  rowsum_c[c,k] = sum_r
// (dY1 holds the nr local pair rows r_lo .. r_lo+nr-1; colsum_r is written for those rows, rowsum_c is the partial sum)
// block = QL float4 lanes (QL = Ph/4 <= 64) x G row groups; each group strides over the summed index; LDS reduce.
__global__ __launch_bounds__(256) void gs_pge_marginal_r_kernel(const float* __restrict__ dY1, int n, int Ph, int r_lo,
                                                                float* __restrict__ colsum_r) {
  const int rl = blockIdx.x;
  const int Q = Ph >> 2, G = 256 / Q;              // Ph in {16..256}, multiple of 4 (checked by the caller)
  const int q = threadIdx.x % Q, g = threadIdx.x / Q;
  __shared__ float4 sh[256];
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g < G) {
    // 4 rows in flight per thread (clamped, unconditional loads; the tail is masked in the adds): the loop is a chain of
    // dependent 16-byte loads otherwise and the kernel has only n' workgroups
    for (int c0 = g; c0 < n; c0 += 4 * G) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        v[u] = *reinterpret_cast<const float4*>(dY1 + ((long)rl * n + min(c0 + u * G, n - 1)) * Ph + q * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c0 + u * G < n) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (g == 0) {
    for (int t = 1; t < G; ++t) {
      const float4 v = sh[t * Q + q];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(colsum_r + (long)(r_lo + rl) * Ph + q * 4) = s;
  }
}
__global__ __launch_bounds__(256) void gs_pge_marginal_c_kernel(const float* __restrict__ dY1, int n, int Ph, int nr,
                                                                float* __restrict__ rowsum_c) {
  const int c = blockIdx.x;
  const int Q = Ph >> 2, G = 256 / Q;
  const int q = threadIdx.x % Q, g = threadIdx.x / Q;
  __shared__ float4 sh[256];
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g < G) {
    for (int r0 = g; r0 < nr; r0 += 4 * G) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        v[u] = *reinterpret_cast<const float4*>(dY1 + ((long)min(r0 + u * G, nr - 1) * n + c) * Ph + q * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r0 + u * G < nr) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (g == 0) {
    for (int t = 1; t < G; ++t) {
      const float4 v = sh[t * Q + q];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(rowsum_c + (long)c * Ph + q * 4) = s;
  }
}
// Both marginals in ONE pass over dY1 (the two kernels above read its 211 MB twice).  Block (cb, rb) owns pair rows
// r in [16 rb, 16 rb + 16) x pair columns c in [64 cb, 64 cb + 64): a thread holds one float4 of k, walks the block's c with
// stride G and keeps 16 running row sums; per c it writes the sum over its 16 r (partU[rb][c][k]), at the end the block
// writes its 16 row sums over its 64 c (partV[cb][r][k]).  gs_reduce_slabs_f32 adds the partials in a fixed order.
#define GS_MARG_RR 16
#define GS_MARG_CC 64
__global__ __launch_bounds__(256) void gs_pge_marginals_fused_kernel(const float* __restrict__ dY1, int n, int Ph, int nr,
                                                                     float* __restrict__ partU /*[rb][n][Ph]*/,
                                                                     float* __restrict__ partV /*[cb][nr][Ph]*/) {
  const int cb = blockIdx.x, rb = blockIdx.y;
  const int Q = Ph >> 2, G = 256 / Q;
  const int q = threadIdx.x % Q, g = threadIdx.x / Q;
  const int r0 = rb * GS_MARG_RR, c0 = cb * GS_MARG_CC;
  const int c1 = min(n, c0 + GS_MARG_CC);
  __shared__ float4 sh[256];
  float4 accv[GS_MARG_RR];
#pragma unroll
  for (int u = 0; u < GS_MARG_RR; ++u) accv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g < G) {
    for (int c = c0 + g; c < c1; c += G) {
      float4 v[GS_MARG_RR];
#pragma unroll
      for (int u = 0; u < GS_MARG_RR; ++u)
        v[u] = *reinterpret_cast<const float4*>(dY1 + ((long)min(r0 + u, nr - 1) * n + c) * Ph + q * 4);
      float4 su = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < GS_MARG_RR; ++u)
        if (r0 + u < nr) {
          su.x += v[u].x; su.y += v[u].y; su.z += v[u].z; su.w += v[u].w;
          accv[u].x += v[u].x; accv[u].y += v[u].y; accv[u].z += v[u].z; accv[u].w += v[u].w;
        }
      *reinterpret_cast<float4*>(partU + ((long)rb * n + c) * Ph + q * 4) = su;
    }
  }
  // the G groups walked different c: add their row sums in group order
  for (int u = 0; u < GS_MARG_RR; ++u) {
    __syncthreads();
    sh[threadIdx.x] = accv[u];
    __syncthreads();
    if (g == 0 && r0 + u < nr) {
      float4 t = sh[q];
      for (int w = 1; w < G; ++w) {
        const float4 x = sh[w * Q + q];
        t.x += x.x; t.y += x.y; t.z += x.z; t.w += x.w;
      }
      *reinterpret_cast<float4*>(partV + ((long)cb * nr + r0 + u) * Ph + q * 4) = t;
    }
  }
}

// dU | dV of one element from the marginals and the column gradients (shared by the stand-alone kernel and the fold in the columns
// kernel: one expression, one rounding)
__device__ __forceinline__ void gs_bn1_bwd_uv_elem(float u, float v, float rc, float cr, float n, float N, float kk, float shf, float sU,
                                                   float sV, float i1, float dbe, float dg, float& du, float& dv) {
#pragma clang fp contract(off)      // (no mul + add fusion: the two call sites must round alike whatever surrounds them)
  const float xr = (n * u + sV + shf) * i1;   // sum over r of xhat[r, c]
  const float xc = (n * v + sU + shf) * i1;   // sum over c of xhat[r, c]
  du = kk * (rc - n * dbe / N - xr * dg / N);
  dv = kk * (cr - n * dbe / N - xc * dg / N);
}

// BN-1 backward from the marginals (see oracle/analytic.py:pge_bwd); block = 32 columns x 8 row groups
__global__ __launch_bounds__(1024) void gs_pge_bn1_bwd_cols_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                                   const float* __restrict__ rowsum_c,
                                                                   const float* __restrict__ colsum_r,
                                                                   const float* __restrict__ b0, const float* __restrict__ mu1,
                                                                   const float* __restrict__ is1, int n, int Ph,
                                                                   float* __restrict__ dg1, float* __restrict__ dbe1,
                                                                   const float* __restrict__ g1 = nullptr,
                                                                   const float* __restrict__ sumU = nullptr,
                                                                   const float* __restrict__ sumV = nullptr, float* __restrict__ dU = nullptr,
                                                                   float* __restrict__ dV = nullptr) {
  // dU (optional, round 4): the workgroup also writes dU | dV of its 32 columns (the arithmetic of gs_pge_bn1_bwd_uv_kernel on the
  // float values of dg1 / dbe1 it has just formed: bitwise that kernel's output) — one launch less in the PGE backward.
  // 32 columns x 32 row groups, 4 rows x 4 arrays of loads in flight per thread (tiny matrices, critical path)
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int k = min(blockIdx.x * 32 + tx, Ph - 1);
  __shared__ double sh[3][GS_BN1_RG][32];
  __shared__ float bc[2][32];
  double sb = 0, su = 0, sv = 0;
  for (int i0 = ty; i0 < n; i0 += 4 * GS_BN1_RG) {
    float rc[4], u[4], v[4], cr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long o = (long)min(i0 + j * GS_BN1_RG, n - 1) * Ph + k;
      rc[j] = rowsum_c[o]; u[j] = U[o]; v[j] = V[o]; cr[j] = colsum_r[o];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i0 + j * GS_BN1_RG < n) {
        sb += rc[j];
        su += (double)u[j] * rc[j];
        sv += (double)v[j] * cr[j];
      }
  }
  sh[0][ty][tx] = sb;
  sh[1][ty][tx] = su;
  sh[2][ty][tx] = sv;
  __syncthreads();
  if (ty == 0 && blockIdx.x * 32 + tx < Ph) {
    sb = su = sv = 0;
    for (int q = 0; q < GS_BN1_RG; ++q) { sb += sh[0][q][tx]; su += sh[1][q][tx]; sv += sh[2][q][tx]; }
    const float dbe = (float)sb, dg = (float)((su + sv + ((double)b0[k] - (double)mu1[k]) * sb) * (double)is1[k]);
    dbe1[k] = dbe;
    dg1[k] = dg;
    if (dU) { bc[0][tx] = dbe; bc[1][tx] = dg; }
  }
  if (!dU) return;      // (kernel argument: uniform)
  __syncthreads();
  if (blockIdx.x * 32 + tx >= Ph) return;
  const float dbe = bc[0][tx], dg = bc[1][tx];
  const float N = (float)n * (float)n;
  const float kk = g1[k] * is1[k];
  const float shf = (float)n * (b0[k] - mu1[k]);
  const float sU = sumU[k], sV = sumV[k], i1 = is1[k];
  for (int i = ty; i < n; i += GS_BN1_RG) {
    const long o = (long)i * Ph + k;
    float du, dv;
    gs_bn1_bwd_uv_elem(U[o], V[o], rowsum_c[o], colsum_r[o], (float)n, N, kk, shf, sU, sV, i1, dbe, dg, du, dv);
    dU[o] = du;
    dV[o] = dv;
  }
}
// out[c][r] = in[r][c] for a small matrix (weights): lets a product that wants W^T as its n-contiguous operand run with both
// operands k-contiguous
__global__ void gs_transpose_small_kernel(const float* __restrict__ in, int rows, int cols, float* __restrict__ out) {
  __shared__ float t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    t[j][threadIdx.x] = (r < rows && c < cols) ? in[(long)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (c < cols && r < rows) out[(long)c * rows + r] = t[threadIdx.x][j];
  }
}
// PGE backward, start of part 2 — ONE launch for what were four: the last stage of the layer-3 statistics reduction
// (sums -> dw2, dbe2, dg2 as floats), the constants table of the generated dE2 operand (optional), and W1^T (both operands
// of the marginals product k-contiguous).  grid (ceil(Ph/32), ceil(Ph/32)), block (32, 8): every block transposes one 32x32
// tile of W1; the blocks of row 0 also finish 32 columns of the statistics.
__global__ void gs_pge_bwd_prep_kernel(const double* __restrict__ sums, int slices, int Ph, float* __restrict__ dw2,
                                       float* __restrict__ dbe2, float* __restrict__ dg2, const float* __restrict__ mu2,
                                       const float* __restrict__ is2, const float* __restrict__ g2, const float* __restrict__ be2,
                                       const float* __restrict__ w2, float invN, float* __restrict__ gen_table,
                                       const float* __restrict__ W1, float* __restrict__ W1T,
                                       __bf16* __restrict__ w1t_split = nullptr) {
  // w1t_split (optional, round 4): also the exact three-way bf16 split image of W1^T (B operand of the marginals product in mode 3:
  // that launch then needs no split launch of its own)
  __shared__ float t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    t[j][threadIdx.x] = (r < Ph && c < Ph) ? W1[(long)r * Ph + c] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (c < Ph && r < Ph) {
      W1T[(long)c * Ph + r] = t[threadIdx.x][j];
      if (w1t_split) gs_split3_store(t[threadIdx.x][j], c, r, Ph, w1t_split);
    }
  }
  if (blockIdx.y == 0 && threadIdx.y == 0) {
    const int k = c0 + threadIdx.x;
    if (k < Ph) {
      const float a = (float)gs_sum_slices(sums, slices, 3, Ph, 0, k), b = (float)gs_sum_slices(sums, slices, 3, Ph, 1, k),
                  c = (float)gs_sum_slices(sums, slices, 3, Ph, 2, k);
      dw2[k] = a; dbe2[k] = b; dg2[k] = c;
      if (gen_table) gs_panel_gen_row(mu2[k], is2[k], g2[k], be2[k], w2[k], b, c, invN, gen_table + 8 * k);
    }
  }
}
// gs_sum_parts_kernel (both stages) + gs_pge_bwd_prep_kernel in ONE launch (unsharded engines): the first ceil(Ph / 32) workgroups
// reduce the layer-3 statistics partials part[blk][3][Ph] (fp64) with the SAME grouping as the two reduction stages (bitwise their
// values) and finish dw2 | dbe2 | dg2 and the constants table; the others transpose one 32 x 32 tile of W1 each (+ the split image).
__global__ __launch_bounds__(1024) void gs_pge_bwd_prep_fold_kernel(const double* __restrict__ part, int nblk, int S, int Ph,
                                                                    float* __restrict__ dw2, float* __restrict__ dbe2,
                                                                    float* __restrict__ dg2, const float* __restrict__ mu2,
                                                                    const float* __restrict__ is2, const float* __restrict__ g2,
                                                                    const float* __restrict__ be2, const float* __restrict__ w2, float invN,
                                                                    float* __restrict__ gen_table, const float* __restrict__ W1,
                                                                    float* __restrict__ W1T, __bf16* __restrict__ w1t_split) {
  const int tx = threadIdx.x & 31, z = threadIdx.x >> 5;
  const int nstat = (Ph + 31) / 32;
  __shared__ double sh[3][32][32];
  if ((int)blockIdx.x >= nstat) {
    float (*t)[33] = reinterpret_cast<float (*)[33]>(&sh[0][0][0]);
    const int nt = (Ph + 31) / 32, idx = blockIdx.x - nstat;
    const int c0 = (idx % nt) * 32, r0 = (idx / nt) * 32;
    {
      const int r = r0 + z, c = c0 + tx;
      t[z][tx] = (r < Ph && c < Ph) ? W1[(long)r * Ph + c] : 0.f;
    }
    __syncthreads();
    const int c = c0 + z, r = r0 + tx;
    if (c < Ph && r < Ph) {
      W1T[(long)c * Ph + r] = t[tx][z];
      if (w1t_split) gs_split3_store(t[tx][z], c, r, Ph, w1t_split);
    }
    return;
  }
  const int k = min(blockIdx.x * 32 + tx, Ph - 1);
  const int per = (nblk + S - 1) / S, b0 = z * per, b1 = min(nblk, b0 + per);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    double s = 0;
    if (z < S) {
      double s8[8];
#pragma unroll
      for (int ty = 0; ty < 8; ++ty) {
        double a = 0;
        for (int b = b0 + ty; b < b1; b += 8) a += part[((long)b * 3 + q) * Ph + k];
        s8[ty] = a;
      }
#pragma unroll
      for (int g = 0; g < 8; ++g) s += s8[g];
    }
    sh[q][z][tx] = s;
  }
  __syncthreads();
  if (z != 0 || blockIdx.x * 32 + tx >= Ph) return;
  float tot[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (S <= 1) { tot[q] = (float)sh[q][0][tx]; continue; }
    double t = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += ((sh[q][g][tx] + sh[q][g + 8][tx]) + sh[q][g + 16][tx]) + sh[q][g + 24][tx];
    tot[q] = (float)t;
  }
  dw2[k] = tot[0]; dbe2[k] = tot[1]; dg2[k] = tot[2];
  if (gen_table) gs_panel_gen_row(mu2[k], is2[k], g2[k], be2[k], w2[k], tot[1], tot[2], invN, gen_table + 8 * k);
}
__global__ void gs_pge_bn1_bwd_uv_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                         const float* __restrict__ rowsum_c, const float* __restrict__ colsum_r,
                                         const float* __restrict__ b0, const float* __restrict__ mu1,
                                         const float* __restrict__ is1, const float* __restrict__ g1,
                                         const float* __restrict__ sumU, const float* __restrict__ sumV,
                                         const float* __restrict__ dg1, const float* __restrict__ dbe1, int n, int Ph,
                                         float* __restrict__ dU, float* __restrict__ dV) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * Ph) return;
  const int k = (int)(idx % Ph);
  const float N = (float)n * (float)n;
  const float kk = g1[k] * is1[k];
  const float sh = (float)n * (b0[k] - mu1[k]);
  float du, dv;
  gs_bn1_bwd_uv_elem(U[idx], V[idx], rowsum_c[idx], colsum_r[idx], (float)n, N, kk, sh, sumU[k], sumV[k], is1[k], dbe1[k], dg1[k], du, dv);
  dU[idx] = du;
  dV[idx] = dv;
}
// column sums of a small [rows x cols] matrix, fp64 accumulation.  One workgroup = 64 columns x 4 row groups: the first form
// walked all rows in ONE thread per column (106 us for 454 x 256 on the critical path of the PGE epochs).
__global__ __launch_bounds__(256) void gs_colsum_small_kernel(const float* __restrict__ M, int rows, int cols, float* __restrict__ out) {
  const int k = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  __shared__ double sh[4][64];
  double s = 0;
  if (k < cols) {
    int i = g;
    for (; i + 12 < rows; i += 16) {            // 4 independent loads in flight
      const float a = M[(long)i * cols + k], b = M[(long)(i + 4) * cols + k], c = M[(long)(i + 8) * cols + k], d = M[(long)(i + 12) * cols + k];
      s += ((double)a + (double)b) + ((double)c + (double)d);
    }
    for (; i < rows; i += 4) s += M[(long)i * cols + k];
  }
  sh[g][threadIdx.x & 63] = s;
  __syncthreads();
  if (g == 0 && k < cols) out[k] = (float)(((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x]);
}
__global__ void gs_fill_kernel(float* __restrict__ p, long n, float v) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void gs_fill_int_kernel(int* __restrict__ p, long n, int v) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// out[0] = sum_i x[i]   (one block of 1024 threads, fp64 accumulation)
__global__ __launch_bounds__(1024) void gs_sum_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
  double s = 0;
  for (long i = threadIdx.x; i < n; i += 1024) s += x[i];
  s = gs_wave_sum_d(s);
  __shared__ double sh[16];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int w = 0; w < 16; ++w) t += sh[w];
    out[0] = (float)t;
  }
}

// batched column sums of [rows_b x cols] matrices: partial[b][chunk][k] = sum over the chunk's rows; the partials are
// then summed in fixed order by gs_reduce_slabs_batched_f32.  grid = (nchunk, nbatch), rows of batch b limited by
// rows_b[b] (device) when given.
__global__ __launch_bounds__(256) void gs_colsum_chunks_kernel(const float* __restrict__ M, long batch_stride, int rows,
                                                               const int* __restrict__ rows_b, int cols, int chunk,
                                                               float* __restrict__ partial) {
  const int b = blockIdx.y;
  const int nr = rows_b ? min(rows, rows_b[b]) : rows;
  const int r0 = blockIdx.x * chunk, r1 = min(nr, r0 + chunk);
  const float* m = M + (long)b * batch_stride;
  float* o = partial + ((long)b * gridDim.x + blockIdx.x) * cols;
  for (int k = threadIdx.x; k < cols; k += 256) {
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += m[(long)r * cols + k];
    o[k] = s;
  }
}

// kcount[c] = n1[c] * f2  (rows of the per-class edge list that are live)
__global__ void gs_kcount_kernel(const int* __restrict__ n1, int f2, int nclass, int* __restrict__ kcount) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nclass) kcount[c] = n1[c] * f2;
}

// ------------------------------------------------------------------------------------------------
// Chunked PGE (parametrized_adj.py:41-55: reddit with r >= 0.01 passes the n'^2 pairs through the MLP in 5 contiguous
// chunks of the flat pair index, EACH with its own BatchNorm batch statistics).  The layer-1 decomposition still holds per
// pair, but the closed forms over the full grid do not: statistics are reduced numerically over each chunk's pairs.
// Pair k = r * n + c;  E1[k, j] = U[c, j] + V[r, j] + b0[j].
// ------------------------------------------------------------------------------------------------
// part[blk][0][j] = sum E1, part[blk][1][j] = sum E1^2 over the block's pairs (fp64)
__global__ __launch_bounds__(256) void gs_pge_chunk_e1_stats_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                                    const float* __restrict__ b0, int n, int Ph, long k0,
                                                                    long count, int rows_per_block, double* __restrict__ part) {
  const long p0 = (long)blockIdx.x * rows_per_block, p1 = min(count, p0 + rows_per_block);
  for (int j = threadIdx.x; j < Ph; j += 256) {
    double s = 0, s2 = 0;
    const float b = b0[j];
    for (long q = p0; q < p1; ++q) {
      const long k = k0 + q;
      const int r = (int)(k / n), c = (int)(k - (long)r * n);
      const double x = (double)(U[(long)c * Ph + j] + V[(long)r * Ph + j] + b);
      s += x;
      s2 += x * x;
    }
    part[((long)blockIdx.x * 2 + 0) * Ph + j] = s;
    part[((long)blockIdx.x * 2 + 1) * Ph + j] = s2;
  }
}
// A1[k, j] for the chunk's pairs (A1 points at the chunk's first row); one thread per 4 consecutive j
__global__ void gs_pge_make_a1_flat_kernel(const float* __restrict__ U, const float* __restrict__ V, const float* __restrict__ b0,
                                           const float* __restrict__ mu1, const float* __restrict__ is1,
                                           const float* __restrict__ g1, const float* __restrict__ be1, int n, int Ph, long k0,
                                           long count, float* __restrict__ A1) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int Q = Ph >> 2;
  if (idx >= count * Q) return;
  const int j = (int)(idx % Q) * 4;
  const long q = idx / Q, k = k0 + q;
  const int r = (int)(k / n), c = (int)(k - (long)r * n);
  const float4 u = *reinterpret_cast<const float4*>(U + (long)c * Ph + j);
  const float4 v = *reinterpret_cast<const float4*>(V + (long)r * Ph + j);
  const float4 b = *reinterpret_cast<const float4*>(b0 + j), m = *reinterpret_cast<const float4*>(mu1 + j);
  const float4 is = *reinterpret_cast<const float4*>(is1 + j), g = *reinterpret_cast<const float4*>(g1 + j);
  const float4 be = *reinterpret_cast<const float4*>(be1 + j);
  float4 o;
  o.x = fmaxf(fmaf(g.x, (u.x + v.x + b.x - m.x) * is.x, be.x), 0.f);
  o.y = fmaxf(fmaf(g.y, (u.y + v.y + b.y - m.y) * is.y, be.y), 0.f);
  o.z = fmaxf(fmaf(g.z, (u.z + v.z + b.z - m.z) * is.z, be.z), 0.f);
  o.w = fmaxf(fmaf(g.w, (u.w + v.w + b.w - m.w) * is.w, be.w), 0.f);
  *reinterpret_cast<float4*>(A1 + q * Ph + j) = o;
}
// BN-1 backward statistics of a chunk: part[blk][0][j] = sum dY1 (-> d beta1), [1] = sum dY1 * xhat1 (-> d gamma1)
__global__ __launch_bounds__(256) void gs_pge_chunk_dy1_stats_kernel(const float* __restrict__ dY1, const float* __restrict__ U,
                                                                     const float* __restrict__ V, const float* __restrict__ b0,
                                                                     const float* __restrict__ mu1, const float* __restrict__ is1,
                                                                     int n, int Ph, long k0, long count, int rows_per_block,
                                                                     double* __restrict__ part) {
  const long p0 = (long)blockIdx.x * rows_per_block, p1 = min(count, p0 + rows_per_block);
  for (int j = threadIdx.x; j < Ph; j += 256) {
    double sb = 0, sg = 0;
    const float b = b0[j], m = mu1[j], is = is1[j];
    for (long q = p0; q < p1; ++q) {
      const long k = k0 + q;
      const int r = (int)(k / n), c = (int)(k - (long)r * n);
      const float xh = (U[(long)c * Ph + j] + V[(long)r * Ph + j] + b - m) * is;
      const float dy = dY1[q * Ph + j];
      sb += (double)dy;
      sg += (double)(dy * xh);
    }
    part[((long)blockIdx.x * 2 + 0) * Ph + j] = sb;
    part[((long)blockIdx.x * 2 + 1) * Ph + j] = sg;
  }
}
// dE1 = g1 * is1 * (dY1 - dbe1 / m - xhat1 * dg1 / m) in place over the chunk's pairs (m = pairs in the chunk)
__global__ void gs_pge_chunk_bn1_bwd_kernel(float* __restrict__ dY1, const float* __restrict__ U, const float* __restrict__ V,
                                            const float* __restrict__ b0, const float* __restrict__ mu1,
                                            const float* __restrict__ is1, const float* __restrict__ g1,
                                            const float* __restrict__ dbe1, const float* __restrict__ dg1, int n, int Ph, long k0,
                                            long count) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= count * Ph) return;
  const int j = (int)(idx % Ph);
  const long q = idx / Ph, k = k0 + q;
  const int r = (int)(k / n), c = (int)(k - (long)r * n);
  const float xh = (U[(long)c * Ph + j] + V[(long)r * Ph + j] + b0[j] - mu1[j]) * is1[j];
  const float m = (float)count;
  dY1[idx] = g1[j] * is1[j] * (dY1[idx] - dbe1[j] / m - xh * dg1[j] / m);
}

