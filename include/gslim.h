/* libgslim_hip.so — C ABI of the MI355X (gfx950) GCond gradient-matching engine.
 *
 * The reference (GraphSlim) is pure Python and has no FFI; its boundary for this path is the Python
 * method GCond(setting, data, args).reduce(data) (graphslim/condensation/gcond.py:14-18,81).  This
 * header is the C ABI that graphslim_amd's host-side mirror of that method binds through ctypes
 * (INTEGRATION.md shows the stub).  Every entry point below names the reference code it replaces.
 *
 * Conventions: plain C, no exceptions.  Every function returns GS_OK (0) or a negative error code;
 * gs_last_error() returns a thread-local message.  All pointers are DEVICE pointers unless a
 * parameter is documented as host.  The library never allocates, frees or retains caller memory
 * beyond the call, except gs_gcond_* which borrows the buffers named in gs_gcond_io until
 * gs_gcond_destroy.  `stream` is a hipStream_t (0 = default stream); calls are asynchronous on it.
 * Indices are int32, values fp32 (fp32 accumulation; BatchNorm statistics accumulate in fp64).
 *
 * Threading.  Functions without a handle (gs_gemm_f32, gs_spmm_csr_f32, gs_adam_step_f32, ...) keep no state between calls and are
 * re-entrant: any thread may call them, concurrently, on any stream.  A gs_gcond / gs_comm handle is NOT internally locked: one
 * thread at a time per handle (different handles may be driven from different threads concurrently — the multi-rank tests do);
 * every state an engine carries between calls lives behind its handle.  What is process-global is ONLY the set of measurement
 * knobs and debug hooks declared in gslim_internal.h: they are not part of this ABI, default to the product configuration, are not
 * thread-safe, and a product build never calls them (the engine-level ones can be pinned per engine with gs_gcond_set_knob).
 */
#ifndef GSLIM_H
#define GSLIM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_OK 0
#define GS_ERR_ARG (-1)
#define GS_ERR_HIP (-2)
#define GS_ERR_STATE (-3)
#define GS_ERR_UNSUPPORTED (-4)

typedef void* gs_stream_t;

int gs_version(void);
const char* gs_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Dense contraction on the matrix cores.  Replaces every ATen GEMM the path issues:
 * MyLinear `input @ self.weight` (models/layers.py:376), dense `adj @ x` (models/sgc.py:49),
 * nn.Linear inside PGE (models/parametrized_adj.py:57-64) and their autograd transposes.
 * C[b] = alpha*A[b]*B[b] (+bias) (relu) (mask) + beta*C[b];  b = b1*nb2 + b2.
 * ------------------------------------------------------------------------------------------- */
typedef struct gs_gemm_desc {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  long sAm, sAk;         /* A(m,k) = A[m*sAm + k*sAk]  */
  long sBk, sBn;         /* B(k,n) = B[k*sBk + n*sBn]  */
  long ldc;              /* C(m,n) = C[m*ldc + n]      */
  int nb1, nb2;          /* batch counts (>=1)         */
  long sAb1, sAb2, sBb1, sBb2, sCb1, sCb2;   /* uniform batch strides (elements) */
  const long* offA;      /* optional [nb1] extra element offsets (grouped GEMM)   */
  const long* offB;
  const long* offC;
  const int* Mb;         /* optional [nb1] per-group M (<= M)  */
  const int* Kb;         /* optional [nb1] per-group K (<= K)  */
  const int* gatherA_m;  /* optional row gather:  m -> gatherA_m[m]  (+ b1*sGb1) */
  const int* gatherA_k;  /* optional:             k -> gatherA_k[k]              */
  const int* gatherB_k;
  long sGb1;
  int kchunk;            /* >0: split K over b2 in chunks of kchunk; C gets nb2 partial slabs (sCb2) */
  float alpha, beta;
  const float* bias;     /* optional [N] (+ b1*sbias_b1), added before relu/mask */
  long sbias_b1;
  int relu;
  const float* mask;     /* optional [M][ldmask]: output kept where mask > 0 else 0 (ReLU backward) */
  long ldmask;
  float* colstat;        /* optional [tiles_m][2][N]: per-M-tile column sums / sums of squares of the written C
                            (BatchNorm batch statistics fused into the epilogue); tiles_m = gs_gemm_tiles_m(desc) */
  /* optional generated B operand (the real-side edge GEMM): B(k,n) = bgen_w[k] * [bgen_mask[bgen_idx[k]][n] > 0] *
     B[(k / bgen_div) * sBk + n]; bgen_idx / bgen_w advance by sGb1 per b1 like the gathers; needs sAm == 1, sBn == 1 */
  const float* bgen_mask;
  long bgen_ldm;
  const int* bgen_idx;
  const float* bgen_w;
  int bgen_div;
  /* operand precision of the row-panel kernel (N == 256, both operands k-contiguous, unbatched, M >= 2048; other shapes
     ignore it and compute in exact fp32): 0 = exact fp32 MFMA (default), 1 = split bf16 x3 (A = Ah + Al, B = Bh + Bl,
     C += Al Bh + Ah Bl + Ah Bh, fp32 accumulate: ~2^-16 relative operand representation), 2 = bf16 operands (Ah Bh only),
     3 = EXACT three-way split (x = h + m + l with 8 + 8 + 8 significant bits = the 24 of an fp32, so the operands are
     represented exactly; C += Al Bh + Ah Bl + Am Bm + Am Bh + Ah Bm + Ah Bh on the bf16 MFMA with fp32 accumulation: the three
     dropped cross terms are <= 2^-23 of a product, the size of fp32's own rounding — measured error against fp64 equal to
     the fp32 MFMA's, tests/test_gpu_panel.py; 6 MFMAs at 16x the fp32 rate).
     prec != 0 needs `scratch`: N * K * 4 bytes (modes 1, 2) or N * K * 6 bytes (mode 3) of device memory for the pre-split B. */
  int prec;
  void* scratch;
  /* sign bits (round 3; both optional, row-panel / edge kernels only).  signbits: the product also writes, per output row m,
     N / 32 words whose bit (n & 31) of word (n >> 5) is [C(m, n) > 0] — GS_ERR_UNSUPPORTED unless the product runs on the
     row-panel kernel.  bgen_bits: the generated B operand reads its mask from such words (row bgen_idx[k], stride bgen_ldbits
     words) instead of the fp32 rows of bgen_mask (32 B instead of 1 KiB per edge at N = 256); bgen_mask must still be given
     (the tiled fallback reads it). */
  unsigned* signbits;
  long ld_signbits;
  const unsigned* bgen_bits;
  long bgen_ldbits;
  /* 1: bgen_bits holds one row of words per EDGE SLOT (row = batch * sGb1 + k: the sign bits the hop-2 gather of the real side wrote
     for the source row it had in registers) instead of one per source node (row = bgen_idx[k]) */
  int bgen_bits_edge;
} gs_gemm_desc;

/* which kernel gs_gemm_f32 launches for this descriptor: 0 = 128x32 tile, 1 = 32x128, 2 = 64x64, 3 = 128x128, 4 = the
   one-tile-per-workgroup latency kernel, 5 = the row-panel kernel (128 rows x all 256 columns per workgroup; PGE layer 2,
   its backward and the SGC transform), 6 = the real-side edge-product kernel — test hook: parity tests assert the instantiation a shape exercises */
int gs_gemm_route(const gs_gemm_desc* d);
/* number of M tiles gs_gemm_f32 will use for this descriptor (first dimension of colstat) */
int gs_gemm_tiles_m(const gs_gemm_desc* d);

int gs_gemm_f32(const gs_gemm_desc* d, gs_stream_t stream);
/* n mutually independent products (none reads another's output).  When all of them are latency-bound (see
   gs_gemm_set_small) and n <= 4 they share ONE launch; otherwise they are launched one after the other.  Results are
   bitwise those of n gs_gemm_f32 calls. */
int gs_gemm_group_f32(const gs_gemm_desc* descs, int n, gs_stream_t stream);

/* out[i] = beta*out[i] + sum_s slabs[s*slab_stride + i]  (fixed order: bitwise reproducible). */
int gs_reduce_slabs_f32(float* out, const float* slabs, long n, int n_slabs, long slab_stride, float beta,
                        gs_stream_t stream);
/* batched form: out[b*out_stride + i] = beta*out[..] + sum_s slabs[(b*n_slabs + s)*slab_stride + i] */
int gs_reduce_slabs_batched_f32(float* out, long out_stride, const float* slabs, long n, int n_slabs, long slab_stride,
                                int nbatch, float beta, gs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Sparse aggregation.  Replaces torch_sparse.matmul / SparseTensor @ dense (spmm_sum; call sites
 * models/sgc.py:47,51, models/layers.py:41, models/base.py:55,63).
 * Y[i,:] = sum_{e in row i} val[e] * X[col[e],:]   — CSR, wave-per-row gather, fp32.
 * The transposed product (autograd of spmm) is the same call on the CSR of A^T, which the host builds
 * once (the graph is static) exactly as torch_sparse pre-computes its csr2csc permutation.
 * ------------------------------------------------------------------------------------------- */
int gs_spmm_csr_f32(const int* rowptr, const int* col, const float* val, int n_rows, const float* X, long ldx, int F,
                    float* Y, long ldy, gs_stream_t stream);

/* out[i,:] = X[idx[i],:]  — `features[n_id]` (condensation/gcond_base.py:214). */
int gs_gather_rows_f32(const float* X, long ldx, const int* idx, int n, int F, float* out, long ldo, gs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Class-wise neighbour-sampled blocks.  Replaces TransAndInd.retrieve_class_sampler
 * (dataset/loader.py:187-224 -> PyG NeighborSampler -> torch_sparse sample_adj, CPU) for ALL classes
 * in one call: per class up to `bcap` seeds drawn without replacement from the class pool, hop 1 keeps
 * <= f1 neighbours per seed, the de-duplicated frontier (seeds first) keeps <= f2 neighbours per node;
 * edge values are the full-graph normalised weights.  Layout: class-major, fixed capacity, padded
 * entries carry val = 0 and index 0.  Outermost-hop sources stay GLOBAL ids (they index X directly).
 * `--nlayers` K >= 3 (dataset/loader.py:205-210: sizes [15, 10, 5], [15, 10, 5, 5], [15, 10, 5, 5, 5]) adds nmid = K - 2 MIDDLE
 * hops between hop 1 and the outermost one: middle hop m (0 = next to hop 1) keeps <= fm[m] neighbours of every node of the
 * frontier before it, de-duplicates into a frontier of capacity capm[m] = cap(m-1) * (1 + fm[m]) (cap(-1) = cap1; its targets
 * first, like every frontier) and carries the CSC of its block for the transposed product of the backward pass.  The outermost
 * block col0 / val0 then has one row per node of the LAST middle frontier: [C][capm[nmid-1]][f2].
 * ------------------------------------------------------------------------------------------- */
#define GS_MAX_MID 3
typedef struct gs_blocks {
  int C, bcap, f1, f2, cap1;   /* cap1 = bcap*(1+f1) */
  int* n_seed;     /* [C]                 seeds actually drawn (min(bcap, pool size))            */
  int* n1;         /* [C]                 frontier size, seeds first                              */
  int* nid1;       /* [C][cap1]           global id of frontier node l (l < n_seed: the seeds)    */
  int* col1;       /* [C][bcap][f1]       hop-1 block (seed b <- frontier l): local frontier ids  */
  float* val1;     /* [C][bcap][f1]                                                               */
  int* tptr1;      /* [C][cap1+1]         CSC of the hop-1 block: in-edges of frontier node l ... */
  int* tedge1;     /* [C][bcap*f1]        ... are slots tedge1[tptr1[l] .. tptr1[l+1]), ascending */
  int* col0;       /* [C][cap1][f2]       hop-2 block (frontier l <- source): GLOBAL node ids     */
  float* val0;     /* [C][cap1][f2]         (with nmid > 0: [C][capm[nmid-1]][f2])                */
  int nmid;                    /* middle hops (nlayers - 2); 0: the two-hop layout, nothing below is read */
  int fm[GS_MAX_MID], capm[GS_MAX_MID];
  int* nm[GS_MAX_MID];         /* [C]                     size of the frontier behind middle hop m       */
  int* nidm[GS_MAX_MID];       /* [C][capm[m]]            its global ids (targets first)                 */
  int* colm[GS_MAX_MID];       /* [C][cap(m-1)][fm[m]]    middle block: local ids into frontier m        */
  float* valm[GS_MAX_MID];     /* [C][cap(m-1)][fm[m]]                                                   */
  int* tptrm[GS_MAX_MID];      /* [C][capm[m] + 1]        CSC of the middle block ...                    */
  int* tedgem[GS_MAX_MID];     /* [C][cap(m-1) * fm[m]]   ... slots ascending per node                   */
  int* scrm[GS_MAX_MID];       /* [C][capm[m] + 1]        sampler scratch (fill cursors)                 */
} gs_blocks;

/* marker: [C * n_nodes] int32 scratch that must hold 0x7fffffff on entry; restored on exit. */
int gs_sample_blocks(const int* rowptr, const int* col, const float* val, int n_nodes, const int* cls_ptr,
                     const int* cls_nodes, int* marker, const gs_blocks* out, uint64_t seed, uint32_t step,
                     gs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * dense_gcn_norm forward / backward (graphslim/utils.py:429-439 and its autograd).
 * fwd: A^ = D^-1/2 (adj + I) D^-1/2 with row-sum degrees; r_out has 2n floats: [0,n) keeps deg^-1/2,
 *      [n,2n) is scratch for the backward pass.
 * bwd: d adj from d A^.  d_adj must NOT alias d_adj_norm when n > 2048 (the large-n path stages column partials in
 *      d_adj before it is written; GS_ERR_ARG otherwise); in-place use is fine for n <= 2048.
 * ------------------------------------------------------------------------------------------- */
int gs_dense_gcn_norm_fwd(const float* adj, int n, float* adj_norm, float* r_out, gs_stream_t stream);
int gs_dense_gcn_norm_bwd(const float* adj, const float* r, const float* d_adj_norm, int n, float* d_adj,
                          gs_stream_t stream);

/* Fused multi-tensor Adam, torch.optim.Adam defaults (gcond_base.py:68-69, gcond.py:44,54-72).
 * step = 1-based step count after this update. */
int gs_adam_step_f32(float* param, const float* grad, float* m, float* v, long n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, gs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Pieces of the evaluation GCN (models/gcn.py, models/layers.py:36-51 GraphConvolution, models/base.py:80-225
 * fit_with_val / test / predict, graphslim/utils.py:516-538 accuracy).  The trainer itself is host-side
 * orchestration (graphslim_amd/models/gcn.py) over gs_gemm_f32, gs_spmm_csr_f32, gs_adam_step_f32 and these.
 * ------------------------------------------------------------------------------------------- */
/* X[i,k] = act(X[i,k] + bias[k]) in place (bias may be NULL). */
int gs_bias_act_f32(float* X, long ld, const float* bias, long n, int F, int relu, gs_stream_t stream);
/* G = (softmax(Z) - onehot(labels)) / n (may be NULL);  loss_rows[i] = -log_softmax(Z[i])[labels[i]] (may be NULL). */
int gs_softmax_nll_grad_f32(const float* Z, long ldz, const int* labels, long n, int C, float* G, float* loss_rows,
                            gs_stream_t stream);
/* out = log_softmax(Z) row-wise (F.log_softmax, models/base.py:78). */
int gs_log_softmax_f32(const float* Z, long ldz, long n, int C, float* out, long ldo, gs_stream_t stream);
/* out[k] = sum_i M[i*ld + k] (fp64 accumulation). */
int gs_colsum_f32(const float* M, long ld, long rows, int cols, float* out, gs_stream_t stream);
/* count[0] += #{i < n : argmax_k Z[idx[i]][k] == labels[i]}  (idx may be NULL = identity). */
int gs_count_correct(const float* Z, long ldz, int C, const int* idx, const int* labels, long n, int* count,
                     gs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Edge-list structure generator (MSGC, graphslim/condensation/msgc.py:31-39 self.pge = nn.Sequential(Linear, BatchNorm1d, ReLU,
 * Linear, BatchNorm1d, ReLU, Linear) applied to cat(x[rows], x[cols]) and msgc.py:224-236 get_adj_t_syn).  The matrix products of
 * the MLP are gs_gemm_f32 calls; these are the row kernels around them, forward and backward.  Host-side composition:
 * graphslim_amd/structure.py.  All of them are deterministic (no float atomics).
 * ------------------------------------------------------------------------------------------- */
/* out[e][k] = U[rows[e]][k] + V[cols[e]][k] + bias[k]: layer 0 on the concatenated pair as two node-level products
 * U = X W0[:, :d]^T, V = X W0[:, d:]^T (the same factorisation as the dense PGE, parametrized_adj.py:57-63). */
int gs_edge_pair_sum_f32(const float* U, const float* V, long ld, const float* bias, const int* rows, const int* cols, long E, int F,
                         float* out, long ldo, gs_stream_t stream);
/* Bytes of the `scratch` argument of the two BatchNorm calls below for F columns. */
size_t gs_bn_scratch_bytes(int F);
/* nn.BatchNorm1d in TRAINING mode followed by ReLU over E rows: mu = column mean, istd = 1/sqrt(biased variance + eps),
 * Y = relu(g (X - mu) istd + be).  mu, istd [F] are outputs (the backward needs them). */
int gs_bn_relu_fwd_f32(const float* X, long ldx, long E, int F, const float* g, const float* be, float eps, float* Y, long ldy,
                       float* mu, float* istd, void* scratch, gs_stream_t stream);
/* Its backward: dz = dY [Y > 0]; dbe = sum dz; dg = sum dz xhat; dX = g istd (dz - dbe/E - xhat dg/E).  dX may alias dY. */
int gs_bn_relu_bwd_f32(const float* dY, long lddy, const float* Y, long ldy, const float* X, long ldx, long E, int F,
                       const float* mu, const float* istd, const float* g, float* dX, long lddx, float* dg, float* dbe,
                       void* scratch, gs_stream_t stream);
/* out[k] = sum over E rows of X[e][k] for a TALL X (E >> F): per-chunk fp64 partial sums, then a fixed-order sum (the same two
 * stages as the BatchNorm statistics; gs_colsum_f32 walks the rows with one workgroup per 32 columns).  scratch as above. */
int gs_colsum_tall_f32(const float* X, long ld, long E, int F, float* out, void* scratch, gs_stream_t stream);
/* s[e] = sigmoid(A[e] . w + b[0])  (the last Linear(hidden, 1) + torch.sigmoid, msgc.py:226-227). */
int gs_edge_score_f32(const float* A, long ld, const float* w, const float* b, long E, int F, float* s, gs_stream_t stream);
/* adj = (M + M^T) / 2 with M[pr[e]][pc[e]] = s[e] on zeros (msgc.py:225-228; M is [n][n] scratch).  Positions pr / pc may be
 * any relabelling of (structure, node) pairs onto the rows of one block-diagonal [n][n] matrix. */
int gs_edge_scatter_sym_f32(const float* s, const int* pr, const int* pc, long E, int n, float* M, float* adj, gs_stream_t stream);
/* Backward of the two calls above: dlogit[e] = (dA[r][c] + dA[c][r]) / 2 * s (1 - s); dA2[e][k] = dlogit[e] w[k]. */
int gs_edge_score_bwd_f32(const float* dA, const int* pr, const int* pc, const float* s, const float* w, long E, int n, int F,
                          float* dlogit, float* dA2, long ld, gs_stream_t stream);
/* out[i][k] = sum over j in [ptr[i], ptr[i+1]) of Xe[perm[j]][k], in that order: the scatter-add of edge rows onto their
 * end points (backward of the gathers x[rows], x[cols]) as a reproducible segment sum over a host-built permutation. */
int gs_segment_sum_rows_f32(const float* Xe, long ld, const int* ptr, const int* perm, int n, int F, float* out, long ldo,
                            gs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * IGNR graphon learner of SGDD (graphslim/models/ignr.py:95-177): two MLP towers over all n'^2 ordered pairs — net0 on
 * cat(x_i, x_j), net1 on the pair's grid coordinates — coupled by element-wise products, mixed by ep_ratio, symmetrised,
 * squashed.  Composed from gs_gemm_f32, the BatchNorm / pair-sum / segment-sum calls above and these (graphslim_amd/structure.py).
 * ------------------------------------------------------------------------------------------- */
int gs_mul_f32(const float* a, const float* b, long total, float* z, gs_stream_t stream);                      /* z = a * b (ignr.py:167) */
/* backward of z = a * b: da = dz * b (da may alias dz); db = dz * a, added to db when accumulate_db */
int gs_mul_bwd_f32(const float* dz, const float* a, const float* b, long total, float* da, float* db, int accumulate_db,
                   gs_stream_t stream);
/* e3[e] = wa (A[e] . ua + ba[0]) + wb (B[e] . ub + bb[0]): the two Linear(nfeat, 1) heads and `(1 - ep_ratio) x + ep_ratio c`
 * (ignr.py:161-170); A, B: [E][F] contiguous */
int gs_mix_heads_f32(const float* A, const float* ua, const float* ba, float wa, const float* B, const float* ub, const float* bb,
                     float wb, long E, int F, float* e3, gs_stream_t stream);
/* backward: dA[e][k] = wa de3[e] ua[k], dB likewise; sa[e] = wa de3[e], sb[e] = wb de3[e] (then d ua = A^T sa, d ba = sum sa, ...) */
int gs_mix_heads_bwd_f32(const float* de3, const float* ua, float wa, const float* ub, float wb, long E, int F, float* dA, float* dB,
                         float* sa, float* sb, gs_stream_t stream);
/* adj = sigmoid((e3 + e3^T) / 2), zero diagonal (ignr.py:174-177; the PGE's final stage, parametrized_adj.py:66-70) and its backward */
int gs_sym_sigmoid_f32(const float* e3, int n, float* adj, gs_stream_t stream);
int gs_sym_sigmoid_bwd_f32(const float* dA, const float* adj, int n, float* de3, gs_stream_t stream);
/* nn.BatchNorm1d running statistics after a training-mode pass (momentum 0.1 upstream; unbiased batch variance) from the mu / istd
 * gs_bn_relu_fwd_f32 returned */
int gs_bn_running_update_f32(const float* mu, const float* istd, long E, int F, float eps, float momentum, float* running_mean,
                             float* running_var, gs_stream_t stream);
/* EVAL-mode BatchNorm1d + ReLU: Y = relu(g (X - running_mean) / sqrt(running_var + eps) + be) (ignr.py:151-152,211-213: the
 * inference structure); mu / istd [F] receive the constants used */
int gs_bn_relu_eval_f32(const float* X, long ldx, long E, int F, const float* g, const float* be, float eps, const float* running_mean,
                        const float* running_var, float* Y, long ldy, float* mu, float* istd, gs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Optimal-transport regulariser of SGDD (graphslim/models/ignr.py:30-41 mx_inv_sqrt, :184-209 opt_loss), evaluated in the
 * eigenbasis of the symmetric structure: adj = Q diag(lam) Q^T, so torch.svd's U f(S) V^T is Q diag(sign(lam) f(|lam|)) Q^T.  The
 * two symmetric eigen-decompositions are library calls (rocSOLVER ssyevd, bound in graphslim_amd/structure.py); the products are
 * gs_gemm_f32; these are the spectrum functions around them.
 * ------------------------------------------------------------------------------------------- */
/* out[0..n) = g_i = sign(lam_i) |lam_i|^-1/2 (Ly_inv_rt's spectrum), [n..2n) = g'_i, [2n..3n) = d(1/lam_i)/dlam_i, [3n] = trace(Ly_inv);
 * entries whose |lam| equals the smallest one are zeroed when that one is below sv_eps (0.009 upstream, ignr.py:32-38). */
int gs_ot_spectrum_f32(const float* lam, int n, float sv_eps, float* out, gs_stream_t stream);
/* value[0] = |trace[0] n - 2 sum sqrt(max(mu_k, 2e-20))| (ignr.py:205-208); acc[0] += value (may be NULL);
 * dmu[k] = scale sign(.) (-1 / sqrt(mu_k)) (0 where clamped), dmu[n] = scale sign(.) n = d value / d trace.  dmu: n + 1 floats. */
int gs_ot_value_f32(const float* mu, const float* trace, int n, float scale, float* dmu, float* value, double* acc, gs_stream_t stream);
/* x <- clamp(x, -bound, bound), NaN -> 0: keeps an fp32 overflow of the regulariser's products (a rank-deficient corner's pseudo-
 * inverse holds reciprocals of rounding-level singular values, 1e14 measured) from reaching the eigen-decomposition as inf / NaN */
int gs_clamp_finite_f32(float* x, long n, float bound, gs_stream_t stream);
/* B[i][k] = A[i][k] v[k] */
int gs_scale_cols_f32(const float* A, const float* v, int rows, int cols, float* B, gs_stream_t stream);
/* K[i][j] <- F_ij K[i][j] + [i == j] dtr[0] hp[i] with the divided differences F_ij = (g_i - g_j) / (lam_i - lam_j) (g'_i on the
 * diagonal and for equal eigenvalues): the derivative of Q diag(g(lam)) Q^T in the eigenbasis (Daleckii-Krein). */
int gs_ot_divdiff_f32(float* K, const float* lam, const float* g, const float* gp, const float* hp, const float* dtr, int n,
                      gs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The GCond engine: one object per GPU holding the learnable synthetic graph (X~, PGE), the
 * per-epoch SGC model and all scratch.  One gs_gcond_outer_step == one iteration of the loop body
 * condensation/gcond.py:47-72 (PGE forward + dense_gcn_norm, train_class over all classes
 * (gcond_base.py:156-241) with match_loss (condensation/utils.py:12-106), the second-order backward
 * into X~ and the PGE parameters, one Adam step on PGE (epoch % 50 < 10) or X~, PGE.inference, and
 * `inner_loop` SGC steps on (X~, A^)).
 * ------------------------------------------------------------------------------------------- */
typedef struct gs_gcond_config {
  int n_nodes, d, ldx;     /* real graph: X is [n_nodes][ldx], ldx >= d                      */
  int nclass, n_syn;
  int ntrans;              /* SGC transforms T: 1 .. 4 (models/sgc.py:22-35; 3 / 4 = one / two hidden-to-hidden transforms between
                              the first and the last: hidden = 256, dis_metric 'ours', nlayers = 2, d % 4 == 0, no with_bn / dropout,
                              unsharded; io.model_params = [W1 b1 | Wm_0 bm_0 | .. | Wl bl]) */
  int hidden;              /* SGC hidden width for T = 2                                       */
  int pge_hidden;          /* parametrized_adj.py:9-17                                         */
  int bcap, f1, f2;        /* loader.py:187,197-210: 256 seeds, fan-outs (f1 from seeds)       */
  int dis_metric;          /* 0 = ours, 1 = mse, 2 = cos (condensation/utils.py:33-63)         */
  int inner_loop;
  float lr_feat, lr_adj, lr_model;
  uint64_t seed;
  int external_blocks;     /* 1: blocks are supplied with gs_gcond_set_blocks (parity mode)    */
  int condense_model;      /* 0 = SGC (models/sgc.py), 1 = GCN (models/gcn.py:8-23) with `nlayers` = 2 .. 4 GraphConvolution layers, one per
                              sampled hop: parameters laid out like SGC with as many transforms ([W1 b1 | W2 b2 | .. | W_L b_L]) */
  int pge_chunks;          /* <= 1: one BatchNorm batch over all n'^2 pairs; 5: the chunked PGE of reddit r >= 0.01
                              (parametrized_adj.py:41-55: np.array_split chunks, per-chunk batch statistics); not with row sharding */
  int with_bn;             /* 1: SGC (ntrans = 2) with a BatchNorm1d between its transforms (models/sgc.py:27-33), which GCond
                              always runs in eval mode with its initial running statistics (gcond_base.py:261-285 check_bn,
                              gcond.py:50): a per-column affine map with TRAINABLE weight / bias.  model_params then is
                              [W1, b1, W2, b2, bn.weight (h), bn.bias (h)] (nn.Module.parameters() order) */
  int nlayers;             /* propagations K of the condensation model (models/sgc.py:45-51; `--nlayers`): 0 or 2 = two hops (f1, f2);
                              3 .. 5 (SGC only) = K sampled hops on the real side — f1 from the seeds, fmid[0 .. K-3] between, f2
                              outermost (dataset/loader.py:205-210) — and A^^K on the synthetic side */
  int fmid[GS_MAX_MID];
  float dropout;           /* models/base.py:20 / models/sgc.py:42: F.dropout behind every transform but the last, active inside train_class
                              (one fresh mask per class on the real batch and one per class on the synthetic graph) and in every inner
                              step (gcond.py:45: training mode).  0 (every shipped configuration): none.  In (0, 1): SGC with two
                              transforms, or the GCN condensation model with nlayers 2 .. 4 (models/base.py:59: nlayers - 1 masks per
                              forward), no BatchNorm; SGC with one transform has no such layer and the value has no effect, as
                              upstream.  torch's generator cannot be replayed, so the masks are the counter-based function of (seed,
                              iteration, kind + 3 x hidden layer, class | step, row, column) stated in csrc/gs_kernels.h (gs_drop_mult4): Bernoulli(1 - p)
                              entries scaled by 1 / (1 - p), p resolved to 1 / 65536 */
} gs_gcond_config;

typedef struct gs_gcond_io {
  const float* X;          /* [n_nodes][ldx]                                                   */
  const int* rowptr;       /* normalised adjacency CSR (utils.py:403-413), [n_nodes+1]         */
  const int* col;
  const float* val;
  const int* cls_ptr;      /* [nclass+1] seed pools per class (loader.py:188-195)              */
  const int* cls_nodes;
  const int* labels_syn;   /* [n_syn] int32, grouped by class (gcond_base.py:93-111)           */
  const int* syn_lo;       /* [nclass] first row of class c in labels_syn                      */
  const int* syn_hi;       /* [nclass] one past the last row                                   */
  float* feat_syn;         /* [n_syn][d]   learnable X~                                        */
  float* pge_params;       /* flat, nn.Module.parameters() order: W0[Ph][2d] b0[Ph] W1[Ph][Ph] b1[Ph]
                              w2[Ph] b2[1] bn1.weight[Ph] bn1.bias[Ph] bn2.weight[Ph] bn2.bias[Ph] */
  float* model_params;     /* flat: T=1: W[d][C] b[C];  T=2: W1[d][h] b1[h] W2[h][C] b2[C] (+ bn.weight[h] bn.bias[h] with with_bn) */
  float* adj_syn;          /* [n_syn][n_syn] raw PGE output of the last inference pass (gcond.py:64,77) */
  void* workspace;
  size_t workspace_bytes;  /* >= gs_gcond_workspace_bytes(cfg)                                 */
} gs_gcond_io;

typedef struct gs_gcond gs_gcond;

int gs_gcond_workspace_bytes(const gs_gcond_config* cfg, size_t* bytes);
int gs_gcond_create(const gs_gcond_config* cfg, const gs_gcond_io* io, gs_stream_t stream, gs_gcond** out);
void gs_gcond_destroy(gs_gcond* e);

/* gcond.py:42-45: call after writing fresh theta into io.model_params; resets the model's Adam state. */
int gs_gcond_new_epoch(gs_gcond* e, gs_stream_t stream);
/* Parity mode: adopt caller-built blocks (same layout as gs_sample_blocks writes). */
int gs_gcond_set_blocks(gs_gcond* e, const gs_blocks* blocks, gs_stream_t stream);
/* PGE.inference at the end of iteration i (gcond.py:63-65) and `adj_syn = pge(self.feat_syn)` at the start of iteration i+1
 * (gcond.py:48) evaluate the same function on the same (X~, theta_PGE).  With forward reuse on, gs_gcond_outer_step adopts the
 * cached pass instead of recomputing it (bitwise the same values).  Off by default; call gs_gcond_invalidate_forward after writing
 * X~ or theta_PGE from the host (and BEFORE the write too when gs_gcond_set_hoist_forward is on, see there).  gs_gcond_forward_cached
 * returns 1 when the next step would adopt. */
int gs_gcond_set_forward_reuse(gs_gcond* e, int on);
/* Opt-in (library default 0 = off): with on = 1 a gs_gcond_outer_step* call that runs inner steps also issues the NEXT iteration's
 * training forward `adj_syn = pge(feat_syn)` (gcond.py:48: a function of X~ and theta_PGE only, both final once this iteration's
 * optimiser has stepped) on the engine's third, lowest-priority stream beside its inner SGC steps; the next call waits for it instead
 * of computing it (unsharded engines without forward reuse).  THE CONTRACT THIS CHANGES: after such a call has returned a kernel of the
 * engine may still be READING the caller-owned io.feat_syn / io.pge_params — a caller that opted in must call
 * gs_gcond_invalidate_forward BEFORE it writes either buffer (the call waits for the pass and drops it).  With the default (off)
 * nothing of the engine reads those buffers between two steps: a C caller may write them freely, as through round 4.  The Python host
 * mirror (graphslim_amd.engine.GCondEngine) opts in for every engine it creates and brackets its own writes (set_feat / set_pge).
 * Values: those of the un-hoisted step up to the BatchNorm-2 column sums' partition (rounding level; repeat runs bitwise equal).
 * on = 2: additionally the edge product plans for 7/8 of the workgroup slots (measurement). */
int gs_gcond_set_hoist_forward(gs_gcond* e, int on);
/* 1 when the last gs_gcond_outer_step* call issued the NEXT iteration's training forward ahead and the next call will wait for it
 * instead of computing it.  Any other entry point that touches what that pass reads or writes waits for it and drops it. */
int gs_gcond_forward_hoisted(gs_gcond* e);
/* cfg.dropout != 0: the iteration counter inside every dropout mask's key — 0 at creation, advanced by every gs_gcond_outer_step* call
 * (the staged calls read it and leave it alone).  Set it to resume a run at iteration i with the masks that run would have drawn, or
 * to pin the masks of staged calls (parity tests). */
int gs_gcond_set_dropout_iter(gs_gcond* e, unsigned long long iter);
int gs_gcond_get_dropout_iter(gs_gcond* e, unsigned long long* iter);
/* In-situ timing of the dominant kernel (PGE layer-2 GEMM, 2 n'^2 Ph^2 FLOP per launch): while on, HIP events are recorded
 * on the launch stream around each of its launches inside the outer steps (at most 256); _read synchronises them and
 * returns the summed duration [ms] and the launch count since the probe was switched on.  bench.py's roofline uses it. */
int gs_gcond_probe(gs_gcond* e, int on);      /* on > 1: the event brackets on every on-th launch only (two stream markers per bracket cost
                                               * ~5 us of the dependent chain); the device-side spans of _read_spans / _read_pass on every launch */
int gs_gcond_probe_read(gs_gcond* e, double* total_ms, long* launches);
int gs_gcond_probe_read_spmm(gs_gcond* e, double* total_ms, long* launches);   /* the hop-2 gather SpMM of the real side, on its own stream */
/* Device-side spans of the same launches (first workgroup start .. last workgroup end, s_memrealtime at 100 MHz), for the
   launches whose kernel records one (the generated-A1 forward): the kernel's duration as a rocprofv3 kernel trace reports it.
   The HIP-event bracket above additionally contains the time the launch queued behind the other stream's resident workgroups. */
int gs_gcond_probe_read_spans(gs_gcond* e, double* total_ms, long* launches);
/* the same per pass of the probed product (0: training forward, 1: inference forward): event bracket, launches, device span, spans recorded */
int gs_gcond_probe_read_pass(gs_gcond* e, int pass, double* bracket_ms, long* launches, double* span_ms, long* span_launches);
int gs_gcond_invalidate_forward(gs_gcond* e);
int gs_gcond_forward_cached(gs_gcond* e);
int gs_gcond_adopt_cached_forward(gs_gcond* e, gs_stream_t stream);
/* gs_gcond_outer_step runs the real-side branch (sampling + class gradients) on an engine-owned second stream and joins
 * it by event before the distance statistics (on = 1, the default).  on = 2: only the sampling of the NEXT iteration's blocks
 * runs on the second stream, the class gradients stay on the caller's stream; on = 0: one stream.  Results are identical;
 * measured 2.12 / 2.30 / 2.36 ms per iteration for 1 / 2 / 0 at the bench shape (profiles/r02/ab_overlap.txt). */
int gs_gcond_set_overlap(gs_gcond* e, int on);
/* Operand form of the large contractions of the iteration (PGE layer 2 forward, its backward, dW1 in PGE epochs, the SGC
 * transform H1 = relu(X W1 + b1), the real-side edge product): 0 = fp32 matrix pipe (the LIBRARY's default = the reference's
 * arithmetic), 1 = two-way bf16 split, 3 products (operands to 2^-16), 2 = bf16 operands (BASELINE.json configs[1]), 3 = three-way
 * bf16 split, 6 of the 9 cross products (operands represented exactly; the dropped terms and the fp32 accumulation leave an error
 * measured equal to the fp32 pipe's, tests/test_gpu_panel.py — not bitwise fp32).  See gs_gemm_desc.prec.  Everything else stays
 * fp32.  NOTE the split: a C caller gets mode 0 unless it asks; the Python host mirror (graphslim_amd.engine.DEFAULT_OPERAND_MODE,
 * environment GS_OPERAND_MODE) selects 3 for every engine it creates, and bench.py's headline value is measured in mode 3 (its
 * `dtype` field says so; `operand_modes` carries mode 0 beside it). */
int gs_gcond_set_operand_mode(gs_gcond* e, int mode);
/* Full-neighbour mode (dataset/loader.py:204 `sizes = [-1, -1]`, commented out upstream; SURVEY section 8e "mode F") for SGC with
 * ntrans = 1: the model is linear up to the softmax, so the 2-hop full-neighbourhood block product of a seed is row `seed` of
 * S_full [n_nodes][ld] = A^ (A^ X) and its bias factor row `seed` of rs_full [n_nodes] = A^ (A^ 1) — constants of the whole job
 * that the caller computes once with gs_spmm_csr_f32 (row-sharded over the ranks of a node, graphslim_amd/parallel.py) and keeps
 * alive.  With them set, sampling draws the class seeds only (the same seeds as the sampled mode) and the real-side stage
 * gathers their rows; NULL returns to neighbour sampling.  GS_ERR_UNSUPPORTED for ntrans = 2 / GCN (a ReLU between the
 * transforms would need one full-graph SpMM per class and iteration in the backward). */
int gs_gcond_set_full_neighbour(gs_gcond* e, const float* S_full, long ld, const float* rs_full);
/* One outer-loop iteration; adds the matching loss to the device-side accumulator. */
int gs_gcond_outer_step(gs_gcond* e, int epoch, gs_stream_t stream);
/* The sibling gradient-matching methods reuse train_class unchanged and differ in the update rule and in whether a
 * structure is learned (condensation/doscond.py:40-50, gcondx.py:30,44-66, doscondx.py:30,42-49):
 * gs_gcond_outer_step(e, epoch) == gs_gcond_outer_step_ex(e, epoch%50<10, !(epoch%50<10), 1, 1). */
int gs_gcond_outer_step_ex(gs_gcond* e, int step_pge, int step_feat, int run_inference, int run_inner, gs_stream_t stream);
/* adj_syn = I (no PGE): GCondX / DosCondX. */
int gs_gcond_set_identity_structure(gs_gcond* e, int on, gs_stream_t stream);
/* Structure supplied by the caller instead of the engine's dense PGE (MSGC, msgc.py:98,109): adj [n_syn][n_syn] raw, symmetric,
 * no self loops.  inference = 0: the structure gs_gcond_stage_syn_match differentiates (it leaves d loss / d adj in buffer
 * "d_adj" and the direct feature gradient in "d_feat"); inference = 1: the structure of gs_gcond_stage_inner_steps, also copied
 * to io.adj_syn.  The caller drives the iteration stage by stage (sample / real grads / syn_match / syn_finish / inner steps)
 * and owns the optimiser steps of X~ and of its generator; the PGE stages are not run. */
int gs_gcond_set_structure(gs_gcond* e, const float* adj, int inference, gs_stream_t stream);
/* The synthetic graph is B disconnected blocks of n_block rows (MSGC's batch of structures as one block-diagonal graph): pos
 * [B][n_block] int32 on the device = row of node i of block b (a bijection onto [0, n_syn)).  The n_syn^3 products of an iteration
 * (A^ A^ twice, dQ A^^T + A^^T dQ) then run per block — 2 B n_block^3 instead of 2 (B n_block)^3 FLOP each — and so do the products of
 * A^ A^ with [n_syn x F] matrices outside the class-by-class terms.  scratch: 3 B n_block^2 + 2 n_syn max(d, hidden) floats,
 * caller-owned and kept alive; pos = NULL returns to dense products.  GS_ERR_UNSUPPORTED for the GCN model. */
int gs_gcond_set_block_structure(gs_gcond* e, const int* pos, int B, int n_block, float* scratch);
/* Synchronises `stream`; returns the accumulated loss and the outer-step count since the last reset
 * (host pointers) and resets both.  This is the loss_avg bookkeeping of gcond.py:52,74. */
int gs_gcond_read_loss(gs_gcond* e, double* loss_sum, long* steps, gs_stream_t stream);

/* Stage-level entry points (what gs_gcond_outer_step runs, in this order); exposed so that each
 * stage can be checked against the oracle in isolation. */
int gs_gcond_stage_sample(gs_gcond* e, gs_stream_t stream);
int gs_gcond_stage_real_grads(gs_gcond* e, gs_stream_t stream);
/* stage_sample + stage_real_grads forked onto the engine's second stream (when overlap is on; otherwise run in line), plus
 * the prefetch of the next iteration's blocks; the stage that first reads the real-side gradients joins it. */
int gs_gcond_stage_real_async(gs_gcond* e, gs_stream_t stream);
int gs_gcond_stage_pge_forward(gs_gcond* e, int inference, gs_stream_t stream);
int gs_gcond_stage_syn_match(gs_gcond* e, gs_stream_t stream);
int gs_gcond_stage_pge_backward(gs_gcond* e, int need_param_grads, int need_feat_grad, gs_stream_t stream);
int gs_gcond_stage_update(gs_gcond* e, int epoch, gs_stream_t stream);
int gs_gcond_stage_inner(gs_gcond* e, gs_stream_t stream);

/* ---- multi-GPU (one process per GPU): shard the independent units across ranks --------------------------------
 * Classes [c_lo, c_hi) are this rank's share of the sampler, of the real-side gradients and of the per-class
 * synthetic terms of train_class (gcond_base.py:210-239: the class terms are independent summands); pair rows
 * [r_lo, r_hi) its share of the PGE n' x n' grid.  Everything that crosses ranks is a SUM of per-rank partials in a
 * contiguous buffer, so the only collective is all-reduce(sum) (RCCL) on these gs_gcond_buffer views, between parts:
 *   pge_forward_part 1 -> "ar_bn2" (fp64, 2*Ph) -> part 2 -> "e3" (n'^2) -> part 3
 *   syn_match -> "ar_syn" ([d_feat | d_adj | loss_c]) -> syn_finish
 *   pge_backward_part 1 -> "ar_l3" (fp64, 3*Ph) -> part 2 -> "ar_marg" (2*n'*Ph) [+ "ar_w1" (Ph^2+Ph) when the PGE
 *   is stepped] -> part 3.     The reference has no distributed code; this is new. */
int gs_gcond_set_shard(gs_gcond* e, int c_lo, int c_hi, int r_lo, int r_hi);
int gs_gcond_stage_pge_forward_part(gs_gcond* e, int inference, int part, gs_stream_t stream);
int gs_gcond_stage_pge_backward_part(gs_gcond* e, int need_param_grads, int need_feat_grad, int part, gs_stream_t stream);
int gs_gcond_stage_syn_finish(gs_gcond* e, gs_stream_t stream);       /* books the (all-reduced) loss_c into the accumulator */
int gs_gcond_stage_inner_steps(gs_gcond* e, gs_stream_t stream);      /* the SGC steps of gs_gcond_stage_inner, without PGE.inference */

/* ---- the exchange itself, inside the library -----------------------------------------------------------------------
 * gs_comm: all-reduce(sum) between the ranks of a sharded run, issued by the library on the engine's stream.  Transports (+ ipc, below):
 *   rccl      ncclAllReduce (RCCL over xGMI; bound with dlopen, the library has no link-time dependency on it).  Rank 0 calls
 *             gs_comm_unique_id and hands the 128 bytes to every rank (host side: any broadcast, e.g. torch.distributed);
 *             every rank then calls gs_comm_create_rccl with its device current.
 *   callback  `fn(user, buf, count, dtype, stream)` performs the exchange (dtype 0 = fp32, 1 = fp64; returns 0 on success):
 *             tests over gloo between processes that share one GPU, in-process virtual ranks.
 *   null      no exchange (world 1; timing-only virtual ranks).
 * gs_gcond_outer_step_sharded = gs_gcond_outer_step for this rank's share (gs_gcond_set_shard), with the exchanges of the
 * part-wise protocol above issued in between: ONE host call per rank and outer iteration; `_ex` takes the update rule of
 * gs_gcond_outer_step_ex.  Exchange points (`tag` of gs_comm_allreduce / index of gs_comm_read_timing's arrays [8]):
 * 0 BN-2 sums, 1 pair scores, 2 [dX~ | d adj | loss_c], 3 BN-2 backward sums, 4 marginals (+ dW1 | db1), 7 other. */
typedef struct gs_comm gs_comm;
typedef int (*gs_allreduce_fn)(void* user, void* buf, long count, int dtype, gs_stream_t stream);
int gs_comm_unique_id(void* id128);                                   /* host: 128 bytes out */
int gs_comm_probe(int* rccl_version);                                 /* is the RCCL binding usable here? (dlopen + symbols + ncclGetVersion: allocates
                                                                         nothing, not collective); rccl_version may be null */
int gs_comm_create_rccl(const void* id128, int rank, int world, gs_comm** out);
int gs_comm_create_callback(gs_allreduce_fn fn, void* user, int rank, int world, gs_comm** out);
int gs_comm_create_null(int rank, int world, gs_comm** out);
/* ipc transport: a one-shot peer-read all-reduce over arenas every rank maps from every other rank (hipIpc; xGMI peer access between
   the GPUs of one node) for the sub-MB joins — one launch per exchange, no host round trip, ranks summed in ascending order on every
   rank (bitwise equal replicas).  gs_comm_ipc_arena allocates THIS rank's arena on the current device (two slots of slot_bytes) and
   returns the 64-byte handle the other ranks open it with; the host side gathers the handles of all ranks in rank order
   ([world][64], e.g. torch.distributed.all_gather_object) and calls gs_comm_create_ipc (world <= 16), which takes ownership of the
   arena.  Exchanges larger than slot_bytes go to `fallback` (an RCCL / callback communicator, may be NULL: GS_ERR_UNSUPPORTED then).
   The transport is chosen from the element COUNT alone (the same on every rank); a buffer that is not 16-byte aligned takes the
   kernel's 4-byte path.  A wait for a peer's flag is bounded (~2 s) and a timeout is FATAL for the communicator: that exchange leaves
   its buffer untouched, the error word is sticky, and every later gs_comm_allreduce on it returns GS_ERR_STATE (RCCL would simply
   wait: use it when ranks can be seconds apart).  gs_comm_ipc_error: 0, or r + 1 for the peer r that did not arrive (waits for the
   device); gs_comm_check: the same verdict as a return code, from a host-visible word without synchronising (any communicator kind;
   what a driver polls once per epoch). */
int gs_comm_ipc_arena(size_t slot_bytes, void** arena, void* handle64);
int gs_comm_create_ipc(void* arena, const void* handles, size_t slot_bytes, int rank, int world, gs_comm* fallback, gs_comm** out);
int gs_comm_ipc_error(gs_comm* c, int* code);
int gs_comm_check(gs_comm* c);
void gs_comm_destroy(gs_comm* c);
int gs_comm_rank(const gs_comm* c);
int gs_comm_world(const gs_comm* c);
int gs_comm_allreduce(gs_comm* c, void* buf, long count, int dtype, int tag, gs_stream_t stream);
int gs_comm_set_timing(gs_comm* c, int on);                           /* event pairs around every exchange from now on */
int gs_comm_read_timing(gs_comm* c, long* calls, long* bytes, double* total_us);   /* host arrays [8]; synchronises */
/* the same + timed_calls[8]: how many of `calls` total_us covers (the event pool holds 8 192 pairs; later calls are counted, not timed) */
int gs_comm_read_timing2(gs_comm* c, long* calls, long* bytes, double* total_us, long* timed_calls);
int gs_gcond_outer_step_sharded(gs_gcond* e, int epoch, gs_comm* comm, gs_stream_t stream);
int gs_gcond_outer_step_sharded_ex(gs_gcond* e, int step_pge, int step_feat, int run_inference, int run_inner, gs_comm* comm,
                                   gs_stream_t stream);







/* Named views into engine-owned scratch for the parity tests (device pointer + element count):
 * "gw_real" [C][|theta|], "gw_syn" [C][|theta|], "loss" [1], "d_feat" [n'][d] (total gradient wrt X~),
 * "d_feat_direct", "d_adj_norm" [n'][n'], "d_adj" [n'][n'], "d_pge" [|theta_pge|], "adj_norm", "adj_raw",
 * "blocks" (gs_blocks struct copied to *ptr as host struct is NOT supported — use gs_gcond_get_blocks). */
int gs_gcond_buffer(gs_gcond* e, const char* name, void** ptr, long* count);
int gs_gcond_get_blocks(gs_gcond* e, gs_blocks* out);

#ifdef __cplusplus
}
#endif
#endif /* GSLIM_H */
