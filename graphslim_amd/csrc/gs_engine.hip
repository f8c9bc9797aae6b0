// The GCond engine: one outer-loop iteration of condensation/gcond.py:47-72 as a fixed sequence of
// HIP launches on one stream (no autograd: the second-order gradient of the matching loss is
// evaluated in closed form, see oracle/analytic.py for the algebra and tests/test_analytic.py for
// its check against autograd).  All classes are processed together: the real-side blocks live in
// class-major fixed-capacity arrays, the synthetic side uses grouped GEMMs over the contiguous row
// range of each class.  Everything stays resident in HBM; the host only launches.
#include <new>

#include "gs_kernels.h"
#include "gs_match.h"

extern "C" int gs_reduce_slabs_batched_f32(float*, long, const float*, long, int, long, int, float, gs_stream_t);
extern "C" int gs_reduce_slabs_f32(float*, const float*, long, int, long, float, gs_stream_t);
int gs_reduce_slabs_active(float*, long, const float*, long, int, long, int, const int*, int, gs_stream_t);
bool gs_edge_eligible(const gs_gemm_desc& p);                                                      // gs_edge.hip
int gs_edge_plan(const int* n1, int f2, int nclass, int target, int* kcount, int* plan, hipStream_t s);
int gs_edge_launch_plan(const gs_gemm_desc& d, const int* plan, int max_slabs, hipStream_t s);
int gs_reduce_slabs_plan(float* out, long out_stride, const float* slabs, long n, long slab_stride, int nbatch, const int* plan, hipStream_t s);
int gs_reduce_slabs_pair(float*, const float*, long, int, long, float*, const float*, long, int, long, gs_stream_t);
struct gs_panel_gen { const float *mu2, *is2, *g2, *be2, *w2, *dbe2, *dg2, *de3; long total_rows; float* table; int table_ready; };
struct gs_panel_a1 { const float *U, *V; long ld; const float *b0, *mu1, *is1, *g1, *be1; float* table; const float *Us, *Vs; int b_ready; };
int gs_panel_marginals(const float*, long, const float*, long, const float*, long, int, int, int, int, float*, float*, int, void*,
                       const gs_panel_gen*, const gs_panel_a1*, hipStream_t);
int gs_dw1_generated(const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, int, int, float*,
                     float*, long, int*, const float*, hipStream_t);
int gs_pge_scale_uv(const float*, const float*, const float*, long, int, float*, float*, hipStream_t);
int gs_gemm_group2_f32(const gs_gemm_desc* descs, const gs_gemm_desc* second, int n, gs_stream_t stream);
int gs_gemm_group2_tiles_f32(const gs_gemm_desc*, const gs_gemm_desc*, int, const int*, const int*, int, int, gs_stream_t);
#define GS_MAX_CLASSES_TILES 1024
int gs_panel_forward_a1(const gs_panel_a1*, int, int, const float*, long, const float*, float*, long, int, int, float*, int*, int, void*,
                        unsigned long long*, int, hipStream_t);
int gs_panel_tiles(int);
struct gs_zg_args {
  const float* A; long sAm, sAk; const float* B; long sBk, sBn; int n, K, C; const float* rs; const float* bias;
  const int* labels; const int* syn_lo; const int* syn_hi; int divide_by_n; float* P; float* G;
};
struct gs_adam_ep { float* theta; float* m; float* v; const float* gbase; float step_size, inv_sqrt_bc2, b1, b2, eps; };
int gs_chain_zg(const gs_zg_args*, hipStream_t);
int gs_chain_group(const gs_gemm_desc*, int, const gs_adam_ep*, hipStream_t);
int gs_panel_rows(const float*, long, const float*, long, const float*, int, float*, long, int, int, const int*, const int*, int, long, int,
                  void*, unsigned*, long, hipStream_t);
int gs_dedup_last(const int*, const float*, const int*, int, int, int, int*, int, int*, int*, int*, int*, int*, int, int*, int*, void*, void*,
                  int*, hipStream_t);   // gs_sampler.hip
extern "C" long gs_dedup_scratch_ints(int nclass, int cap0);
int gs_edge_launch_entries(const gs_gemm_desc& d, const void* erow, const void* ew, const int* plan, int max_slabs, hipStream_t s);   // gs_edge.hip
// gs_inner.hip: the inner SGC steps as one persistent launch
struct gs_inner_ws {
  float *XT, *W1T, *WlT, *pb, *TT, *GT, *RT, *H1T, *dH1T, *rs, *Q, *QT;
  unsigned* sync;
  int npad, dpad, Cp;
};
extern "C" long gs_inner_persist_ws_floats(int n, int d, int h, int C);
extern "C" int gs_inner_persist_eligible(int n, int d, int h, int C, int steps);
extern "C" void gs_inner_persist_carve(float* ws, int n, int d, int h, int C, gs_inner_ws* o);
int gs_inner_persist_launch(const gs_inner_ws& w, int n, int d, int h, int C, int steps, const float* X, long ldx, const int* labels,
                            float* theta, float* m, float* v, float* g, long oW1, long ob1, long oWl, long obl, const float* step_size,
                            const float* inv_sqrt_bc2, float b1, float b2, float eps, unsigned* counter_host, hipStream_t s);
extern "C" int gs_comm_allreduce(gs_comm*, void*, long, int, int, gs_stream_t);
extern "C" int gs_comm_world(const gs_comm*);

namespace {

struct Carver {
  char* base;
  size_t off;
  template <class T>
  T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

gs_gemm_desc mk(const float* A, long sAm, long sAk, const float* B, long sBk, long sBn, float* C, long ldc, int M, int N,
                int K) {
  gs_gemm_desc g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.sAm = sAm; g.sAk = sAk;
  g.B = B; g.sBk = sBk; g.sBn = sBn;
  g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K;
  g.nb1 = g.nb2 = 1;
  g.alpha = 1.f;
  return g;
}

}  // namespace

struct gs_gcond {
  gs_gcond_config cfg;
  gs_gcond_io io;
  int N, d, ldx, C, n, T, h, Ph, bcap, f1, f2, cap1, F, Fp;     // Fp = F rounded up to 4: row stride of the real side's P0 / Sr
  long P, Ppge;
  long oW1, ob1, oWl, obl;                                   // model parameter offsets
  // ntrans >= 3 (models/sgc.py:32-35): `deep` = ntrans - 2 hidden-to-hidden transforms between the first and the last; T stays 2
  // (every two-transform buffer exists), the stages dispatch on `deep` (gs_deep.inc).  theta = [W1 b1 | Wm_0 bm_0 | .. | Wl bl]
  int deep; long oWm[2], obm[2];
  // GCN condensation model with nlayers = L > 2 (models/gcn.py:8-23): gdeep = L - 2 hidden-to-hidden GraphConvolution layers, same
  // parameter layout; the stages dispatch to gs_gcnk.inc
  int gdeep;
  bool gk;                           // the GCN model runs the layer loops of gs_gcnk.inc: nlayers > 2, or dropout (any nlayers)
  float* gkd[7];                     // [C][n'][h] each: per-class activations / adjoints of the synthetic forwards under dropout
  float *gk_sbuf[3];                   // [n'][h] s_k = A a_{k-1}, k = 2 .. L
  float *gk_slabA;                     // [C][n'][n'] per-class contributions to d A^
  float *gr_P1, *gr_H[4], *gr_T, *gr_dT, *gr_dHa, *gr_dHb;      // real side: [C][capL][d], [C][rows_k][h] per layer, scratch of capL rows
  float *Hmid_full[2];                 // [N][h] hidden activations of layers 2 .. 1 + deep for every node
  float *WmT;                          // [h][h] scratch: a hidden-to-hidden weight transposed (k-contiguous operand)
  float *Hmid_s[2];                    // [n'][h] the same on the synthetic rows
  float *dp_cls[8], *dp_n[6];          // [C][n'][h] / [n'][h] scratch of the deep synthetic match
  float *dr_out;                       // [C][max(d, h)][h] a weight's per-class gradients, contiguous (copied into gw_real's rows of stride P)
  float *dr_a, *dr_b, *dr_c;           // [C][cap0][h] per-source backward rows of the deep real side (sum of in-edges, masked, through W^T)
  float *ones0;                        // [C][cap0] unit weights of the per-source edge products
  long pW0, pb0, pW1, pb1, pw2, pb2, pg1, pbe1, pg2, pbe2;   // PGE parameter offsets
  gs_pdesc pd;
  int max_nc;
  int nsplit_r, kchunk_r, nsplit_w, kchunk_w, nblk_stats, rows_per_block;
  int edge_target, edge_cap;           // balanced slabs of the real-side edge product: slabs aimed at / slabs the buffer holds (gs_edge_plan)
  int* edge_plan;                      // [2 C + 1] device ints
  int K, nmid, fmid[GS_MAX_MID], capm[GS_MAX_MID], capL;      // propagations (nlayers), middle hops, their fan-outs / frontier capacities, rows of the outermost block
  float *Pmid[GS_MAX_MID], *rsmid[GS_MAX_MID], *Ymid[GS_MAX_MID];   // per middle hop m: the forward rows / row sums of frontier m - 1 ([C][cap(m-1)]), the backward rows of frontier m ([C][capm[m]][h])
  float *Apow, *Apow_t;                // nlayers != 2: [K][n'^2] powers A^, A^^2, .. of the normalised structure (training / scratch)
  bool real_beside;                    // this real-side stage runs beside the training pass of the PGE forward (fused step, forward recomputed)
  // real side
  gs_blocks blk;
  int *marker, *kcount;
  int *h1_flag, *h1_list, *h1_count;   // rows of H1full this rank's sampled blocks gather (sharded ranks: only those are computed)
  unsigned* H1bits;                    // [N][h / 32] sign bits of H1full, written by its product's epilogue: the edge product's mask
  double* sum_part;                    // [128] scratch of sum_f32
  bool tiles_ok;                     // tile_b / tile_my / tile_start are filled (C <= GS_MAX_CLASSES_TILES)
  bool skip_dfeat;                   // this call of the matching stage need not leave d loss / d X~: set by outer_step_impl (sharded or not) around its
                                     // stage call and cleared right behind it; a class shard zeroes d_feat first, so the exchanged buffer holds zeros
  bool l3_direct;                    // part 1 of the PGE backward left the statistics partials for part 2's prep launch to reduce
  bool bn2_done;                     // part 1 of the PGE forward has already left the BatchNorm-2 statistics (unsharded, fused)
  bool fold_finish;                  // this call of the matching stage also books the loss (unsharded outer_step_impl only; cleared right behind the call)
  int knob[GS_KNOB_COUNT];             // per-engine value of a measurement knob (gs_gcond_set_knob), -1 = follow the process default
  unsigned* edge_bits;                 // [C][cap1 * f2][h / 32] sign bits of the H1 row each hop-2 edge slot gathered (gs_ell_spmm_kernel): the
                                       // edge product reads its ReLU mask from these 32 B per edge instead of the 1-KiB row of H1full
  int h1_rows_max;
  float *H1full, *P0, *rs0, *Sr, *rsr, *Zr, *gw_real, *Tm, *Y1, *Bmat, *slabs_r, *colpart, *colpart_r;
  // synthetic side
  float *adj_raw, *adj_norm, *rnorm, *Q, *rsq, *H1s, *Ss, *Zs, *Ps, *Gs, *gw_syn, *stats, *ab, *loss_c, *D, *Ug, *Vs, *dSs,
      *drs, *dQ, *dAn, *d_adj, *d_feat, *dH2, *dHp, *A1m, *Bc, *dH1m, *slabs_x;
  // PGE
  float *U, *V, *mu1, *is1, *mu2, *is2, *sumU, *sumV, *A1, *E2, *e3, *de3, *dE2, *dY1, *rowsum_c, *colsum_r, *dU, *dV, *d_pge,
      *slabs_w, *W1T, *W1mT, *chunk_fwd, *chunk_bwd;   // chunked PGE: [chunk][mu1|is1|mu2|is2][Ph], [chunk][dw2|dbe2|dg2|dbe1|dg1][Ph]
  int pge_chunks;
  double* parts;
  // inner loop
  float *adj_norm_in, *rnorm_in, *Qin, *rsq_in, *S_in, *g_model;
  // distinct sources of the outermost block (gs_dedup_last_kernel), one set per block buffer: the edge product's K dimension (round 6)
  // (n0 / src0 / sptr0 / sedge0: distinct sources + CSC; nent / ent_src / ent_row / ent_w: the edge product's K entries — a source and up
  // to two of its in-edges as (row of Y, weight))
  struct { int *n0, *src0, *sptr0, *sedge0, *cur0, *nent, *ent_src; int2* ent_row; float2* ent_w; } dd[2];
  int* dd_scr;                       // scratch of the de-duplication's chain of launches (gs_dedup_scratch_ints)
  int cap0;                            // capacity per class = capL * f2 (0: the form is not available for this configuration)
  bool dd_valid[2];                    // dd[q] describes the blocks of buffer q (built behind a sampling / upload while the knob is on, else on demand)
  // the inner steps as one persistent launch (gs_inner.hip): workspace block, its carve, the host's copy of the arrival counter
  float* ip_ws; gs_inner_ws ipw; unsigned ip_counter; long ip_ws_floats;
  // optimiser state
  float *m_feat, *v_feat, *m_pge, *v_pge, *m_model, *v_model;
  int step_feat, step_pge, step_model;
  uint32_t sample_step;
  // bookkeeping (device)
  double* loss_sum;
  float* loss_last;
  long* steps;
  double* dist_part;                 // gs_dist_fused_kernel: [C][GS_DIST_SPLIT] shares of the class losses
  int dist_shares;                   // > 0: the last distance launch left this many loss shares per class for the V kernel to add
  int *tile_b, *tile_my;             // the 32-row M tiles that exist over the classes' synthetic rows (class, tile within the class)
  int tile_start[GS_MAX_CLASSES_TILES + 1];      // first entry of each class in that list (host)
  long *off_lo, *off_loF, *off_loC, *off_lon;
  int* ncls;
  float* ones;
  bool have_blocks;
  // shard of this rank: classes [c_lo, c_hi) (real side + per-class synthetic terms), PGE pair rows [r_lo, r_hi)
  int c_lo, c_hi, r_lo, r_hi;
  double* sums;     // [3][Ph] fp64 column sums exchanged between ranks
  double* sums2;    // [32][3][Ph] first-stage slices of the partial-block reduction
  float* statp;     // [ceil(n'^2 / 32)][2][Ph] per-tile column sums written by the E2 GEMM epilogue
  float *ar_syn, *ar_marg;
  // the real-side branch (sampling + class gradients) is independent of the synthetic-side branch until the
  // distance statistics: it runs on its own stream and joins through an event
  hipStream_t aux;
  hipEvent_t ev_fork, ev_join, ev_prefetch;   // ev_prefetch: the block set sampled ahead on aux is complete
  // hoisted training forward (round 5): `adj_syn = pge(feat_syn)` of the NEXT iteration (gcond.py:48) depends only on (X~, theta_PGE),
  // final once this iteration's optimiser has stepped — the outer step issues it on a third, lowest-priority stream beside the inner
  // SGC steps (latency-bound launches on a few CUs) instead of at the top of the next call; the next call waits for ev_hoist
  hipStream_t low;
  hipEvent_t ev_inf, ev_hoist;
  int hoist_fwd;                     // gs_gcond_set_hoist_forward: 0 (the library's default) = never, 1 = hoist, 2 = hoist + the 7/8 slab plan
  bool hoist_pending;                // a hoisted forward is (possibly) still in flight on `low`
  bool fwd_hoisted;                  // ... and is the training forward of the current (X~, theta_PGE)
  bool in_step;                      // inside gs_gcond_outer_step*: the stage entry points are being called by the step itself
  bool hoisting;                     // the forward being issued is the hoisted one (its long product leaves compute units to the inner steps)
  bool overlap, join_pending;
  bool overlap_sampling_only;        // level 2 of gs_gcond_set_overlap: the second stream only samples the NEXT iteration's blocks
  // PGE.inference at the end of iteration i and the training forward of iteration i+1 are the same function of the
  // same (X~, theta_PGE): with reuse_fwd the second one is not recomputed (its activations are still in place)
  bool reuse_fwd, fwd_cached;
  // blocks are double-buffered: the set for iteration i+1 is sampled on the aux stream while iteration i finishes
  gs_blocks blkbuf[2];
  int blk_cur;
  bool prefetched;
  // in-situ timing of the dominant kernel (PGE layer-2 GEMM): event pairs recorded around its launches while probing
  bool probing;
  int probe_n;
  unsigned char probe_pass[256];       // 1: the probed launch was the inference pass, 0: the training pass
  unsigned char probe_has_ev[256];     // the launch also carries a HIP-event bracket (every probe_stride-th launch of its pass)
  int probe_stride, probe_seen[2], probe2_seen;
  hipEvent_t probe_ev[2 * 256];
  int probe2_n;
  hipEvent_t probe2_ev[2 * 256];     // the same for the hop-2 gather SpMM of the real side (gs_ell_spmm_kernel), on ITS stream
  unsigned long long* probe_span;    // [256][2] device ticks (s_memrealtime): first workgroup start / last workgroup end of each probed launch
  bool gcn;         // condense model GCN (2 layers) instead of SGC
  float *P0g, *T2r, *dT2r, *dH1r, *S1, *dS1, *Tmat, *Rmat;   // GCN-only scratch
  bool identity;    // structure-free variants (GCondX / DosCondX): adj_syn = I, no PGE
  // operand precision of the three row-panel products (E2 = A1 W1^T, dY1 = dE2 W1, H1full = X W1): gs_gemm_desc.prec
  int operand_mode;
  float *bsplit_main, *bsplit_aux;   // pre-split B operands (bf16 hi | lo) of the products on the main / the real-side stream
  int bn2_slices, l3_slices;         // slices of e->sums2 the consumer of a deferred last reduction stage adds up (1: e->sums is final)
  // with_bn (SGC, ntrans = 2): theta = [W1, b1, W2, b2, bn.weight, bn.bias]; every stage reads the FOLDED parameters
  // theta_eff = [W1 diag(s), b1 s + bn.bias, W2, b2] (gs_bn_fold_kernel), refreshed whenever theta changes
  // full-neighbour mode (SGC, ntrans = 1; gs_gcond_set_full_neighbour): S_full = A^ A^ X and rs_full = A^ A^ 1 of the whole graph,
  // computed once by the caller (X never changes): the real-side class gradient needs their seed rows only
  const float* S_full; long ldS; const float* rs_full;
  bool with_bn;
  // dropout of the condensation model (models/sgc.py:42; SGC with two transforms): masks by gs_drop_mult4 (gs_kernels.h)
  unsigned drop_thr;                 // 0: no dropout; else round((1 - p) 65536)
  float drop_q;                      // 1 / (1 - p)
  unsigned long long drop_iter;      // matching iterations so far (gs_gcond_set_dropout_iter): part of every mask's key
  float* H1c;                        // [C][n'][h] the hidden activations of the synthetic graph under each class's mask
  long og, obeta, P_base;            // offsets of bn.weight / bn.bias inside theta; parameter count without them
  float *theta_eff, *bn_s2;          // [P], [h]: s_j^2 for the distance statistics of the W1 gradient columns
  float *gw_true_r, *gw_true_s;      // [C][P] each (with_bn under 'mse' / 'cos'): the class gradients mapped to the TRUE parameters
  float* a1_tab;                     // [Ph][8] layer-0 bias / BatchNorm-1 constants of the generated A1 operand and mask (gs_panel.hip)
  float* UsVs;                       // [2][n][Ph] folded layer-1 terms Us = s U + t, Vs = s V of the CURRENT forward (gs_pge_scale_uv): A1 = max(Us[c] + Vs[r], 0)
  bool uvs_valid;
  bool a1_valid;                     // A1 holds the activations of the current U, V (false: the forward formed them in registers only)
  float* gen_tab;                    // [Ph][8] folded BatchNorm-2 / layer-3 constants of the generated dE2 operand (gs_panel.hip)
  // block structure (gs_gcond_set_block_structure): the synthetic graph is `blk_B` disconnected blocks of `blk_n` rows each
  // (MSGC's batch of structures); blk_pos[b][i] = row of node i of block b.  The n'^3 products (A^ A^, dQ A^^T + A^^T dQ) then
  // run as blk_B small products on gathered blocks.  Caller-owned scratch: 3 blk_B blk_n^2 floats.
  const int* blk_pos; int blk_B, blk_n; float *blk_A, *blk_X, *blk_Y, *blk_F1, *blk_F2;
};

static void derive(gs_gcond* e, const gs_gcond_config& c) {
  e->cfg = c;
  e->N = c.n_nodes; e->d = c.d; e->ldx = c.ldx; e->C = c.nclass; e->n = c.n_syn; e->T = c.ntrans; e->h = c.hidden;
  e->Ph = c.pge_hidden; e->bcap = c.bcap; e->f1 = c.f1; e->f2 = c.f2; e->cap1 = c.bcap * (1 + c.f1);
  e->K = c.nlayers > 0 ? c.nlayers : 2;
  e->nmid = e->K > 2 ? e->K - 2 : 0;
  e->capL = e->cap1;
  for (int m = 0; m < GS_MAX_MID; ++m) {
    e->fmid[m] = m < e->nmid ? c.fmid[m] : 0;
    e->capm[m] = m < e->nmid ? e->capL * (1 + e->fmid[m]) : 0;
    if (m < e->nmid) e->capL = e->capm[m];
  }
  e->gcn = (c.condense_model == 1);
  e->pge_chunks = c.pge_chunks > 1 ? c.pge_chunks : 1;
  if (e->gcn) e->T = 2;                       // GCN: two GraphConvolution layers, same parameter layout as SGC T = 2
  e->deep = 0; e->gdeep = 0;
  if (!e->gcn && c.ntrans > 2) { e->deep = c.ntrans - 2; e->T = 2; }
  if (e->gcn && e->K > 2) e->gdeep = e->K - 2;
  e->F = (e->T == 1) ? c.d : c.hidden;
  e->Fp = (e->F + 3) & ~3;
  gs_pdesc& pd = e->pd;
  memset(&pd, 0, sizeof(pd));
  if (e->T == 1) {
    e->oW1 = e->ob1 = 0;
    e->oWl = 0; e->obl = (long)e->d * e->C;
    e->P = e->obl + e->C;
    pd.nparams = 2;
    pd.rows[0] = e->d; pd.cols[0] = e->C; pd.off[0] = 0; pd.is_bias[0] = 0;
    pd.rows[1] = e->C; pd.cols[1] = 1; pd.off[1] = e->obl; pd.is_bias[1] = 1;
  } else {
    e->oW1 = 0; e->ob1 = (long)e->d * e->h;
    long o = e->ob1 + e->h;
    int q = 2;
    pd.rows[0] = e->d; pd.cols[0] = e->h; pd.off[0] = 0; pd.is_bias[0] = 0;
    pd.rows[1] = e->h; pd.cols[1] = 1; pd.off[1] = e->ob1; pd.is_bias[1] = 1;
    for (int k = 0; k < e->deep + e->gdeep; ++k) {      // hidden-to-hidden transforms / GraphConvolution layers (nn.Module.parameters() order: layer by layer, weight then bias)
      e->oWm[k] = o; e->obm[k] = o + (long)e->h * e->h;
      pd.rows[q] = e->h; pd.cols[q] = e->h; pd.off[q] = e->oWm[k]; pd.is_bias[q] = 0; ++q;
      pd.rows[q] = e->h; pd.cols[q] = 1; pd.off[q] = e->obm[k]; pd.is_bias[q] = 1; ++q;
      o = e->obm[k] + e->h;
    }
    e->oWl = o; e->obl = e->oWl + (long)e->h * e->C;
    e->P = e->obl + e->C;
    pd.nparams = q + 2;
    pd.rows[q] = e->h; pd.cols[q] = e->C; pd.off[q] = e->oWl; pd.is_bias[q] = 0;
    pd.rows[q + 1] = e->C; pd.cols[q + 1] = 1; pd.off[q + 1] = e->obl; pd.is_bias[q + 1] = 1;
  }
  e->P_base = e->P;
  e->with_bn = c.with_bn != 0;
  // dropout acts behind the hidden transform only (models/sgc.py:38-43): with one transform the value has no effect, as upstream
  e->drop_thr = 0; e->drop_q = 1.f;
  if (c.dropout > 0.f && e->T == 2) {
    const double keep = 1.0 - (double)c.dropout;
    long thr = (long)(keep * 65536.0 + 0.5);
    e->drop_thr = (unsigned)(thr < 1 ? 1 : thr > 65535 ? 65535 : thr);
    e->drop_q = (float)(1.0 / keep);
  }
  e->gk = e->gcn && (e->gdeep > 0 || e->drop_thr != 0);
  e->og = e->obeta = 0;
  if (e->with_bn) {      // nn.Module.parameters() order: layers.*, then bns.0.weight, bns.0.bias (models/sgc.py:27-35)
    e->og = e->P; e->obeta = e->P + e->h;
    e->P += 2L * e->h;
    pd.rows[4] = e->h; pd.cols[4] = 1; pd.off[4] = e->og; pd.is_bias[4] = 1;        // 1-D: distance 0 under 'ours' (utils.py:98-101)
    pd.rows[5] = e->h; pd.cols[5] = 1; pd.off[5] = e->obeta; pd.is_bias[5] = 1;
    pd.nparams = 6;
  }
  int co = 0;
  for (int p = 0; p < pd.nparams; ++p) { pd.col_off[p] = co; co += pd.cols[p]; }
  pd.total_cols = co;
  const long Ph = e->Ph, d = e->d;
  e->pW0 = 0; e->pb0 = Ph * 2 * d; e->pW1 = e->pb0 + Ph; e->pb1 = e->pW1 + Ph * Ph; e->pw2 = e->pb1 + Ph;
  e->pb2 = e->pw2 + Ph; e->pg1 = e->pb2 + 1; e->pbe1 = e->pg1 + Ph; e->pg2 = e->pbe1 + Ph; e->pbe2 = e->pg2 + Ph;
  e->Ppge = e->pbe2 + Ph;
  // K-slab length of the real-side edge product (one workgroup per class, slab and 128-feature tile; 2 resident per CU = 512
  // slots).  1 024-edge slabs when they still fill the chip in ONE round (ogbn-arxiv: 40 classes x <= 14 slabs; in situ
  // -12 us fp32 / -33 us bf16x3 per iteration against 512, half the slab traffic), else 512 (few classes: keep the slots
  // busy); 768 and 1 408 measured worse at arxiv (a part-filled second round / idle slots): 706 / 758 us for the real-side stage
  // against 670 (512) and 660 (1 024).
  e->kchunk_r = (long)e->C * gs_cdiv((long)e->capL * e->f2, 1024) * gs_cdiv(e->d, 128) >= 400 ? 1024 : 512;
  if (e->nmid) e->kchunk_r = 1024 * (1 << (2 * e->nmid));      // (more hops: the edge lists grow ~16x per hop; the balanced plan carries them)
  e->nsplit_r = gs_cdiv((long)e->capL * e->f2, e->kchunk_r);
  // balanced slabs (round 5, gs_edge.hip): the 512 resident workgroup slots (2 per compute unit) per 128-feature tile, cut over the
  // classes in proportion to their live edge counts on the device; the buffer holds `edge_cap` slabs (the plan uses <= target + C)
  e->edge_target = 512 / gs_cdiv(e->d, 128) > 0 ? 512 / gs_cdiv(e->d, 128) : 1;
  if (e->deep && 512 / gs_cdiv(e->h, 128) < e->edge_target) e->edge_target = 512 / gs_cdiv(e->h, 128);      // (the hidden-to-hidden weights' products have h feature rows)
  e->edge_cap = e->C * e->nsplit_r > e->edge_target + e->C ? e->C * e->nsplit_r : e->edge_target + e->C;
  // dW1 = dE2^T A1 over K = n'^2 pair rows: Ph/128 x Ph/128 output tiles per slab; aim at one full round of the 768
  // resident 128x128 workgroups (2 048 gave 404 workgroups at n' = 454: half the chip idle)
  {
    const long tiles = (long)gs_cdiv(e->Ph, 128) * gs_cdiv(e->Ph, 128);
    const long slabs = 768 / (tiles > 0 ? tiles : 1) > 0 ? 768 / tiles : 1;
    long kc = gs_cdiv((long)e->n * e->n, slabs);
    kc = ((kc + 15) / 16) * 16;
    e->kchunk_w = (int)(kc < 256 ? 256 : kc);
  }
  e->nsplit_w = gs_cdiv((long)e->n * e->n, e->kchunk_w);
  e->rows_per_block = 64;
  e->nblk_stats = gs_cdiv((long)e->n * e->n, e->rows_per_block);
}

static size_t carve(gs_gcond* e, void* base) {
  Carver w{(char*)base, 0};
  const long C = e->C, n = e->n, d = e->d, h = e->h, Ph = e->Ph, F = e->F, P = e->P, N = e->N;
  const long bcap = e->bcap, cap1 = e->cap1, f1 = e->f1, f2 = e->f2, n2 = n * n, capL = e->capL;
  for (int q = 0; q < 2; ++q) {
    gs_blocks& b = e->blkbuf[q];
    memset(&b, 0, sizeof(b));
    b.C = e->C; b.bcap = e->bcap; b.f1 = e->f1; b.f2 = e->f2; b.cap1 = e->cap1;
    b.n_seed = w.take<int>(C); b.n1 = w.take<int>(C); b.nid1 = w.take<int>(C * cap1);
    b.col1 = w.take<int>(C * bcap * f1); b.val1 = w.take<float>(C * bcap * f1);
    b.tptr1 = w.take<int>(C * (cap1 + 1)); b.tedge1 = w.take<int>(C * bcap * f1);
    b.nmid = e->nmid;
    long capp = cap1;
    for (int m = 0; m < e->nmid; ++m) {      // middle hops (nlayers >= 3)
      const long fm = e->fmid[m], capc = e->capm[m];
      b.fm[m] = (int)fm; b.capm[m] = (int)capc;
      b.nm[m] = w.take<int>(C); b.nidm[m] = w.take<int>(C * capc);
      b.colm[m] = w.take<int>(C * capp * fm); b.valm[m] = w.take<float>(C * capp * fm);
      b.tptrm[m] = w.take<int>(C * (capc + 1)); b.tedgem[m] = w.take<int>(C * capp * fm); b.scrm[m] = w.take<int>(C * (capc + 1));
      capp = capc;
    }
    b.col0 = w.take<int>(C * capL * f2); b.val0 = w.take<float>(C * capL * f2);
  }
  e->blk = e->blkbuf[0];
  // distinct-source form of the edge product: SGC, two transforms, two hops, 256 hidden units (the dedicated edge kernel's shape)
  e->cap0 = (e->T == 2 && !e->gcn && e->nmid == 0 && h == 256 && !e->drop_thr && e->cfg.dis_metric == 0) ? (int)(capL * f2) : 0;
  const long dh = d > h ? d : h;      // rows of the widest weight whose per-class gradient is an edge product
  e->marker = (e->cfg.external_blocks && !e->cap0) ? nullptr : w.take<int>(C * N);
  e->kcount = w.take<int>(C);
  {
    // chunk counts of colsum_batched: <= 64 chunks of >= 8 rows, or rows/256 chunks
    long nch_r = gs_cdiv(capL * f2, 256) > 72 ? gs_cdiv(capL * f2, 256) : 72;
    long nch_s = gs_cdiv(n, 256) > 72 ? gs_cdiv(n, 256) : 72;
    e->colpart = w.take<float>(C * nch_s * (h > 0 ? h : 1));
    e->sum_part = w.take<double>(128);      // sum_f32's 128 partial sums (it used to borrow colpart, which tiny shapes undersize)
    e->colpart_r = w.take<float>(C * nch_r * (h > 0 ? h : 1));      // the real-side branch may run on its own stream
  }
  e->H1full = e->T == 2 ? w.take<float>(N * h) : nullptr;
  e->H1bits = (e->T == 2 && !e->gcn && h == 256) ? w.take<unsigned>(N * (h / 32)) : nullptr;
  e->edge_bits = (e->T == 2 && !e->gcn && h == 256 && !e->nmid) ? w.take<unsigned>(C * cap1 * e->f2 * (h / 32)) : nullptr;
  if (e->T == 2 && !e->gcn) {
    e->h1_flag = w.take<int>(N); e->h1_list = w.take<int>(N); e->h1_count = w.take<int>(1);
  } else {
    e->h1_flag = e->h1_list = e->h1_count = nullptr;
  }
  e->P0 = w.take<float>(C * capL * e->Fp); e->rs0 = w.take<float>(C * capL);
  {
    long capp = cap1;
    for (int m = 0; m < GS_MAX_MID; ++m) {
      e->Pmid[m] = e->rsmid[m] = e->Ymid[m] = nullptr;
      if (m >= e->nmid) continue;
      e->Pmid[m] = w.take<float>(C * capp * e->Fp); e->rsmid[m] = w.take<float>(C * capp);
      if (e->T == 2) e->Ymid[m] = w.take<float>(C * (long)e->capm[m] * h);
      capp = e->capm[m];
    }
  }
  e->Sr = w.take<float>(C * bcap * e->Fp); e->rsr = w.take<float>(C * bcap);
  e->Zr = w.take<float>(C * bcap * C); e->gw_real = w.take<float>(C * P);
  if (e->T == 2) {
    e->Tm = w.take<float>(C * bcap * h); e->Y1 = w.take<float>(C * cap1 * h);
    // (the materialised per-edge operand: only the metrics that need the bias gradient use it — with more than two hops it would be
    // tens of GB, so it exists there only when it is needed)
    e->Bmat = (e->nmid == 0 || e->cfg.dis_metric != 0 || e->drop_thr) ? w.take<float>(C * capL * f2 * h) : nullptr;
    e->slabs_r = w.take<float>((long)e->edge_cap * (e->deep ? dh : d) * h);
    e->edge_plan = w.take<int>(2 * C + 1);
  } else {
    e->Tm = e->Y1 = e->Bmat = e->slabs_r = nullptr;
    e->edge_plan = nullptr;
  }
  e->adj_raw = w.take<float>(n2); e->adj_norm = w.take<float>(n2); e->rnorm = w.take<float>(2 * n);
  e->Q = w.take<float>(n2); e->rsq = w.take<float>(n);
  e->Apow = e->K != 2 ? w.take<float>(n2 * (e->K + 1)) : nullptr;
  e->Apow_t = e->K != 2 ? w.take<float>(n2 * 2) : nullptr;
  e->H1s = e->T == 2 ? w.take<float>(n * h) : nullptr;
  e->Ss = w.take<float>(n * F); e->Zs = w.take<float>(n * C); e->Ps = w.take<float>(n * C); e->Gs = w.take<float>(n * C);
  e->gw_syn = w.take<float>(C * P); e->stats = w.take<float>(C * e->pd.total_cols * 3);
  e->ab = w.take<float>(C * e->pd.total_cols * 2); e->D = w.take<float>(C * P);
  e->Ug = w.take<float>(n * C); e->Vs = w.take<float>(n * C); e->dSs = w.take<float>(n * F); e->drs = w.take<float>(n);
  e->dQ = w.take<float>(n2); e->dAn = w.take<float>(n2);
  // one contiguous block [d_feat | d_adj | loss_c]: a single all-reduce when classes are sharded over ranks
  e->ar_syn = w.take<float>(n * d + n2 + C);
  e->d_feat = e->ar_syn; e->d_adj = e->ar_syn ? e->ar_syn + n * d : nullptr; e->loss_c = e->ar_syn ? e->ar_syn + n * d + n2 : nullptr;
  if (e->T == 2) {
    e->dH2 = w.take<float>(C * n * C); e->dHp = w.take<float>(C * n * h); e->A1m = w.take<float>(C * n * h);
    e->Bc = w.take<float>(C * n * C); e->dH1m = w.take<float>(n * h); e->slabs_x = w.take<float>(C * n * d);
  } else {
    e->dH2 = e->dHp = e->A1m = e->Bc = e->dH1m = e->slabs_x = nullptr;
  }
  e->H1c = e->drop_thr ? w.take<float>(C * n * h) : nullptr;
  e->U = w.take<float>(n * Ph); e->V = w.take<float>(n * Ph);
  e->mu1 = w.take<float>(Ph); e->is1 = w.take<float>(Ph); e->mu2 = w.take<float>(Ph); e->is2 = w.take<float>(Ph);
  e->sumU = w.take<float>(Ph); e->sumV = w.take<float>(Ph);
  e->A1 = w.take<float>(n2 * Ph); e->E2 = w.take<float>(n2 * Ph); e->e3 = w.take<float>(n2); e->de3 = w.take<float>(n2);
  e->dE2 = w.take<float>(n2 * Ph); e->dY1 = w.take<float>(n2 * Ph);
  e->ar_marg = w.take<float>(2 * n * Ph + Ph * Ph + Ph);     // [rowsum_c | colsum_r | dW1 | db1] contiguous (one all-reduce; the tail in PGE epochs)
  e->rowsum_c = e->ar_marg; e->colsum_r = e->ar_marg ? e->ar_marg + n * Ph : nullptr; e->dU = w.take<float>(n * Ph); e->W1T = w.take<float>(Ph * Ph); e->W1mT = w.take<float>(d * h);
  e->bsplit_main = w.take<float>(Ph * Ph * 3 / 2 + 16);             // pre-split B images: 4 B (modes 1, 2) or 6 B (mode 3) per element
  e->bsplit_aux = w.take<float>((h > 0 ? h : 1) * (e->deep ? dh : d) * 3 / 2 + 16);
  e->probe_span = w.take<unsigned long long>(2 * 256);
  e->theta_eff = e->with_bn ? w.take<float>(P) : nullptr;
  e->bn_s2 = e->with_bn ? w.take<float>(h) : nullptr;
  const bool bn_true = e->with_bn && e->cfg.dis_metric != 0;
  e->gw_true_r = bn_true ? w.take<float>(C * P) : nullptr;
  e->gw_true_s = bn_true ? w.take<float>(C * P) : nullptr;
  e->gen_tab = w.take<float>(Ph * 8);
  e->a1_tab = w.take<float>(Ph * 8);
  e->UsVs = w.take<float>(2 * n * Ph);
  e->uvs_valid = false;
  for (int i = 0; i < GS_KNOB_COUNT; ++i) e->knob[i] = -1;
  e->fold_finish = false;
  e->dist_shares = 0;
  e->bn2_done = false;
  e->l3_direct = false;
  e->skip_dfeat = false;
  e->a1_valid = false;
  e->bn2_slices = e->l3_slices = 1;
  e->chunk_fwd = w.take<float>((long)e->pge_chunks * 4 * Ph); e->chunk_bwd = w.take<float>((long)e->pge_chunks * 5 * Ph);
  e->dV = w.take<float>(n * Ph); e->d_pge = w.take<float>(e->Ppge); e->slabs_w = w.take<float>(e->nsplit_w * Ph * Ph);
  e->parts = w.take<double>((long)e->nblk_stats * 3 * Ph);
  e->sums = w.take<double>(3 * Ph);
  e->sums2 = w.take<double>(32 * 3 * Ph);
  e->statp = w.take<float>((long)gs_cdiv(n2, 32) * 2 * Ph);
  e->adj_norm_in = w.take<float>(n2); e->rnorm_in = w.take<float>(2 * n); e->Qin = w.take<float>(n2);
  e->rsq_in = w.take<float>(n); e->S_in = w.take<float>(n * F); e->g_model = w.take<float>(P);
  for (int q = 0; q < 2; ++q) {
    const long c0p = e->cap0;
    e->dd[q].n0 = c0p ? w.take<int>(C) : nullptr; e->dd[q].src0 = c0p ? w.take<int>(C * c0p) : nullptr;
    e->dd[q].sptr0 = c0p ? w.take<int>(C * (c0p + 1)) : nullptr; e->dd[q].sedge0 = c0p ? w.take<int>(C * c0p) : nullptr;
    e->dd[q].cur0 = c0p ? w.take<int>(C * (c0p + 1)) : nullptr;
    e->dd[q].nent = c0p ? w.take<int>(C) : nullptr; e->dd[q].ent_src = c0p ? w.take<int>(C * c0p) : nullptr;
    e->dd[q].ent_row = c0p ? w.take<int2>(C * c0p) : nullptr; e->dd[q].ent_w = c0p ? w.take<float2>(C * c0p) : nullptr;
  }
  e->dd_scr = e->cap0 ? w.take<int>(gs_dedup_scratch_ints((int)C, e->cap0)) : nullptr;
  for (int k = 0; k < 2; ++k) {
    e->Hmid_full[k] = k < e->deep ? w.take<float>(N * h) : nullptr;
    e->Hmid_s[k] = k < e->deep ? w.take<float>(n * h) : nullptr;
  }
  e->WmT = e->deep ? w.take<float>(h * h) : nullptr;
  const bool anyd = e->deep || e->gk;
  for (int k = 0; k < 2; ++k)
    if (k < e->gdeep) e->Hmid_s[k] = w.take<float>(n * h);
  for (int k = 0; k < 8; ++k) e->dp_cls[k] = (e->deep && k < 6) || e->gk ? w.take<float>(C * n * h) : nullptr;
  for (int k = 0; k < 6; ++k) e->dp_n[k] = anyd ? w.take<float>(n * h) : nullptr;
  for (int k = 0; k < 3; ++k) e->gk_sbuf[k] = k < e->gdeep + 1 && e->gk ? w.take<float>(n * h) : nullptr;
  e->gk_slabA = e->gk ? w.take<float>(C * n2) : nullptr;
  e->gr_P1 = e->gk ? w.take<float>(C * capL * d) : nullptr;
  {      // dropout: a_k per class (k = 2 .. L - 1), s_k per class (k = 2 .. L), shat_k per class (k = 2 .. L - 1) — gcnk_syn_match_drop
    const bool gd = e->gk && e->drop_thr;
    const int L = 2 + e->gdeep;
    for (int k = 0; k < 7; ++k) e->gkd[k] = nullptr;
    for (int k = 2; gd && k <= L - 1; ++k) e->gkd[k - 2] = w.take<float>(C * n * h);
    for (int k = 2; gd && k <= L; ++k) e->gkd[2 + k - 2] = w.take<float>(C * n * h);
    for (int k = 2; gd && k <= L - 1; ++k) e->gkd[5 + k - 2] = w.take<float>(C * n * h);
  }
  for (int k = 0; k < 4; ++k) e->gr_H[k] = nullptr;
  if (e->gk) {
    // layer k's output rows: the frontier it lands on (k = 1: the last middle frontier, capL rows; .. ; k = L - 1: hop 1's, cap1 rows)
    const int L = 2 + e->gdeep;
    for (int k = 1; k < L; ++k) {
      const int f = L - k;
      const long rows = f == 1 ? cap1 : e->capm[f - 2];
      e->gr_H[k] = w.take<float>(C * rows * h);
    }
    e->gr_T = w.take<float>(C * capL * h); e->gr_dT = w.take<float>(C * capL * h);
    e->gr_dHa = w.take<float>(C * capL * h); e->gr_dHb = w.take<float>(C * capL * h);
  } else {
    e->gr_T = e->gr_dT = e->gr_dHa = e->gr_dHb = nullptr;
  }
  e->dr_out = e->deep ? w.take<float>(C * dh * h) : nullptr;
  e->dr_a = e->deep ? w.take<float>(C * (long)e->cap0 * h) : nullptr;
  e->dr_b = e->deep ? w.take<float>(C * (long)e->cap0 * h) : nullptr;
  e->dr_c = e->deep ? w.take<float>(C * (long)e->cap0 * h) : nullptr;
  e->ones0 = e->deep ? w.take<float>(C * (long)e->cap0) : nullptr;
  e->dd_valid[0] = e->dd_valid[1] = false;
  e->ip_ws = nullptr; e->ip_ws_floats = 0; e->ip_counter = 0;
  if (e->T == 2 && !e->gcn && !e->with_bn && !e->deep && gs_inner_persist_eligible((int)n, (int)d, (int)h, (int)C, 1)) {
    e->ip_ws_floats = gs_inner_persist_ws_floats((int)n, (int)d, (int)h, (int)C);
    e->ip_ws = w.take<float>(e->ip_ws_floats);
    if (e->ip_ws) gs_inner_persist_carve(e->ip_ws, (int)n, (int)d, (int)h, (int)C, &e->ipw);
  }
  e->m_feat = w.take<float>(n * d); e->v_feat = w.take<float>(n * d);
  e->m_pge = w.take<float>(e->Ppge); e->v_pge = w.take<float>(e->Ppge);
  e->m_model = w.take<float>(P); e->v_model = w.take<float>(P);
  e->loss_sum = w.take<double>(1); e->loss_last = w.take<float>(1); e->steps = w.take<long>(1);
  e->off_lo = w.take<long>(C); e->off_loF = w.take<long>(C); e->off_loC = w.take<long>(C); e->off_lon = w.take<long>(C);
  e->ncls = w.take<int>(C); e->ones = w.take<float>(n > C * bcap ? n : C * bcap);
  e->dist_part = w.take<double>(C * GS_DIST_SPLIT);
  e->tile_b = w.take<int>(n / 32 + C + 1); e->tile_my = w.take<int>(n / 32 + C + 1);
  if (e->gcn) {
    e->P0g = w.take<float>(C * cap1 * d); e->T2r = w.take<float>(C * cap1 * C); e->dT2r = w.take<float>(C * cap1 * C);
    e->dH1r = w.take<float>(C * cap1 * h); e->S1 = w.take<float>(n * d); e->dS1 = w.take<float>(n * d);
    e->Tmat = w.take<float>(n * h); e->Rmat = w.take<float>(n * h);
  } else {
    e->P0g = e->T2r = e->dT2r = e->dH1r = e->S1 = e->dS1 = e->Tmat = e->Rmat = nullptr;
  }
  return (w.off + 255) & ~(size_t)255;
}

static int check_cfg(const gs_gcond_config* c) {
  if (!c) GS_FAIL(GS_ERR_ARG, "gs_gcond: null config");
  if (c->condense_model != 0 && c->condense_model != 1) GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond: condense_model must be 0 (SGC) or 1 (GCN)");
  if (c->condense_model == 0 && (c->ntrans < 1 || c->ntrans > 4))
    GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond: SGC ntrans must be 1 .. 4 (got %d)", c->ntrans);
  if (c->condense_model == 0 && c->ntrans > 2) {
    // models/sgc.py:32-35: ntrans - 2 hidden-to-hidden transforms.  Built on the two-transform kernels' shapes, for the shipped metric
    if (c->hidden != 256 || c->with_bn || c->dropout > 0.f || c->dis_metric != 0 || (c->nlayers != 0 && c->nlayers != 2) || c->d % 4 != 0)
      GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond: ntrans > 2 is built for hidden = 256, dis_metric 'ours', nlayers = 2, d %% 4 == 0, without "
                                  "BatchNorm / dropout");
  }
  if (c->condense_model == 1 && c->hidden < 1) GS_FAIL(GS_ERR_ARG, "gs_gcond: hidden must be > 0 for GCN");
  if (c->n_nodes < 1 || c->d < 1 || c->ldx < c->d || c->nclass < 1 || c->n_syn < 1 || c->pge_hidden < 1)
    GS_FAIL(GS_ERR_ARG, "gs_gcond: bad dimensions");
  if (c->nclass > 512) GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond: nclass > 512");
  if (c->pge_hidden % 4 != 0 || c->pge_hidden > 1024 || c->pge_hidden < 4)
    GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond: pge_hidden must be a multiple of 4 in [4, 1024] (the reference uses 128 / 256)");
  if (c->condense_model == 0 && c->ntrans >= 2 && c->hidden < 1) GS_FAIL(GS_ERR_ARG, "gs_gcond: hidden must be > 0 for ntrans = 2");
  if (c->bcap < 1 || c->bcap > 256 || c->f1 < 1 || c->f1 > 16 || c->f2 < 1 || c->f2 > 16)
    GS_FAIL(GS_ERR_ARG, "gs_gcond: bcap in [1,256], fan-outs in [1,16]");
  if (c->dis_metric < 0 || c->dis_metric > 2) GS_FAIL(GS_ERR_ARG, "DC error: unknown distance function");
  if (c->with_bn && (c->condense_model != 0 || c->ntrans != 2))
    GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond: with_bn is built for condense_model SGC with ntrans = 2 (models/sgc.py:27-33 puts BatchNorm "
                                "layers between transforms only; GCN + with_bn raises upstream, golden G22)");
  if (c->inner_loop < 0) GS_FAIL(GS_ERR_ARG, "gs_gcond: negative inner_loop");
  if (!(c->dropout >= 0.f && c->dropout < 1.f)) GS_FAIL(GS_ERR_ARG, "gs_gcond: dropout must be in [0, 1)");
  if (c->dropout > 0.f && !(c->condense_model == 0 && c->ntrans == 1)) {
    // models/sgc.py:38-43: F.dropout behind every transform but the last, in training mode inside train_class and the inner steps
    // (GCN: models/base.py:59, behind relu in every layer but the last — the layer loops of gs_gcnk.inc, golden G29)
    if (c->with_bn) GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond: dropout != 0 is built without BatchNorm");
    if (c->hidden % 4 != 0 || c->hidden > 1024) GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond: dropout != 0 needs hidden % 4 == 0, hidden <= 1024");
  }
  if (c->nlayers != 0 && c->nlayers != 2) {
    // models/sgc.py:45-51 propagates nlayers times, dataset/loader.py:205-210 samples as many hops; nlayers = 1 has no reference run
    // (PyG's sampler hands train_class the single hop's Adj tuple instead of a list: gcond_base.py:213 fails on it)
    if (c->nlayers < 3 || c->nlayers > 2 + GS_MAX_MID) GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond: nlayers must be 2 .. %d", 2 + GS_MAX_MID);
    if (c->condense_model != 0 && (c->nlayers > 4 || c->with_bn))
      GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond: a GCN condensation model takes nlayers 2 .. 4, without BatchNorm");
    for (int m = 0; m < c->nlayers - 2; ++m)
      if (c->fmid[m] < 1 || c->fmid[m] > 16) GS_FAIL(GS_ERR_ARG, "gs_gcond: middle fan-outs in [1, 16]");
  }
  return GS_OK;
}

extern "C" int gs_gcond_workspace_bytes(const gs_gcond_config* cfg, size_t* bytes) {
  GS_TRY(check_cfg(cfg));
  if (!bytes) GS_FAIL(GS_ERR_ARG, "gs_gcond_workspace_bytes: null out");
  gs_gcond tmp;
  derive(&tmp, *cfg);
  *bytes = carve(&tmp, nullptr);
  return GS_OK;
}

// Engine-level measurement knobs: a process-wide default (the gs_gcond_set_<knob>(int) setters of include/gslim_internal.h) that an
// engine follows unless it has its own value (gs_gcond_set_knob): two engines with different settings do not see each other.
static inline int knob_of(const gs_gcond* e, int id, int process_default) { return e->knob[id] >= 0 ? e->knob[id] : process_default; }
static int g_dw1_generated = 1;
static int g_fused_chains = 1;      // gs_gcond_set_fused_chains
static int g_defer_last_stage = 1;  // gs_gcond_set_folded_reductions
static int g_edge_bits = 0;         // the hop-2 gather also writes per-edge sign bits and the edge product reads those (0: the mask rows of H1full)
static int g_h1_bits = 0;
static int g_h1_rows = 1;
static int g_edge_plan = 1;         // real-side edge product on balanced slabs (gs_edge_plan); 0 = fixed-length slabs per class (the form of rounds 2-4)
static int g_edge_src = 0;          // real-side edge product over the DISTINCT sources of the outermost block (round 6: measured, not the default); 0 = over the edge slots
static int g_inner_persist = 1;     // the inner SGC steps (two transforms) as ONE persistent launch (gs_inner.hip); 0 = the launch chain of round 5
static int g_chained_products = 1;  // the per-class products of the synthetic match chained through the MFMA accumulator registers, two
                                    // launches (gs_match.hip); 0 = the six batched products
// measurement / test knob: 1 (default) = PGE epochs run the backward of layer 2 on generated operands (marginals as in feature epochs,
// dW1 | db1 by gs_dw1_gen_kernel); 0 = dE2 and A1 are materialised for a general tiled product (round 2's form)
extern "C" int gs_gcond_set_dw1_generated(int on) {
  g_dw1_generated = on != 0;
  return GS_OK;
}

static int g_aux_priority = -1;
extern "C" int gs_gcond_set_aux_priority(int level) {
  g_aux_priority = level;
  return GS_OK;
}

extern "C" int gs_gcond_create(const gs_gcond_config* cfg, const gs_gcond_io* io, gs_stream_t stream, gs_gcond** out) {
  GS_TRY(check_cfg(cfg));
  if (!io || !out) GS_FAIL(GS_ERR_ARG, "gs_gcond_create: null argument");
  if (!io->X || !io->labels_syn || !io->syn_lo || !io->syn_hi || !io->feat_syn || !io->pge_params || !io->model_params ||
      !io->adj_syn || !io->workspace)
    GS_FAIL(GS_ERR_ARG, "gs_gcond_create: null buffer in gs_gcond_io");
  if (!cfg->external_blocks && (!io->rowptr || !io->col || !io->val || !io->cls_ptr || !io->cls_nodes))
    GS_FAIL(GS_ERR_ARG, "gs_gcond_create: graph / class pools required unless external_blocks");
  gs_gcond* e = new (std::nothrow) gs_gcond;
  if (!e) GS_FAIL(GS_ERR_STATE, "gs_gcond_create: out of host memory");
  derive(e, *cfg);
  e->io = *io;
  size_t need = carve(e, nullptr);
  if (io->workspace_bytes < need) {
    delete e;
    GS_FAIL(GS_ERR_ARG, "gs_gcond_create: workspace too small (%zu < %zu)", io->workspace_bytes, need);
  }
  carve(e, io->workspace);
  hipStream_t s = (hipStream_t)stream;
  // Once, off the hot path: the fixed-capacity gather outputs keep their padding rows UNWRITTEN (gs_gcond_set_pad_zero) and
  // nothing reads them through a live index — but a buffer read (gs_gcond_buffer) or a caller's blocks whose padding slots point at
  // them must find finite values, whatever the caller's allocation held.
  (void)hipMemsetAsync(io->workspace, 0, need, s);
  const int C = e->C;
  int* lo = new int[C];
  int* hi = new int[C];
  hipError_t he = hipMemcpyAsync(lo, io->syn_lo, C * sizeof(int), hipMemcpyDeviceToHost, s);
  if (he == hipSuccess) he = hipMemcpyAsync(hi, io->syn_hi, C * sizeof(int), hipMemcpyDeviceToHost, s);
  if (he == hipSuccess) he = hipStreamSynchronize(s);
  if (he != hipSuccess) {
    delete[] lo; delete[] hi; delete e;
    GS_FAIL(GS_ERR_HIP, "gs_gcond_create: reading class ranges failed: %s", hipGetErrorString(he));
  }
  long *o1 = new long[C], *oF = new long[C], *oC = new long[C], *on = new long[C];
  int* nc = new int[C];
  e->max_nc = 0;
  bool bad = false;
  for (int c = 0; c < C; ++c) {
    if (lo[c] < 0 || hi[c] < lo[c] || hi[c] > e->n) bad = true;
    nc[c] = hi[c] - lo[c];
    if (nc[c] > e->max_nc) e->max_nc = nc[c];
    o1[c] = lo[c]; oF[c] = (long)lo[c] * e->F; oC[c] = (long)lo[c] * C; on[c] = (long)lo[c] * e->n;
  }
  float one = 1.f;
  if (!bad) {
    (void)hipMemcpyAsync(e->off_lo, o1, C * sizeof(long), hipMemcpyHostToDevice, s);
    (void)hipMemcpyAsync(e->off_loF, oF, C * sizeof(long), hipMemcpyHostToDevice, s);
    (void)hipMemcpyAsync(e->off_loC, oC, C * sizeof(long), hipMemcpyHostToDevice, s);
    (void)hipMemcpyAsync(e->off_lon, on, C * sizeof(long), hipMemcpyHostToDevice, s);
    (void)hipMemcpyAsync(e->ncls, nc, C * sizeof(int), hipMemcpyHostToDevice, s);
    {      // list of the M tiles that exist (gs_gemm_group2_tiles_f32): class-major
      const int cap = e->n / 32 + C + 1;
      int* tb = new int[cap];
      int* tm = new int[cap];
      int t = 0;
      e->tile_start[0] = 0;
      for (int c = 0; c < C; ++c) {
        if (C <= GS_MAX_CLASSES_TILES)
          for (int m = 0; m * 32 < nc[c] && t < cap; ++m) { tb[t] = c; tm[t] = m; ++t; }
        if (c < GS_MAX_CLASSES_TILES) e->tile_start[c + 1] = t;
      }
      e->tiles_ok = C <= GS_MAX_CLASSES_TILES;
      if (t > 0) {
        (void)hipMemcpyAsync(e->tile_b, tb, t * sizeof(int), hipMemcpyHostToDevice, s);
        (void)hipMemcpyAsync(e->tile_my, tm, t * sizeof(int), hipMemcpyHostToDevice, s);
        (void)hipStreamSynchronize(s);      // (the host arrays are freed below)
      }
      delete[] tb; delete[] tm;
    }
    hipLaunchKernelGGL(gs_fill_kernel, dim3(gs_cdiv(e->n > C * e->bcap ? e->n : C * e->bcap, 256)), dim3(256), 0, s, e->ones,
                       (long)(e->n > C * e->bcap ? e->n : C * e->bcap), 1.f);
    (void)one;
    (void)hipMemsetAsync(e->m_feat, 0, sizeof(float) * e->n * e->d, s);
    (void)hipMemsetAsync(e->v_feat, 0, sizeof(float) * e->n * e->d, s);
    (void)hipMemsetAsync(e->m_pge, 0, sizeof(float) * e->Ppge, s);
    (void)hipMemsetAsync(e->v_pge, 0, sizeof(float) * e->Ppge, s);
    (void)hipMemsetAsync(e->m_model, 0, sizeof(float) * e->P, s);
    (void)hipMemsetAsync(e->v_model, 0, sizeof(float) * e->P, s);
    (void)hipMemsetAsync(e->gw_real, 0, sizeof(float) * C * e->P, s);
    (void)hipMemsetAsync(e->gw_syn, 0, sizeof(float) * C * e->P, s);
    (void)hipMemsetAsync(e->loss_sum, 0, sizeof(double), s);
    (void)hipMemsetAsync(e->loss_last, 0, sizeof(float), s);
    (void)hipMemsetAsync(e->steps, 0, sizeof(long), s);
    // (the arrival counter / error word, and the pad columns of Q | Q^T the products never write)
    if (e->ip_ws) (void)hipMemsetAsync(e->ip_ws, 0, sizeof(float) * e->ip_ws_floats, s);
    if (e->ones0) hipLaunchKernelGGL(gs_fill_kernel, dim3(gs_cdiv((long)C * e->cap0, 256)), dim3(256), 0, s, e->ones0, (long)C * e->cap0, 1.f);
    if (e->marker)
      hipLaunchKernelGGL(gs_fill_int_kernel, dim3(gs_cdiv((long)C * e->N, 256)), dim3(256), 0, s, e->marker, (long)C * e->N,
                         0x7fffffff);
    he = hipStreamSynchronize(s);
  }
  delete[] lo; delete[] hi; delete[] o1; delete[] oF; delete[] oC; delete[] on; delete[] nc;
  if (bad) { delete e; GS_FAIL(GS_ERR_ARG, "gs_gcond_create: class row ranges out of bounds"); }
  if (he != hipSuccess) { delete e; GS_FAIL(GS_ERR_HIP, "gs_gcond_create: initialisation failed: %s", hipGetErrorString(he)); }
  e->step_feat = e->step_pge = e->step_model = 0;
  e->drop_iter = 0;
  e->sample_step = 0;
  e->have_blocks = false;
  e->c_lo = 0; e->c_hi = e->C; e->r_lo = 0; e->r_hi = e->n;
  e->aux = nullptr; e->ev_fork = e->ev_join = e->ev_prefetch = nullptr; e->join_pending = false;
  e->low = nullptr; e->ev_inf = e->ev_hoist = nullptr; e->hoist_pending = e->fwd_hoisted = e->in_step = e->hoisting = false; e->hoist_fwd = 0;
  e->reuse_fwd = false; e->fwd_cached = false; e->real_beside = false;
  e->blk_cur = 0; e->prefetched = false; e->identity = false;
  e->probing = false; e->probe_n = 0; e->probe2_n = 0;
  e->probe_stride = 1; e->probe_seen[0] = e->probe_seen[1] = e->probe2_seen = 0;
  for (int i = 0; i < 2 * 256; ++i) e->probe2_ev[i] = nullptr;
  e->operand_mode = 0;
  e->S_full = nullptr; e->ldS = 0; e->rs_full = nullptr;
  e->blk_pos = nullptr; e->blk_B = e->blk_n = 0; e->blk_A = e->blk_X = e->blk_Y = e->blk_F1 = e->blk_F2 = nullptr;
  for (int i = 0; i < 2 * 256; ++i) e->probe_ev[i] = nullptr;
  // the real-side branch is the longer of the two concurrent branches (kernel trace, r01): its stream gets the higher
  // priority so that its latency-bound kernels are not queued behind the PGE forward's big GEMM
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  const int prio = g_aux_priority < 0 ? prio_hi : (g_aux_priority > 0 ? prio_lo : 0);
  e->overlap = (hipStreamCreateWithPriority(&e->aux, hipStreamNonBlocking, prio) == hipSuccess) &&
               (hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) == hipSuccess) &&
               (hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) == hipSuccess) &&
               (hipEventCreateWithFlags(&e->ev_prefetch, hipEventDisableTiming) == hipSuccess);
  e->overlap_sampling_only = false;
  if (hipStreamCreateWithPriority(&e->low, hipStreamNonBlocking, prio_lo) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_inf, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_hoist, hipEventDisableTiming) != hipSuccess) {
    if (e->low) (void)hipStreamDestroy(e->low);
    e->low = nullptr;      // (no hoisting then)
  }
  *out = e;
  return GS_OK;
}

extern "C" void gs_gcond_destroy(gs_gcond* e) {
  if (!e) return;
  if (e->aux) { (void)hipStreamSynchronize(e->aux); (void)hipStreamDestroy(e->aux); }
  if (e->low) { (void)hipStreamSynchronize(e->low); (void)hipStreamDestroy(e->low); }
  if (e->ev_inf) (void)hipEventDestroy(e->ev_inf);
  if (e->ev_hoist) (void)hipEventDestroy(e->ev_hoist);
  for (int i = 0; i < 2 * 256; ++i) if (e->probe_ev[i]) (void)hipEventDestroy(e->probe_ev[i]);
  for (int i = 0; i < 2 * 256; ++i) if (e->probe2_ev[i]) (void)hipEventDestroy(e->probe2_ev[i]);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  if (e->ev_prefetch) (void)hipEventDestroy(e->ev_prefetch);
  delete e;
}

// A hoisted forward lives between two consecutive gs_gcond_outer_step* calls only: every other entry point that reads or writes what it
// touches waits for it on the host first and drops it (the next step then recomputes the forward at its top, as before).
static int quiesce_hoist(gs_gcond* e) {
  if (!e || e->in_step) return GS_OK;
  if (e->hoist_pending) { GS_HIP(hipEventSynchronize(e->ev_hoist)); e->hoist_pending = false; }
  e->fwd_hoisted = false;
  return GS_OK;
}
// Opt-in PER ENGINE (VERDICT r05 "next" 5): while a hoisted pass is in flight a kernel on the engine's third stream READS the caller-owned
// io.feat_syn / io.pge_params after gs_gcond_outer_step has returned — a contract a C caller must ask for (the library's default is
// off: between two default-configured steps nothing of the engine touches the caller's buffers); graphslim_amd.engine turns it on
// for every engine it creates and brackets its own writes with gs_gcond_invalidate_forward.
extern "C" int gs_gcond_set_hoist_forward(gs_gcond* e, int on) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  if (on < 0 || on > 2) GS_FAIL(GS_ERR_ARG, "gs_gcond_set_hoist_forward: on must be 0, 1 or 2 (2: with the edge product's 7/8 slab plan)");
  GS_TRY(quiesce_hoist(e));
  e->hoist_fwd = on;
  return GS_OK;
}
// process-wide measurement override (gslim_internal.h): -1 (default) = every engine follows its own setting, 0 / 1 / 2 = all engines
static int g_hoist_fwd = -1;
extern "C" int gs_gcond_set_hoist_forward_all(int on) {
  g_hoist_fwd = on < -1 ? -1 : (on > 2 ? 2 : on);
  return GS_OK;
}
static inline int hoist_mode(const gs_gcond* e) { return g_hoist_fwd >= 0 ? g_hoist_fwd : e->hoist_fwd; }

extern "C" int gs_gcond_set_forward_reuse(gs_gcond* e, int on) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  GS_TRY(quiesce_hoist(e));
  e->reuse_fwd = on != 0;
  e->fwd_cached = false;
  return GS_OK;
}

extern "C" int gs_gcond_set_dropout_iter(gs_gcond* e, unsigned long long iter) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  e->drop_iter = iter;
  return GS_OK;
}
extern "C" int gs_gcond_get_dropout_iter(gs_gcond* e, unsigned long long* iter) {
  if (!e || !iter) GS_FAIL(GS_ERR_ARG, "gs_gcond_get_dropout_iter: null argument");
  *iter = e->drop_iter;
  return GS_OK;
}

extern "C" int gs_gcond_invalidate_forward(gs_gcond* e) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  GS_TRY(quiesce_hoist(e));
  e->fwd_cached = false;
  return GS_OK;
}

// 0 = exact fp32 (default, the reference's precision), 1 = split bf16 x3, 2 = bf16 operands — for the three row-panel
// products only (gs_gemm_desc.prec); everything else stays fp32.  Switchable at any time (A/B measurements).
extern "C" int gs_gcond_set_operand_mode(gs_gcond* e, int mode) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  GS_TRY(quiesce_hoist(e));
  if (mode < 0 || mode > 3)
    GS_FAIL(GS_ERR_ARG, "gs_gcond_set_operand_mode: mode must be 0 (fp32), 1 (2-way bf16 split, 3 products), 2 (bf16) or 3 (exact 3-way bf16 split, 6 products)");
  e->operand_mode = mode;
  e->fwd_cached = false;
  return GS_OK;
}

extern "C" int gs_gcond_set_overlap(gs_gcond* e, int on) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  GS_TRY(quiesce_hoist(e));
  e->overlap = on && e->aux && e->ev_fork && e->ev_join && e->ev_prefetch;
  e->overlap_sampling_only = e->overlap && on == 2;
  return GS_OK;
}

// Full-neighbour mode (loader.py:204 `sizes = [-1, -1]`, commented out upstream; SURVEY section 8e "mode F"): with SGC and ONE
// transform the model is linear up to the softmax, so the 2-hop full-neighbourhood block product of a seed is row `seed` of
// S_full = A^ (A^ X) — a constant of the whole job, computed once (row-sharded over the ranks: graphslim_amd/parallel.py) —
// and its bias factor row `seed` of rs_full = A^ (A^ 1).  With the pointers set, sampling draws the seeds only and the
// real-side stage gathers their rows; NULL returns to neighbour sampling.  The arrays stay owned by the caller.
extern "C" int gs_gcond_set_full_neighbour(gs_gcond* e, const float* S_full, long ld, const float* rs_full) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  if (S_full && (e->gcn || e->T != 1))
    GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond_set_full_neighbour: built for SGC with ntrans = 1 (with a ReLU between two transforms the "
                                "backward needs one full-graph SpMM per class and iteration, DESIGN.md section 6)");
  if (S_full && (!rs_full || ld < e->d)) GS_FAIL(GS_ERR_ARG, "gs_gcond_set_full_neighbour: bad argument");
  e->S_full = S_full; e->ldS = ld; e->rs_full = S_full ? rs_full : nullptr;
  e->prefetched = false;           // a set sampled ahead belongs to the other mode
  return GS_OK;
}

// the parameters the stages compute with: theta itself, or its BatchNorm-folded image (with_bn)
static inline const float* model_th(const gs_gcond* e) { return e->with_bn ? e->theta_eff : e->io.model_params; }

// with_bn: theta_eff / s^2 from the current theta (after every upload and every Adam step on theta)
static int refresh_folded_model(gs_gcond* e, hipStream_t s) {
  if (!e->with_bn) return GS_OK;
  hipLaunchKernelGGL(gs_bn_fold_kernel, dim3(gs_cdiv(e->P_base, 256)), dim3(256), 0, s, e->io.model_params, e->d, e->h, e->oW1, e->ob1,
                     e->og, e->obeta, e->P_base, e->theta_eff, e->bn_s2);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_gcond_new_epoch(gs_gcond* e, gs_stream_t stream) {
  if (!e) GS_FAIL(GS_ERR_ARG, "gs_gcond_new_epoch: null engine");
  hipStream_t s = (hipStream_t)stream;
  GS_TRY(refresh_folded_model(e, s));          // (the caller has just written a fresh theta)
  GS_HIP(hipMemsetAsync(e->m_model, 0, sizeof(float) * e->P, s));
  GS_HIP(hipMemsetAsync(e->v_model, 0, sizeof(float) * e->P, s));
  e->step_model = 0;
  return GS_OK;
}

extern "C" int gs_gcond_get_blocks(gs_gcond* e, gs_blocks* out) {
  if (!e || !out) GS_FAIL(GS_ERR_ARG, "gs_gcond_get_blocks: null argument");
  *out = e->blk;
  return GS_OK;
}

static int dedup_into(gs_gcond* e, int which, gs_stream_t stream, bool force = false);
extern "C" int gs_gcond_set_blocks(gs_gcond* e, const gs_blocks* b, gs_stream_t stream) {
  if (!e || !b) GS_FAIL(GS_ERR_ARG, "gs_gcond_set_blocks: null argument");
  e->prefetched = false;
  const gs_blocks& d = e->blk;
  if (b->C != d.C || b->bcap != d.bcap || b->f1 != d.f1 || b->f2 != d.f2 || b->cap1 != d.cap1 || b->nmid != d.nmid)
    GS_FAIL(GS_ERR_ARG, "gs_gcond_set_blocks: block geometry differs from the engine's");
  for (int m = 0; m < d.nmid; ++m)
    if (b->fm[m] != d.fm[m] || b->capm[m] != d.capm[m]) GS_FAIL(GS_ERR_ARG, "gs_gcond_set_blocks: middle hop %d differs from the engine's", m);
  hipStream_t s = (hipStream_t)stream;
  const long C = d.C, bcap = d.bcap, f1 = d.f1, f2 = d.f2, cap1 = d.cap1;
#define CP(f, cnt, T) GS_HIP(hipMemcpyAsync(d.f, b->f, (cnt) * sizeof(T), hipMemcpyDeviceToDevice, s))
  CP(n_seed, C, int); CP(n1, C, int); CP(nid1, C * cap1, int); CP(col1, C * bcap * f1, int); CP(val1, C * bcap * f1, float);
  CP(tptr1, C * (cap1 + 1), int); CP(tedge1, C * bcap * f1, int);
  long capp = cap1;
  for (int m = 0; m < d.nmid; ++m) {
    const long fm = d.fm[m], capc = d.capm[m];
    CP(nm[m], C, int); CP(nidm[m], C * capc, int); CP(colm[m], C * capp * fm, int); CP(valm[m], C * capp * fm, float);
    CP(tptrm[m], C * (capc + 1), int); CP(tedgem[m], C * capp * fm, int);
    capp = capc;
  }
  CP(col0, C * capp * f2, int); CP(val0, C * capp * f2, float);
#undef CP
  e->have_blocks = true;
  return dedup_into(e, e->blk_cur, stream);
}

// sums[q][k] = sum over nblk partial blocks (fp64, fixed order): two stages so that the first one fills the chip
template <class T>
static int reduce_parts(gs_gcond* e, const T* part, int nblk, int nq, int nq_out, hipStream_t s, bool defer_last = false) {
  // Returns the number of slices left in e->sums2 ([slice][nq][Ph]) when `defer_last` asks for the second stage to be folded
  // into the consumer (gs_sum_slices, same order = same bits), else 1 with the result in e->sums.
  const int Ph = e->Ph;
  const int S = nblk > 64 ? 32 : 1;
  dim3 g1(gs_cdiv(Ph, 32), nq_out, S), g2(gs_cdiv(Ph, 32), nq_out, 1);
  double* out1 = S > 1 ? e->sums2 : e->sums;
  if (sizeof(T) == 8)
    hipLaunchKernelGGL(gs_sum_parts_kernel, g1, dim3(256), 0, s, (const double*)part, nblk, nq, Ph, out1);
  else
    hipLaunchKernelGGL(gs_sum_parts_f32_kernel, g1, dim3(256), 0, s, (const float*)part, nblk, nq, Ph, out1);
  // stage 1 writes [slice][nq_out... laid out with stride nq]: it used nq as the row stride of its OUTPUT too
  if (S > 1 && defer_last) return S;
  if (S > 1) hipLaunchKernelGGL(gs_sum_parts_kernel, g2, dim3(256), 0, s, (const double*)e->sums2, S, nq, Ph, e->sums);
  return 1;
}

// column sums of nbatch [rows x cols] matrices: chunked partials + fixed-order reduce (no atomics)
static int colsum_batched(gs_gcond* e, const float* M, long batch_stride, int rows, const int* rows_b, int cols, int nbatch,
                          float* out, long out_stride, gs_stream_t stream, float* scratch = nullptr) {
  int chunk = gs_cdiv(rows, 64);          // ~64 chunks per matrix: short serial loops, enough blocks
  chunk = chunk < 8 ? 8 : (chunk > 256 ? 256 : chunk);
  const int nchunk = gs_cdiv(rows, chunk);
  float* part = scratch ? scratch : e->colpart;
  hipLaunchKernelGGL(gs_colsum_chunks_kernel, dim3(nchunk, nbatch), dim3(256), 0, (hipStream_t)stream, M, batch_stride, rows,
                     rows_b, cols, chunk, part);
  GS_LAUNCH_CHECK();
  return gs_reduce_slabs_batched_f32(out, out_stride, part, cols, nchunk, cols, nbatch, 0.f, stream);
}

// ================================================================================================
// sharding over ranks (one process per GPU).  Classes [c_lo, c_hi) are this rank's share of the real side and of
// the per-class synthetic terms; pair rows [r_lo, r_hi) its share of the PGE grid.  Every quantity that crosses
// ranks is a SUM of per-rank partials laid out in a contiguous buffer (see gs_gcond_buffer "ar_*"), so the only
// collective is all-reduce(sum).  With the default full ranges the stages are the single-GPU path.
// ================================================================================================
extern "C" int gs_gcond_set_shard(gs_gcond* e, int c_lo, int c_hi, int r_lo, int r_hi) {
  if (e && e->deep && (c_lo != 0 || c_hi != e->C)) GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond_set_shard: ntrans > 2 is built unsharded");
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  GS_TRY(quiesce_hoist(e));
  if (c_lo < 0 || c_hi < c_lo || c_hi > e->C || r_lo < 0 || r_hi < r_lo || r_hi > e->n)
    GS_FAIL(GS_ERR_ARG, "gs_gcond_set_shard: ranges out of bounds");
  if (e->pge_chunks > 1 && (r_lo != 0 || r_hi != e->n))
    GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond_set_shard: the chunked PGE (pge_chunks > 1) cannot be combined with pair-row sharding");
  e->c_lo = c_lo; e->c_hi = c_hi; e->r_lo = r_lo; e->r_hi = r_hi;
  return GS_OK;
}

static inline bool sharded_c(const gs_gcond* e) { return e->c_lo != 0 || e->c_hi != e->C; }
static inline bool sharded_r(const gs_gcond* e) { return e->r_lo != 0 || e->r_hi != e->n; }

// measurement knob: 1 (default) = the last stage of the BN-2 / layer-3 statistics reductions is folded into its consumer
// launch (unsharded engines); 0 = separate launch.  Same bits either way.
extern "C" int gs_gcond_set_knob(gs_gcond* e, int knob, int value) {
  if (!e || knob < 0 || knob >= GS_KNOB_COUNT) GS_FAIL(GS_ERR_ARG, "gs_gcond_set_knob: null engine or unknown knob %d", knob);
  GS_TRY(quiesce_hoist(e));
  e->knob[knob] = value < 0 ? -1 : (value != 0);
  return GS_OK;
}
extern "C" int gs_gcond_get_knob(const gs_gcond* e, int knob) {
  if (!e || knob < 0 || knob >= GS_KNOB_COUNT) return -1;
  const int def[GS_KNOB_COUNT] = {g_fused_chains, g_h1_rows != 0, g_h1_bits, g_edge_bits, g_dw1_generated, g_defer_last_stage, g_chained_products, 0, g_edge_plan != 0, g_inner_persist, g_edge_src};
  return knob_of(e, knob, def[knob]);
}

extern "C" int gs_gcond_set_folded_reductions(int on) {
  g_defer_last_stage = on != 0;
  return GS_OK;
}

// gs_blocks view of the classes [c_lo, c_hi)
static gs_blocks blocks_view(const gs_gcond* e) {
  gs_blocks b = e->blk;
  const long c0 = e->c_lo, bcap = b.bcap, f1 = b.f1, f2 = b.f2, cap1 = b.cap1;
  b.C = e->c_hi - e->c_lo;
  b.n_seed += c0; b.n1 += c0; b.nid1 += c0 * cap1; b.col1 += c0 * bcap * f1; b.val1 += c0 * bcap * f1;
  b.tptr1 += c0 * (cap1 + 1); b.tedge1 += c0 * bcap * f1;
  long capp = cap1;
  for (int m = 0; m < b.nmid; ++m) {
    const long fm = b.fm[m], capc = b.capm[m];
    b.nm[m] += c0; b.nidm[m] += c0 * capc; b.colm[m] += c0 * capp * fm; b.valm[m] += c0 * capp * fm;
    b.tptrm[m] += c0 * (capc + 1); b.tedgem[m] += c0 * capp * fm; b.scrm[m] += c0 * (capc + 1);
    capp = capc;
  }
  b.col0 += c0 * capp * f2; b.val0 += c0 * capp * f2;
  return b;
}

int gs_sample_blocks_ex(const int*, const int*, const float*, int, const int*, const int*, int*, const gs_blocks*, uint64_t,
                        uint32_t, int, gs_stream_t, int);

// ================================================================================================
// stage: sampling
// ================================================================================================
// the distinct sources of the outermost block of block buffer `which` (this rank's classes): behind every sampling / upload of a set
static int dedup_into(gs_gcond* e, int which, gs_stream_t stream, bool force) {
  e->dd_valid[which] = false;
  // (the deep real side needs the per-source structure whatever the knob says)
  if (!e->cap0 || e->c_hi <= e->c_lo || e->S_full || !(force || e->deep || knob_of(e, GS_KNOB_EDGE_SRC, g_edge_src))) return GS_OK;
  e->dd_valid[which] = true;
  const gs_blocks keep = e->blk;
  e->blk = e->blkbuf[which];
  const gs_blocks b = blocks_view(e);
  e->blk = keep;
  const long c0 = e->c_lo, cap0 = e->cap0;
  return gs_dedup_last(b.col0, b.val0, b.n1, e->capL, e->f2, e->N, e->marker + c0 * e->N, e->c_hi - e->c_lo, e->dd[which].n0 + c0,
                       e->dd[which].src0 + c0 * cap0, e->dd[which].sptr0 + c0 * (cap0 + 1), e->dd[which].sedge0 + c0 * cap0,
                       e->dd[which].cur0 + c0 * (cap0 + 1), e->cap0, e->dd[which].nent + c0, e->dd[which].ent_src + c0 * cap0,
                       e->dd[which].ent_row + c0 * cap0, e->dd[which].ent_w + c0 * cap0, e->dd_scr, (hipStream_t)stream);
}

static int sample_into(gs_gcond* e, int which, gs_stream_t stream) {
  if (e->c_hi > e->c_lo) {
    const gs_blocks keep = e->blk;
    e->blk = e->blkbuf[which];
    gs_blocks b = blocks_view(e);
    e->blk = keep;
    GS_TRY(gs_sample_blocks_ex(e->io.rowptr, e->io.col, e->io.val, e->N, e->io.cls_ptr + e->c_lo, e->io.cls_nodes,
                               e->marker + (long)e->c_lo * e->N, &b, e->cfg.seed, e->sample_step, e->c_lo, stream,
                               e->S_full != nullptr));
    GS_TRY(dedup_into(e, which, stream));
  }
  e->sample_step++;
  return GS_OK;
}

extern "C" int gs_gcond_stage_sample(gs_gcond* e, gs_stream_t stream) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  if (e->cfg.external_blocks) {
    if (!e->have_blocks) GS_FAIL(GS_ERR_STATE, "gs_gcond: external_blocks set but gs_gcond_set_blocks was never called");
    return GS_OK;
  }
  if (e->prefetched) {          // the set for this iteration was sampled ahead on the aux stream: adopt it
    // whoever adopts it on another stream (a direct stage call, get_blocks, overlap switched off meanwhile) must see it complete
    if ((hipStream_t)stream != e->aux && e->ev_prefetch) GS_HIP(hipStreamWaitEvent((hipStream_t)stream, e->ev_prefetch, 0));
    e->blk_cur ^= 1;
    e->blk = e->blkbuf[e->blk_cur];
    e->prefetched = false;
  } else {
    GS_TRY(sample_into(e, e->blk_cur, stream));
  }
  e->have_blocks = true;
  return GS_OK;
}

// ================================================================================================
// stage: real-side per-class gradients gw_real[c] (gcond_base.py:212-224), propagate-first form
// ================================================================================================
// (ebits: per-edge sign bits of the gathered rows, see gs_ell_spmm_kernel; returns whether they were written — 16-byte path, F = 256)
static int g_pad_zero = 0;          // measurement knob: 1 = the gathers feeding another gather also zero their padding rows (rounds 1-4)
static int g_ell_xcd = 1;
extern "C" int gs_gcond_set_pad_zero(int on) {
  g_pad_zero = on != 0;
  return GS_OK;
}
extern "C" int gs_gcond_set_ell_xcd(int on) {
  g_ell_xcd = on != 0;
  return GS_OK;
}
static bool launch_ell(const int* col, const float* val, const int* nrows, int cap_rows, int f, long col_base_stride,
                       const float* src, long ld_src, const float* rs_in, int F, float* out, long ld_out, float* rs_out, int C,
                       hipStream_t s, unsigned* ebits = nullptr, unsigned long long dkey0 = 0, int dclass0 = 0, unsigned dthr = 0,
                       float dq = 1.f, int pad_zero = 1) {
  long waves = (long)C * cap_rows;
  int blocks = gs_cdiv(waves, 4);
  // class -> XCD affinity of the gather's workgroups (gs_ell_spmm_kernel; g_ell_xcd: measurement knob, default on)
  const int xcd = (g_ell_xcd && cap_rows % 4 == 0 && C > 1) ? 1 : 0;
  if (xcd) blocks = 8 * gs_cdiv(C, 8) * (cap_rows / 4);
  const int F4 = (F + 3) & ~3;
  bool vec = (ld_src % 4 == 0) && (ld_out % 4 == 0) && ld_src >= F4 && ld_out >= F4 && (((uintptr_t)src | (uintptr_t)out) % 16 == 0);
  const bool bits = vec && ebits && F == 256;
  if (dthr) {      // dropout on the gathered rows (the hop-2 gather of the hidden activations; 16-byte rows: checked at creation)
    if (!vec) return false;
    hipLaunchKernelGGL((gs_ell_spmm_kernel<4, true>), dim3(blocks), dim3(256), 0, s, col, val, nrows, cap_rows, f, col_base_stride,
                       src, ld_src, rs_in, F, out, ld_out, rs_out, C, (unsigned*)nullptr, xcd, dkey0, dclass0, dthr, dq, pad_zero);
    return false;
  }
  if (vec)
    hipLaunchKernelGGL((gs_ell_spmm_kernel<4>), dim3(blocks), dim3(256), 0, s, col, val, nrows, cap_rows, f, col_base_stride,
                       src, ld_src, rs_in, F, out, ld_out, rs_out, C, bits ? ebits : (unsigned*)nullptr, xcd, 0ull, 0, 0u, 1.f, pad_zero);
  else
    hipLaunchKernelGGL((gs_ell_spmm_kernel<1>), dim3(blocks), dim3(256), 0, s, col, val, nrows, cap_rows, f, col_base_stride,
                       src, ld_src, rs_in, F, out, ld_out, rs_out, C, (unsigned*)nullptr, xcd, 0ull, 0, 0u, 1.f, pad_zero);
  return bits;
}

// ---- the rows of H1 = relu(X W1 + b1) a rank needs: the hop-2 sources of ITS classes' blocks ---------------------------------
// flag[v] = 1 for every source id an edge slot of a live frontier row names.  Padding slots carry (0, 0.f) and the gather still
// multiplies row 0 by their zero weight: row 0 is listed with them, or a never-written (non-finite) row 0 would turn 0 * x into NaN
__global__ __launch_bounds__(256) void gs_h1_mark_kernel(const int* __restrict__ col0, const float* __restrict__ val0,
                                                         const int* __restrict__ n1, int cap1, int f2, int C, int* __restrict__ flag) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)C * cap1 * f2) return;
  const int c = (int)(e / ((long)cap1 * f2)), l = (int)((e / f2) % cap1);
  if (l < n1[c]) flag[col0[e]] = 1;
}

// list = the flagged ids, count = how many.  One block compacts 2 048 flags (8 per thread, block-wide exclusive scan) and takes
// its output range with ONE atomic (a wave-per-atomic form spent 32 us on 2 646 serialised atomics): the order of the list is
// arbitrary across blocks, the values computed from it do not depend on it.
__global__ __launch_bounds__(256) void gs_h1_compact_kernel(const int* __restrict__ flag, int N, int* __restrict__ list,
                                                            int* __restrict__ count) {
  __shared__ int s_w[4];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int v0 = (blockIdx.x * 256 + tid) * 8;
  int f[8], mine = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) { f[q] = (v0 + q < N) && flag[v0 + q] != 0; mine += f[q]; }
  int x = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
  if (lane == 63) s_w[w] = x;
  __syncthreads();
  if (tid == 0) {
    const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    s_base = tot ? atomicAdd(count, tot) : 0;
  }
  __syncthreads();
  int pos = s_base + x - mine;
  for (int i = 0; i < w; ++i) pos += s_w[i];
#pragma unroll
  for (int q = 0; q < 8; ++q) if (f[q]) list[pos++] = v0 + q;
}

// sum of n floats in two stages (fp64 partials of 128 blocks, then one wave): the one-block form took 53 us at n = n'^2 = 206 116
__global__ __launch_bounds__(256) void gs_sum_stage1_kernel(const float* __restrict__ x, long n, double* __restrict__ part) {
  double s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += x[i];
  s = gs_wave_sum_d(s);
  __shared__ double sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(64) void gs_sum_stage2_kernel(const double* __restrict__ part, int np, float* __restrict__ out) {
  double s = 0;
  for (int i = threadIdx.x; i < np; i += 64) s += part[i];
  s = gs_wave_sum_d(s);
  if (threadIdx.x == 0) out[0] = (float)s;
}
static int sum_f32(gs_gcond* e, const float* x, long n, float* out, hipStream_t s) {
  double* part = e->sum_part;
  hipLaunchKernelGGL(gs_sum_stage1_kernel, dim3(128), dim3(256), 0, s, x, n, part);
  hipLaunchKernelGGL(gs_sum_stage2_kernel, dim3(1), dim3(64), 0, s, (const double*)part, 128, out);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_gcond_set_edge_bits(int on) {
  g_edge_bits = on != 0;
  return GS_OK;
}
extern "C" int gs_gcond_set_edge_plan(int on) {
  g_edge_plan = on < 0 ? 0 : on;      // 0 off, 1 on (default number of slabs), > 1: on with that many slabs aimed at (measurement)
  return GS_OK;
}
// measurement / test knob: 1 = H1's product also writes its sign bits and the edge product reads its ReLU mask from them (32 B
// instead of 1 KiB per sampled edge); 0 (default) = the mask is read from the fp32 rows of H1full.  Same predicate, same bits out.
// Measured at the headline shape (round 3): the real side is 10-17 us SLOWER with the bits (the 128 ballots + 4-byte scattered
// stores per wave in H1's epilogue cost more than the mask rows, which the hop-2 gather has just pulled into L2, save).
extern "C" int gs_gcond_set_h1_bits(int on) {
  g_h1_bits = on != 0;
  return GS_OK;
}

// measurement knob: 1 (default) = a sharded rank computes only the H1 rows its blocks gather; 0 = every rank computes all N rows
extern "C" int gs_gcond_set_h1_rows(int on) {
  g_h1_rows = on < 0 ? 0 : on;      // (2: the row list also when the blocks name more slots than there are nodes — measurement)
  return GS_OK;
}

// out[c][s][:] = in[c][s][:] * [Hfull[src0[c][s]][:] > 0] for the live sources s < n0[c] (h = 256: one wave per row, 16 bytes per lane)
__global__ __launch_bounds__(256) void gs_gather_mask_rows_kernel(const float* __restrict__ in, const float* __restrict__ Hfull,
                                                                  const int* __restrict__ src0, const int* __restrict__ n0, int cap0,
                                                                  float* __restrict__ out, int nclass) {
  const int lane = threadIdx.x & 63;
  const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= (long)nclass * cap0) return;
  const int c = (int)(row / cap0), sidx = (int)(row - (long)c * cap0);
  if (sidx >= n0[c]) return;
  const float4 v = *reinterpret_cast<const float4*>(in + row * 256 + 4 * lane);
  const float4 m = *reinterpret_cast<const float4*>(Hfull + (long)src0[row] * 256 + 4 * lane);
  float4 o;
  o.x = m.x > 0.f ? v.x : 0.f; o.y = m.y > 0.f ? v.y : 0.f; o.z = m.z > 0.f ? v.z : 0.f; o.w = m.w > 0.f ? v.w : 0.f;
  *reinterpret_cast<float4*>(out + row * 256 + 4 * lane) = o;
}

// ntrans >= 3, real side backward (oracle/analytic.py real_class_grad_deep).  YL = the backward rows of the outermost block's TARGETS
// ((A1^T (G Wl^T))[row], h wide).  The gradient wrt the last hidden layer's output on a SOURCE is the sum over its in-edges (the
// outermost block's own transposed product over its CSC, gs_dedup_last_kernel); from there the layers are walked down per distinct
// source: gW_k[c] = sum_s a_{k-1}[src s] (x) ([a_k[src s] > 0] * cur[c][s]) is the edge product in its per-source form (unit weights, one
// Y row per K entry), and cur <- ([a_k[src] > 0] * cur) W_k^T is a batched product over the classes' live sources.
static int deep_real_backward(gs_gcond* e, const gs_blocks& b, const float* YL, int Cl, int c0, float* gw, int* kcount, gs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const int d = e->d, h = e->h, L = 2 + e->deep;
  const long P = e->P, cap0 = e->cap0;
  const float* th = e->io.model_params;
  if (!cap0 || !e->edge_plan) GS_FAIL(GS_ERR_STATE, "gs_gcond: ntrans > 2 needs the per-source structure (hidden = 256, 'ours')");
  if (!e->dd_valid[e->blk_cur]) {
    if (e->prefetched && e->ev_prefetch && s != e->aux) GS_HIP(hipStreamWaitEvent(s, e->ev_prefetch, 0));
    GS_TRY(dedup_into(e, e->blk_cur, stream, true));
  }
  const auto& dd = e->dd[e->blk_cur];
  const int* n0 = dd.n0 + c0;
  const int* src0 = dd.src0 + (long)c0 * cap0;
  float* cur = e->dr_a + (long)c0 * cap0 * h; float* dlt = e->dr_b + (long)c0 * cap0 * h; float* nxt = e->dr_c + (long)c0 * cap0 * h;
  // the gradient wrt a_{L-1} per distinct source: sum of its in-edges
  hipLaunchKernelGGL(gs_csc_spmm_t_v4_kernel, dim3(gs_cdiv((long)Cl * cap0, 4)), dim3(256), 0, s, dd.sptr0 + (long)c0 * (cap0 + 1),
                     dd.sedge0 + (long)c0 * cap0, b.val0, n0, (int)cap0, e->capL, e->f2, YL, cur, Cl);
  GS_LAUNCH_CHECK();
  for (int k = L - 1; k >= 1; --k) {      // hidden layer k: p_k = a_{k-1} W_k + b_k, a_k = relu(p_k)
    const float* Aprev = k == 1 ? e->io.X : (k == 2 ? e->H1full : e->Hmid_full[k - 3]);
    const long ldA = k == 1 ? e->ldx : h;
    const int M = k == 1 ? d : h;
    const float* Hk = k == 1 ? e->H1full : e->Hmid_full[k - 2];
    const long oW = k == 1 ? e->oW1 : e->oWm[k - 2];
    int target = 512 / gs_cdiv(M, 128);
    if (target > e->edge_cap - Cl) target = e->edge_cap - Cl;
    GS_TRY(gs_edge_plan(n0, 1, Cl, target < 1 ? 1 : target, kcount, e->edge_plan, s));      // kcount[c] = n0[c]
    gs_gemm_desc g = mk(Aprev, 1, ldA, cur, h, 1, e->slabs_r, h, M, h, (int)cap0);
    g.gatherA_k = src0; g.sGb1 = cap0;
    g.nb1 = Cl; g.nb2 = 1; g.kchunk = 16; g.Kb = kcount;
    g.sBb1 = cap0 * h; g.sCb1 = (long)M * h; g.sCb2 = (long)M * h;
    g.bgen_mask = Hk; g.bgen_ldm = h; g.bgen_idx = src0; g.bgen_w = e->ones0 + (long)c0 * cap0; g.bgen_div = 1;
    g.prec = e->operand_mode;
    if (!gs_edge_eligible(g)) GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond: ntrans > 2: the per-source edge product does not fit the edge kernel");
    GS_TRY(gs_edge_launch_plan(g, e->edge_plan, e->edge_cap, s));
    // (reduced into a contiguous, 16-byte aligned buffer, then copied into the rows of gw_real: P need not be a multiple of 4)
    GS_TRY(gs_reduce_slabs_plan(e->dr_out, (long)M * h, e->slabs_r, (long)M * h, (long)M * h, Cl, e->edge_plan, s));
    GS_HIP(hipMemcpy2DAsync(gw + oW, sizeof(float) * P, e->dr_out, sizeof(float) * (size_t)M * h, sizeof(float) * (size_t)M * h, Cl,
                            hipMemcpyDeviceToDevice, s));
    if (k > 1) {
      hipLaunchKernelGGL(gs_gather_mask_rows_kernel, dim3(gs_cdiv((long)Cl * cap0, 4)), dim3(256), 0, s, cur, Hk, src0, n0, (int)cap0, dlt, Cl);
      GS_LAUNCH_CHECK();
      gs_gemm_desc t = mk(dlt, h, 1, th + oW, 1, h, nxt, h, (int)cap0, h, h);      // (delta_k W_k^T)[s][i] = sum_j delta[s][j] W_k[i][j]
      t.nb1 = Cl; t.sAb1 = cap0 * h; t.sCb1 = cap0 * h; t.Mb = n0;
      GS_TRY(gs_gemm_f32(&t, stream));
      float* sw = cur; cur = nxt; nxt = sw;
    }
  }
  return GS_OK;
}

static int gcnk_real_grads(gs_gcond* e, const gs_blocks& b, int Cl, long c0, gs_stream_t stream);      // gs_gcnk.inc

extern "C" int gs_gcond_stage_real_grads(gs_gcond* e, gs_stream_t stream) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  if (!e->have_blocks) GS_FAIL(GS_ERR_STATE, "gs_gcond: no sampled blocks yet");
  hipStream_t s = (hipStream_t)stream;
  const int Cl = e->c_hi - e->c_lo, C = e->C, d = e->d, h = e->h, F = e->F, bcap = e->bcap, cap1 = e->cap1, f1 = e->f1, f2 = e->f2;
  if (Cl == 0) return GS_OK;
  const long P = e->P, c0 = e->c_lo;
  const gs_blocks b = blocks_view(e);
  // nlayers >= 3: the outermost block has one row per node of the LAST middle frontier (capL rows per class, nL[c] live)
  const int nmid = b.nmid, capL = e->capL;
  const int* nL = nmid ? b.nm[nmid - 1] : b.n1;
  if (nmid && (e->S_full || (e->gcn && !e->gdeep))) GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond: nlayers != 2 with the full-neighbour mode");
  if (e->gk) return gcnk_real_grads(e, b, Cl, c0, stream);
  // class-major scratch of this rank's classes
  const long Fp = e->Fp;
  float* P0 = e->P0 + c0 * capL * Fp; float* rs0 = e->rs0 + c0 * capL;
  float* Sr = e->Sr + c0 * bcap * Fp; float* rsr = e->rsr + c0 * bcap;
  float* Zr = e->Zr + c0 * bcap * C; float* gw = e->gw_real + c0 * P;
  const float* th = model_th(e);
  if (e->gcn) {
    // GCN (layers.py:36-51), propagate-first: P0 = A0 X_g ; H1 = relu(P0 W1 + b1) ; Z = A1 (H1 W2) + b2
    float* P0g = e->P0g + c0 * cap1 * d; float* H1r = e->Y1 + c0 * cap1 * h; float* T2r = e->T2r + c0 * cap1 * C;
    float* dT2 = e->dT2r + c0 * cap1 * C; float* dH1 = e->dH1r + c0 * cap1 * h;
    launch_ell(b.col0, b.val0, b.n1, cap1, f2, 0, e->io.X, e->ldx, nullptr, d, P0g, d, nullptr, Cl, s);
    GS_LAUNCH_CHECK();
    gs_gemm_desc g1 = mk(P0g, d, 1, th + e->oW1, h, 1, H1r, h, Cl * cap1, h, d);
    g1.bias = th + e->ob1; g1.relu = 1;
    GS_TRY(gs_gemm_f32(&g1, stream));
    gs_gemm_desc g2 = mk(H1r, h, 1, th + e->oWl, C, 1, T2r, C, Cl * cap1, C, h);
    GS_TRY(gs_gemm_f32(&g2, stream));
    launch_ell(b.col1, b.val1, b.n_seed, bcap, f1, cap1, T2r, C, nullptr, C, Zr, C, nullptr, Cl, s);
    hipLaunchKernelGGL(gs_real_softmax_grad_kernel, dim3(gs_cdiv((long)Cl * bcap, 4)), dim3(256), 0, s, Zr, e->ones, th + e->obl,
                       b.n_seed, bcap, C, (int)c0, Cl);                      // Z + 1 * b2 ; G in place
    GS_LAUNCH_CHECK();
    GS_TRY(colsum_batched(e, Zr, (long)bcap * C, bcap, nullptr, C, Cl, gw + e->obl, P, stream, e->colpart_r));   // gb2
    hipLaunchKernelGGL(gs_csc_spmm_t_kernel, dim3(gs_cdiv((long)Cl * cap1, 4)), dim3(256), 0, s, b.tptr1, b.tedge1, b.val1, b.n1,
                       cap1, bcap, f1, Zr, C, dT2, Cl);                      // dT2 = A1^T G   (rows >= n1 are zeros)
    GS_LAUNCH_CHECK();
    gs_gemm_desc w2 = mk(H1r, 1, h, dT2, C, 1, gw + e->oWl, C, h, C, cap1);   // gW2_c = H1_c^T dT2_c
    w2.nb1 = Cl; w2.sAb1 = (long)cap1 * h; w2.sBb1 = (long)cap1 * C; w2.sCb1 = P;
    GS_TRY(gs_gemm_f32(&w2, stream));
    gs_gemm_desc dh = mk(dT2, C, 1, th + e->oWl, 1, C, dH1, h, Cl * cap1, h, C);   // dH1p = (dT2 W2^T) * [H1 > 0]
    dh.mask = H1r; dh.ldmask = h;
    GS_TRY(gs_gemm_f32(&dh, stream));
    GS_TRY(colsum_batched(e, dH1, (long)cap1 * h, cap1, nullptr, h, Cl, gw + e->ob1, P, stream, e->colpart_r));   // gb1
    gs_gemm_desc w1 = mk(P0g, 1, d, dH1, h, 1, gw + e->oW1, h, d, h, cap1);   // gW1_c = P0_c^T dH1p_c
    w1.nb1 = Cl; w1.sAb1 = (long)cap1 * d; w1.sBb1 = (long)cap1 * h; w1.sCb1 = P;
    GS_TRY(gs_gemm_f32(&w1, stream));
    return GS_OK;
  }
  const float* src = e->io.X;
  long ld_src = e->ldx;
  bool use_bits = false, have_ebits = false;
  if (e->T == 2) {   // H1full = relu(X W1 + b1) for every node once: theta is shared by all classes
    // (W1 transposed first so that both operands are k-contiguous — see the dY1 product)
    hipLaunchKernelGGL(gs_transpose_small_kernel, dim3(gs_cdiv(h, 32), gs_cdiv(d, 32)), dim3(32, 8), 0, s, th + e->oW1, d, h, e->W1mT);
    GS_LAUNCH_CHECK();
    // Unsharded, the 40 classes' blocks touch (nearly) every node: all N rows in one product.  A sharded rank's few classes
    // gather fewer slots than there are nodes: their distinct sources are listed (mark + compact) and only those rows of H1full
    // are produced — same arithmetic per row, so every value downstream is bitwise the full product's.
    const long slots = (long)Cl * capL * f2;
    int rows_rc = GS_ERR_UNSUPPORTED;
    // (the sign bits are an output of the row-panel kernel: asked for only where the product takes that route)
    const bool want_bits = knob_of(e, GS_KNOB_H1_BITS, g_h1_bits) && e->H1bits && e->cfg.dis_metric == 0 && !e->drop_thr;
    if (knob_of(e, GS_KNOB_H1_ROWS, g_h1_rows) && !e->S_full && e->h1_flag && (slots < (long)e->N || g_h1_rows > 1)) {
      GS_HIP(hipMemsetAsync(e->h1_flag, 0, sizeof(int) * (size_t)e->N, s));
      GS_HIP(hipMemsetAsync(e->h1_count, 0, sizeof(int), s));
      hipLaunchKernelGGL(gs_h1_mark_kernel, dim3(gs_cdiv(slots, 256)), dim3(256), 0, s, b.col0, b.val0, nL, capL, f2, Cl, e->h1_flag);
      hipLaunchKernelGGL(gs_h1_compact_kernel, dim3(gs_cdiv(e->N, 2048)), dim3(256), 0, s, e->h1_flag, e->N, e->h1_list, e->h1_count);
      GS_LAUNCH_CHECK();
      rows_rc = gs_panel_rows(e->io.X, e->ldx, e->W1mT, d, th + e->ob1, 1, e->H1full, h, h, d, e->h1_list, e->h1_count,
                              (int)(slots < (long)e->N ? slots : (long)e->N),
                              (long)e->N, e->operand_mode, e->bsplit_aux, want_bits ? e->H1bits : nullptr, h / 32, s);
      if (rows_rc != GS_OK && rows_rc != GS_ERR_UNSUPPORTED) return rows_rc;
      use_bits = rows_rc == GS_OK && want_bits;
    }
    if (rows_rc != GS_OK) {
      gs_gemm_desc g = mk(e->io.X, e->ldx, 1, e->W1mT, 1, d, e->H1full, h, e->N, h, d);
      g.bias = th + e->ob1;
      g.relu = 1;
      g.prec = e->operand_mode; g.scratch = e->bsplit_aux;
      use_bits = want_bits && gs_gemm_route(&g) == 5;       // 5 = the row-panel kernel
      if (use_bits) { g.signbits = e->H1bits; g.ld_signbits = h / 32; }
      GS_TRY(gs_gemm_f32(&g, stream));
    }
    src = e->H1full;
    ld_src = h;
    // ntrans >= 3: the hidden-to-hidden transforms act row by row — every node's rows once per iteration, like H1full
    for (int k = 0; k < e->deep; ++k) {
      hipLaunchKernelGGL(gs_transpose_small_kernel, dim3(gs_cdiv(h, 32), gs_cdiv(h, 32)), dim3(32, 8), 0, s, th + e->oWm[k], h, h, e->WmT);
      GS_LAUNCH_CHECK();
      gs_gemm_desc g = mk(src, h, 1, e->WmT, 1, h, e->Hmid_full[k], h, e->N, h, h);
      g.bias = th + e->obm[k]; g.relu = 1;
      g.prec = e->operand_mode; g.scratch = e->bsplit_aux;
      GS_TRY(gs_gemm_f32(&g, stream));
      src = e->Hmid_full[k];
    }
  }
  if (e->S_full) {      // full-neighbour mode: S_c = S_full[seeds of class c], rs_c = rs_full[seeds]
    hipLaunchKernelGGL(gs_gather_seed_rows_kernel, dim3(gs_cdiv((long)Cl * bcap, 4)), dim3(256), 0, s, e->S_full, e->ldS, e->rs_full,
                       b.nid1, cap1, b.n_seed, bcap, F, Sr, Fp, rsr, Cl);
  } else {
  // hop 2 (outermost): frontier rows <- global sources;  hop 1: seeds <- frontier
  // (event brackets cost two stream markers per launch, ~8 us on this chain: they are recorded on every probe_stride-th launch)
  const bool probe2 = e->probing && e->probe2_n < 256 && (e->probe2_seen++ % e->probe_stride) == 0;
  if (probe2) {
    for (int q = 0; q < 2; ++q)
      if (!e->probe2_ev[2 * e->probe2_n + q]) GS_HIP(hipEventCreate(&e->probe2_ev[2 * e->probe2_n + q]));
    GS_HIP(hipEventRecord(e->probe2_ev[2 * e->probe2_n], s));
  }
  const bool want_ebits = knob_of(e, GS_KNOB_EDGE_BITS, g_edge_bits) && e->T == 2 && e->cfg.dis_metric == 0 && e->edge_bits && !use_bits &&
                          !e->drop_thr;
  // dropout (models/sgc.py:42): class c's batch draws its own mask over its n_id rows — here a function of (class, node id, column)
  const unsigned long long dkey_r = e->drop_thr ? gs_drop_key0(e->cfg.seed, e->drop_iter, 0) : 0ull;
  if (e->drop_thr && !(ld_src % 4 == 0 && Fp % 4 == 0 && ld_src >= ((F + 3) & ~3) && Fp >= ((F + 3) & ~3) &&
                       (((uintptr_t)src | (uintptr_t)P0) % 16 == 0)))      // (the masked gather exists in the 16-byte form only)
    GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond: dropout needs 16-byte rows in the hop-2 gather (hidden %% 4 == 0, aligned buffers)");
  have_ebits = launch_ell(b.col0, b.val0, nL, capL, f2, 0, src, ld_src, nullptr, F, P0, Fp, rs0, Cl, s,
                                     want_ebits ? e->edge_bits + c0 * cap1 * f2 * 8 : nullptr, dkey_r, (int)c0, e->drop_thr, e->drop_q,
                                     g_pad_zero);      // (P0's padding rows are never read: the next gather walks live indices)
  if (probe2) {
    GS_HIP(hipEventRecord(e->probe2_ev[2 * e->probe2_n + 1], s));
    e->probe2_n++;
  }
  // middle hops, outermost first: frontier m - 1 <- frontier m (rows of class c live in [c][cap(m-1)])
  const float* Pin = P0; const float* rsin = rs0;
  for (int m = nmid - 1; m >= 0; --m) {
    const int capp = m == 0 ? cap1 : b.capm[m - 1];
    float* Pm = e->Pmid[m] + c0 * capp * Fp; float* rsm = e->rsmid[m] + c0 * capp;
    launch_ell(b.colm[m], b.valm[m], m == 0 ? b.n1 : b.nm[m - 1], capp, b.fm[m], b.capm[m], Pin, Fp, rsin, F, Pm, Fp, rsm, Cl, s, nullptr, 0ull,
               0, 0u, 1.f, g_pad_zero);
    Pin = Pm; rsin = rsm;
  }
  launch_ell(b.col1, b.val1, b.n_seed, bcap, f1, cap1, Pin, Fp, rsin, F, Sr, Fp, rsr, Cl, s);
  }
  GS_LAUNCH_CHECK();
  {  // Z = S Wl  (all classes at once), then G in place
    gs_gemm_desc g = mk(Sr, Fp, 1, th + e->oWl, C, 1, Zr, C, Cl * bcap, C, F);
    GS_TRY(gs_gemm_f32(&g, stream));
    hipLaunchKernelGGL(gs_real_softmax_grad_kernel, dim3(gs_cdiv((long)Cl * bcap, 4)), dim3(256), 0, s, Zr, rsr, th + e->obl,
                       b.n_seed, bcap, C, (int)c0, Cl);
    GS_LAUNCH_CHECK();
  }
  {  // gWl_c = S_c^T G_c ;  gbl_c = rs_c^T G_c      (padding rows of G are exact zeros): same batch grid, one launch.
     // (Tm below has a 160x larger tile grid: grouping it in as well fills the launch with empty workgroups — measured
     // 341 instead of 360 outer-iters/s)
    gs_gemm_desc g2[2];
    g2[0] = mk(Sr, 1, Fp, Zr, C, 1, gw + e->oWl, C, F, C, bcap);
    g2[0].nb1 = Cl; g2[0].sAb1 = (long)bcap * Fp; g2[0].sBb1 = (long)bcap * C; g2[0].sCb1 = P;
    g2[1] = mk(rsr, 0, 1, Zr, C, 1, gw + e->obl, C, 1, C, bcap);
    g2[1].nb1 = Cl; g2[1].sAb1 = bcap; g2[1].sBb1 = (long)bcap * C; g2[1].sCb1 = P;
    GS_TRY(gs_gemm_group_f32(g2, 2, stream));
  }
  if (e->T == 2) {
    // Y1 = A1^T Tm ; Bmat = per-edge masked contribution ; gW1_c = X[col0]^T Bmat_c ; gb1_c = colsum
    const long K = (long)capL * f2;
    float* Tm = e->Tm + c0 * bcap * h; float* Y1 = e->Y1 + c0 * cap1 * h; float* Bmat = e->Bmat ? e->Bmat + c0 * K * h : nullptr;
    int* kcount = e->kcount + c0;
    gs_gemm_desc g = mk(Zr, C, 1, th + e->oWl, 1, C, Tm, h, Cl * bcap, h, C);      // Tm = G W2^T
    GS_TRY(gs_gemm_f32(&g, stream));
    if (h == 256)
      hipLaunchKernelGGL(gs_csc_spmm_t_v4_kernel, dim3(gs_cdiv((long)Cl * cap1, 4)), dim3(256), 0, s, b.tptr1, b.tedge1, b.val1, b.n1,
                         cap1, bcap, f1, Tm, Y1, Cl);
    else
      hipLaunchKernelGGL(gs_csc_spmm_t_kernel, dim3(gs_cdiv((long)Cl * cap1, 4)), dim3(256), 0, s, b.tptr1, b.tedge1, b.val1, b.n1,
                         cap1, bcap, f1, Tm, h, Y1, Cl);
    // middle hops: Y(m) = A(m)^T Y(m - 1) down to the rows of the outermost block (CSC of every middle block)
    const float* YL = Y1;
    for (int m = 0; m < nmid; ++m) {
      const int capp = m == 0 ? cap1 : b.capm[m - 1];
      float* Ym = e->Ymid[m] + c0 * (long)b.capm[m] * h;
      if (h == 256)
        hipLaunchKernelGGL(gs_csc_spmm_t_v4_kernel, dim3(gs_cdiv((long)Cl * b.capm[m], 4)), dim3(256), 0, s, b.tptrm[m], b.tedgem[m], b.valm[m],
                           b.nm[m], b.capm[m], capp, b.fm[m], YL, Ym, Cl);
      else
        hipLaunchKernelGGL(gs_csc_spmm_t_kernel, dim3(gs_cdiv((long)Cl * b.capm[m], 4)), dim3(256), 0, s, b.tptrm[m], b.tedgem[m], b.valm[m],
                           b.nm[m], b.capm[m], capp, b.fm[m], YL, h, Ym, Cl);
      YL = Ym;
    }
    if (e->deep) return deep_real_backward(e, b, YL, Cl, (int)c0, gw, kcount, stream);
    // 'ours' needs no bias gradient -> Bmat is generated inside the GEMM (dropout: the materialised form carries the mask)
    const bool fused = (e->cfg.dis_metric == 0) && !e->drop_thr;
    // balanced slabs (knob EDGE_PLAN, default on): the chip's workgroup slots are cut over the classes by their live edge counts
    const bool plan = fused && Cl <= 64 && knob_of(e, GS_KNOB_EDGE_PLAN, g_edge_plan != 0) && e->edge_plan && ((long)d * h) % 4 == 0 &&
                      P % 4 == 0 && e->oW1 % 4 == 0;
    if (plan) {
      // beside the training forward the match head of the main stream needs workgroup slots while the edge product runs: 7/8 of the
      // slots measured best (profiles/r05/stage_times_edge_sweep.txt: 384 .. 512 slabs -> 1 547 / 1 541 / 1 534 / 1 574 / 1 609 us per
      // iteration, fixed slabs 1 556); alone on the chip (forward reuse, staged calls) every slot
      int target = g_edge_plan > 1 ? g_edge_plan / gs_cdiv(d, 128) : (e->real_beside ? e->edge_target - e->edge_target / 8 : e->edge_target);
      if (target > e->edge_cap - Cl) target = e->edge_cap - Cl;
      GS_TRY(gs_edge_plan(nL, f2, Cl, target < 1 ? 1 : target, kcount, e->edge_plan, s));
    }
    else if (fused)
      hipLaunchKernelGGL(gs_kcount_kernel, dim3(gs_cdiv(Cl, 64)), dim3(64), 0, s, nL, f2, Cl, kcount);
    else
      hipLaunchKernelGGL(gs_make_bmat_kernel, dim3(gs_cdiv((long)Cl * capL * f2, 4)), dim3(256), 0, s, b.col0, b.val0, nL, capL,
                         f2, e->H1full, (long)h, YL, h, Bmat, kcount, Cl,
                         e->drop_thr ? gs_drop_key0(e->cfg.seed, e->drop_iter, 0) : 0ull, (int)c0, e->drop_thr, e->drop_q);
    GS_LAUNCH_CHECK();
    // K slabs: the engine's choice fills the chip with ALL classes (derive()); a class shard has fewer classes, so its slabs are
    // shortened until the launch again has ~512 workgroups (one workgroup = one class x slab x 128-feature tile is bound by ONE
    // compute unit's matrix rate: 5 classes in 1 024-edge slabs ran 194 us on 70 workgroups), within the slab scratch's size
    int kchunk = e->kchunk_r, nsplit = e->nsplit_r;
    if (Cl < C) {
      const long budget = (long)C * e->nsplit_r / Cl;                          // slabs per class that fit slabs_r
      long want = gs_cdiv(512, (long)Cl * gs_cdiv(d, 128));
      want = want > budget ? budget : want;
      if (want > nsplit) {
        long kc = gs_cdiv(K, want);
        kc = ((kc + 15) / 16) * 16;
        kchunk = (int)(kc < 64 ? 64 : kc);
        nsplit = gs_cdiv(K, kchunk);
        if ((long)nsplit > budget) { kchunk = e->kchunk_r; nsplit = e->nsplit_r; }
      }
    }
    gs_gemm_desc gwd = mk(e->io.X, 1, e->ldx, Bmat, h, 1, e->slabs_r, h, d, h, (int)K);
    gwd.gatherA_k = b.col0; gwd.sGb1 = K;
    gwd.nb1 = Cl; gwd.nb2 = nsplit; gwd.kchunk = kchunk; gwd.Kb = kcount;
    gwd.sBb1 = K * h; gwd.sCb1 = (long)nsplit * d * h; gwd.sCb2 = (long)d * h;
    if (fused) {   // B(k, :) = val0[k] * [H1full[col0[k]] > 0] * Y1[k / f2]
      gwd.B = YL; gwd.sBb1 = (long)capL * h;
      gwd.bgen_mask = e->H1full; gwd.bgen_ldm = h; gwd.bgen_idx = b.col0; gwd.bgen_w = b.val0; gwd.bgen_div = f2;
      if (use_bits) { gwd.bgen_bits = e->H1bits; gwd.bgen_ldbits = h / 32; }
      else if (have_ebits) { gwd.bgen_bits = e->edge_bits + c0 * cap1 * f2 * 8; gwd.bgen_ldbits = h / 32; gwd.bgen_bits_edge = 1; }
      gwd.prec = e->operand_mode;      // honoured by the dedicated edge kernel (gs_edge.hip); the tiled fallback is fp32
    }
    // Distinct-source form (knob EDGE_SRC, default OFF: measured net-neutral): the K dimension runs over ENTRIES = (distinct source, up to two of its in-edges)
    // built behind every sampling (gs_dedup_last: a chain of chip-wide launches): x and the mask row of a source are fetched once per entry and its in-edges
    // are summed in the operand path — the same sum in another association, K ~ 0.7 of the edge slots at ogbn-arxiv (8 700 live slots,
    // 5 400 distinct sources, ~6 100 entries per class).  (The first form materialised the per-source sums: the edge kernel 168 -> 126 us
    // but +140 us for that gather: profiles/r06/edge_src_materialised_not_kept.txt.)
    if (plan && e->cap0 && knob_of(e, GS_KNOB_EDGE_SRC, g_edge_src) && !have_ebits && nmid == 0 && gs_edge_eligible(gwd)) {
      const long cap0 = e->cap0;
      if (!e->dd_valid[e->blk_cur]) {      // (the knob was switched on after these blocks were sampled: build the entries now — the
        // marker rows are shared with a sampling possibly in flight on the second stream: behind it)
        if (e->prefetched && e->ev_prefetch && s != e->aux) GS_HIP(hipStreamWaitEvent(s, e->ev_prefetch, 0));
        GS_TRY(dedup_into(e, e->blk_cur, stream));
      }
      const auto& dd = e->dd[e->blk_cur];
      int target = g_edge_plan > 1 ? g_edge_plan / gs_cdiv(d, 128) : (e->real_beside ? e->edge_target - e->edge_target / 8 : e->edge_target);
      if (target > e->edge_cap - Cl) target = e->edge_cap - Cl;
      GS_TRY(gs_edge_plan(dd.nent + c0, 1, Cl, target < 1 ? 1 : target, kcount, e->edge_plan, s));      // (re-plans over the entries: kcount[c] = nent[c])
      gs_gemm_desc gs2 = gwd;
      gs2.K = (int)cap0;
      gs2.gatherA_k = dd.ent_src + c0 * cap0; gs2.sGb1 = cap0;
      gs2.bgen_idx = gs2.gatherA_k; gs2.bgen_div = 1;
      GS_TRY(gs_edge_launch_entries(gs2, dd.ent_row + c0 * cap0, dd.ent_w + c0 * cap0, e->edge_plan, e->edge_cap, s));
      GS_TRY(gs_reduce_slabs_plan(gw + e->oW1, P, e->slabs_r, (long)d * h, (long)d * h, Cl, e->edge_plan, s));
      return GS_OK;
    }
    if (plan && gs_edge_eligible(gwd)) {
      GS_TRY(gs_edge_launch_plan(gwd, e->edge_plan, e->edge_cap, s));
      GS_TRY(gs_reduce_slabs_plan(gw + e->oW1, P, e->slabs_r, (long)d * h, (long)d * h, Cl, e->edge_plan, s));
      return GS_OK;
    }
    GS_TRY(gs_gemm_f32(&gwd, stream));
    if (((long)d * h) % 4 == 0 && P % 4 == 0 && e->oW1 % 4 == 0)   // 16-byte form that skips the slabs beyond each class's edge count
      GS_TRY(gs_reduce_slabs_active(gw + e->oW1, P, e->slabs_r, (long)d * h, nsplit, (long)d * h, Cl, kcount, kchunk, stream));
    else
      GS_TRY(gs_reduce_slabs_batched_f32(gw + e->oW1, P, e->slabs_r, (long)d * h, nsplit, (long)d * h, Cl, 0.f, stream));
    if (!fused) GS_TRY(colsum_batched(e, Bmat, K * h, (int)K, kcount, h, Cl, gw + e->ob1, P, stream, e->colpart_r));
  }
  return GS_OK;
}

// ================================================================================================
// stage: PGE forward (parametrized_adj.py:40-71) + dense_gcn_norm, in three parts so that ranks can exchange
// (1 -> 2) the BN-2 column sums "ar_bn2" and (2 -> 3) the pre-sigmoid pair scores "e3".
// ================================================================================================
// ---- chunked PGE (parametrized_adj.py:41-55): np.array_split of the n'^2 flat pair indices into `pge_chunks` contiguous
// chunks, each normalised by its own BatchNorm batch statistics -------------------------------------------------------
static void chunk_range(const gs_gcond* e, int ch, long* k0, long* count) {
  const long n2 = (long)e->n * e->n, q = n2 / e->pge_chunks, rem = n2 % e->pge_chunks;   // the first `rem` chunks get q + 1
  *count = q + (ch < rem ? 1 : 0);
  *k0 = (long)ch * q + (ch < rem ? ch : rem);
}

static int pge_forward_chunked(gs_gcond* e, gs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const int n = e->n, d = e->d, Ph = e->Ph;
  const float* X = e->io.feat_syn;
  const float* pp = e->io.pge_params;
  e->fwd_cached = false;
  {
    gs_gemm_desc uv[2];
    uv[0] = mk(X, d, 1, pp + e->pW0, 1, 2L * d, e->U, Ph, n, Ph, d);
    uv[1] = uv[0]; uv[1].B = pp + e->pW0 + d; uv[1].C = e->V;
    GS_TRY(gs_gemm_group_f32(uv, 2, stream));
  }
  for (int ch = 0; ch < e->pge_chunks; ++ch) {
    long k0, cnt;
    chunk_range(e, ch, &k0, &cnt);
    float* mu1 = e->chunk_fwd + ((long)ch * 4 + 0) * Ph; float* is1 = mu1 + Ph; float* mu2 = is1 + Ph; float* is2 = mu2 + Ph;
    const int nblk = gs_cdiv(cnt, e->rows_per_block);
    hipLaunchKernelGGL(gs_pge_chunk_e1_stats_kernel, dim3(nblk), dim3(256), 0, s, e->U, e->V, pp + e->pb0, n, Ph, k0, cnt,
                       e->rows_per_block, e->parts);
    reduce_parts(e, e->parts, nblk, 2, 2, s);
    hipLaunchKernelGGL(gs_bn2_from_sums_kernel, dim3(gs_cdiv(Ph, 64)), dim3(64), 0, s, e->sums, 1, cnt, Ph, mu1, is1);
    hipLaunchKernelGGL(gs_pge_make_a1_flat_kernel, dim3(gs_cdiv(cnt * (Ph / 4), 256)), dim3(256), 0, s, e->U, e->V, pp + e->pb0, mu1,
                       is1, pp + e->pg1, pp + e->pbe1, n, Ph, k0, cnt, e->A1 + k0 * Ph);
    GS_LAUNCH_CHECK();
    gs_gemm_desc g = mk(e->A1 + k0 * Ph, Ph, 1, pp + e->pW1, 1, Ph, e->E2 + k0 * Ph, Ph, (int)cnt, Ph, Ph);
    g.bias = pp + e->pb1;
    g.colstat = e->statp;
    g.prec = e->operand_mode; g.scratch = e->bsplit_main;
    GS_TRY(gs_gemm_f32(&g, stream));
    reduce_parts(e, e->statp, gs_gemm_tiles_m(&g), 2, 2, s);
    hipLaunchKernelGGL(gs_bn2_from_sums_kernel, dim3(gs_cdiv(Ph, 64)), dim3(64), 0, s, e->sums, 1, cnt, Ph, mu2, is2);
    hipLaunchKernelGGL(gs_pge_layer3_kernel, dim3(gs_cdiv(cnt, 4)), dim3(256), 0, s, e->E2 + k0 * Ph, mu2, is2, pp + e->pg2,
                       pp + e->pbe2, pp + e->pw2, pp + e->pb2, cnt, Ph, e->e3 + k0);
    GS_LAUNCH_CHECK();
  }
  return GS_OK;
}

// parts 1 + 2 of the backward for the chunked PGE; leaves dU, dV (the sums of dE1 over r and over c) for part 3
static int pge_backward_chunked(gs_gcond* e, int need_param_grads, gs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const int n = e->n, Ph = e->Ph, nc = e->pge_chunks;
  const long n2 = (long)n * n;
  const float* pp = e->io.pge_params;
  float* g = e->d_pge;
  hipLaunchKernelGGL(gs_pge_adj_bwd_kernel, dim3(gs_cdiv(n2, 256)), dim3(256), 0, s, e->d_adj, e->adj_raw, n, e->de3);
  for (int ch = 0; ch < nc; ++ch) {       // BN-2 / layer-3 backward, chunk by chunk
    long k0, cnt;
    chunk_range(e, ch, &k0, &cnt);
    const float* mu2 = e->chunk_fwd + ((long)ch * 4 + 2) * Ph; const float* is2 = mu2 + Ph;
    float* bw = e->chunk_bwd + (long)ch * 5 * Ph;     // dw2 | dbe2 | dg2 | dbe1 | dg1
    const int nblk = gs_cdiv(cnt, e->rows_per_block);
    hipLaunchKernelGGL(gs_pge_l3_bwd_stats_kernel, dim3(nblk), dim3(256), 0, s, e->E2 + k0 * Ph, e->de3 + k0, mu2, is2, pp + e->pg2,
                       pp + e->pbe2, pp + e->pw2, cnt, Ph, e->rows_per_block, e->parts);
    reduce_parts(e, e->parts, nblk, 3, 3, s);
    hipLaunchKernelGGL(gs_sums_to_float_kernel, dim3(gs_cdiv(Ph, 64)), dim3(64), 0, s, e->sums, 3, Ph, bw, bw + Ph, bw + 2 * Ph);
    hipLaunchKernelGGL(gs_pge_bn2_bwd_v4_kernel, dim3(gs_cdiv(gs_cdiv(cnt, 8) * (long)(Ph / 4), 256)), dim3(256), 0, s, e->E2 + k0 * Ph,
                       e->de3 + k0, mu2, is2, pp + e->pg2, pp + e->pbe2, pp + e->pw2, bw + Ph, bw + 2 * Ph, cnt, cnt, Ph,
                       e->dE2 + k0 * Ph);
    GS_LAUNCH_CHECK();
  }
  if (need_param_grads) {
    // the parameters are shared by the chunks: their gradients are sums over chunks
    GS_TRY(gs_reduce_slabs_f32(g + e->pw2, e->chunk_bwd, Ph, nc, 5L * Ph, 0.f, stream));
    GS_TRY(gs_reduce_slabs_f32(g + e->pbe2, e->chunk_bwd + Ph, Ph, nc, 5L * Ph, 0.f, stream));
    GS_TRY(gs_reduce_slabs_f32(g + e->pg2, e->chunk_bwd + 2 * Ph, Ph, nc, 5L * Ph, 0.f, stream));
    GS_TRY(sum_f32(e, e->de3, n2, g + e->pb2, s));
    const int nblk = gs_cdiv(n2, e->rows_per_block);
    const int nsplit = gs_cdiv(n2, e->kchunk_w);
    gs_gemm_desc w = mk(e->dE2, 1, Ph, e->A1, Ph, 1, e->slabs_w, Ph, Ph, Ph, (int)n2);
    w.nb2 = nsplit; w.kchunk = e->kchunk_w; w.sCb2 = (long)Ph * Ph;
    GS_TRY(gs_gemm_f32(&w, stream));
    GS_TRY(gs_reduce_slabs_f32(g + e->pW1, e->slabs_w, (long)Ph * Ph, nsplit, (long)Ph * Ph, 0.f, stream));
    hipLaunchKernelGGL(gs_col_stats_kernel, dim3(nblk), dim3(256), 0, s, e->dE2, n2, Ph, e->rows_per_block, e->parts);
    reduce_parts(e, e->parts, nblk, 2, 1, s);
    hipLaunchKernelGGL(gs_sums_to_float_kernel, dim3(gs_cdiv(Ph, 64)), dim3(64), 0, s, e->sums, 1, Ph, g + e->pb1, (float*)nullptr,
                       (float*)nullptr);
    GS_LAUNCH_CHECK();
  }
  {  // dY1 = (dE2 W1) * [A1 > 0] over all pairs
    hipLaunchKernelGGL(gs_transpose_small_kernel, dim3(gs_cdiv(Ph, 32), gs_cdiv(Ph, 32)), dim3(32, 8), 0, s, pp + e->pW1, Ph, Ph, e->W1T);
    GS_LAUNCH_CHECK();
    gs_gemm_desc a = mk(e->dE2, Ph, 1, e->W1T, 1, Ph, e->dY1, Ph, (int)n2, Ph, Ph);
    a.mask = e->A1; a.ldmask = Ph;
    a.prec = e->operand_mode; a.scratch = e->bsplit_main;
    GS_TRY(gs_gemm_f32(&a, stream));
  }
  for (int ch = 0; ch < nc; ++ch) {       // BN-1 backward, chunk by chunk, in place: dY1 becomes dE1
    long k0, cnt;
    chunk_range(e, ch, &k0, &cnt);
    const float* mu1 = e->chunk_fwd + (long)ch * 4 * Ph; const float* is1 = mu1 + Ph;
    float* bw = e->chunk_bwd + (long)ch * 5 * Ph;
    const int nblk = gs_cdiv(cnt, e->rows_per_block);
    hipLaunchKernelGGL(gs_pge_chunk_dy1_stats_kernel, dim3(nblk), dim3(256), 0, s, e->dY1 + k0 * Ph, e->U, e->V, pp + e->pb0, mu1, is1,
                       n, Ph, k0, cnt, e->rows_per_block, e->parts);
    reduce_parts(e, e->parts, nblk, 2, 2, s);
    hipLaunchKernelGGL(gs_sums_to_float_kernel, dim3(gs_cdiv(Ph, 64)), dim3(64), 0, s, e->sums, 2, Ph, bw + 3 * Ph, bw + 4 * Ph,
                       (float*)nullptr);
    hipLaunchKernelGGL(gs_pge_chunk_bn1_bwd_kernel, dim3(gs_cdiv(cnt * Ph, 256)), dim3(256), 0, s, e->dY1 + k0 * Ph, e->U, e->V,
                       pp + e->pb0, mu1, is1, pp + e->pg1, bw + 3 * Ph, bw + 4 * Ph, n, Ph, k0, cnt);
    GS_LAUNCH_CHECK();
  }
  if (need_param_grads) {
    GS_TRY(gs_reduce_slabs_f32(g + e->pbe1, e->chunk_bwd + 3 * Ph, Ph, nc, 5L * Ph, 0.f, stream));
    GS_TRY(gs_reduce_slabs_f32(g + e->pg1, e->chunk_bwd + 4 * Ph, Ph, nc, 5L * Ph, 0.f, stream));
  }
  // dV[r] = sum_c dE1[(r, c)] ; dU[c] = sum_r dE1[(r, c)]
  hipLaunchKernelGGL(gs_pge_marginal_r_kernel, dim3(n), dim3(256), 0, s, e->dY1, n, Ph, 0, e->dV);
  hipLaunchKernelGGL(gs_pge_marginal_c_kernel, dim3(n), dim3(256), 0, s, e->dY1, n, Ph, n, e->dU);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// the layer-1 activations as a function of U, V (this rank's pair rows start at r_lo) — see gs_panel_a1 in gs_panel.hip
static gs_panel_a1 pge_a1(const gs_gcond* e) {
  const float* pp = e->io.pge_params;
  return gs_panel_a1{e->U, e->V + (long)e->r_lo * e->Ph, (long)e->Ph, pp + e->pb0, e->mu1, e->is1, pp + e->pg1, pp + e->pbe1, e->a1_tab,
                     e->uvs_valid ? e->UsVs : nullptr, e->uvs_valid ? e->UsVs + (long)e->n * e->Ph + (long)e->r_lo * e->Ph : nullptr, 0};
}

// materialise A1 of the current U, V for this rank's pair rows if the forward did not (products that read it as an operand)
static int ensure_a1(gs_gcond* e, hipStream_t s) {
  const int n = e->n, Ph = e->Ph, nr = e->r_hi - e->r_lo;
  const long rows = (long)nr * n;
  const float* pp = e->io.pge_params;
  if (e->a1_valid || rows <= 0) return GS_OK;
  if (Ph == 256)
    hipLaunchKernelGGL((gs_pge_make_a1_v5_kernel<64>), dim3(gs_cdiv(n, 4 * 8), nr), dim3(256), 0, s, e->U, e->V, pp + e->pb0, e->mu1,
                       e->is1, pp + e->pg1, pp + e->pbe1, n, e->r_lo, nr, e->A1);
  else if (Ph == 128)
    hipLaunchKernelGGL((gs_pge_make_a1_v5_kernel<32>), dim3(gs_cdiv(n, 8 * 8), nr), dim3(256), 0, s, e->U, e->V, pp + e->pb0, e->mu1,
                       e->is1, pp + e->pg1, pp + e->pbe1, n, e->r_lo, nr, e->A1);
  else
    hipLaunchKernelGGL(gs_pge_make_a1_v4_kernel, dim3(gs_cdiv(rows * (Ph / 4), 256)), dim3(256), 0, s, e->U, e->V, pp + e->pb0,
                       e->mu1, e->is1, pp + e->pg1, pp + e->pbe1, n, Ph, e->r_lo, nr, e->A1);
  GS_LAUNCH_CHECK();
  e->a1_valid = true;
  return GS_OK;
}

extern "C" int gs_gcond_stage_pge_forward_part(gs_gcond* e, int inference, int part, gs_stream_t stream) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  GS_TRY(quiesce_hoist(e));
  hipStream_t s = (hipStream_t)stream;
  const int n = e->n, d = e->d, Ph = e->Ph, nr = e->r_hi - e->r_lo;
  const long n2 = (long)n * n, rows = (long)nr * n;
  const float* X = e->io.feat_syn;
  const float* pp = e->io.pge_params;
  if (e->pge_chunks > 1 && part != 3) return part == 1 ? pge_forward_chunked(e, stream) : GS_OK;
  if (part == 1) {
    e->fwd_cached = false;
    {  // U = X W0[:, :d]^T ; V = X W0[:, d:]^T
      gs_gemm_desc uv[2];
      uv[0] = mk(X, d, 1, pp + e->pW0, 1, 2L * d, e->U, Ph, n, Ph, d);
      uv[1] = uv[0]; uv[1].B = pp + e->pW0 + d; uv[1].C = e->V;
      GS_TRY(gs_gemm_group_f32(uv, 2, stream));          // independent: one launch
    }
    // the folded layer-1 terms of this forward (exact-split kernels of gs_panel2.hip and gs_dw1_generated: forward operand, backward
    // mask and dW1 operand are then the SAME function max(Us[c] + Vs[r], 0) of the same bits) and the split image of W1 for the
    // layer-2 product come out of the statistics launch (round 4: were two launches of their own per pass; same bits)
    const bool fold_uv = e->operand_mode == 3 && Ph == 256 && n >= 128;
    const bool pre_split = knob_of(e, GS_KNOB_FUSED_CHAINS, g_fused_chains) && e->operand_mode == 3 && rows > 0 && (Ph & 31) == 0;
    const bool fold_here = fold_uv && knob_of(e, GS_KNOB_FUSED_CHAINS, g_fused_chains);
    hipLaunchKernelGGL(gs_pge_bn1_stats_kernel, dim3(gs_cdiv(Ph, 32) + (pre_split ? gs_cdiv((long)Ph * Ph, 1024) : 0)), dim3(1024), 0, s, e->U,
                       e->V, pp + e->pb0, n, Ph, e->mu1, e->is1, e->sumU, e->sumV, pp + e->pg1, pp + e->pbe1, e->a1_tab,   // (+ the table of the generated A1)
                       fold_here ? e->UsVs : (float*)nullptr, fold_here ? e->UsVs + (long)n * Ph : (float*)nullptr, pp + e->pW1,
                       pre_split ? (__bf16*)e->bsplit_main : (__bf16*)nullptr);
    GS_LAUNCH_CHECK();
    e->uvs_valid = false;
    if (fold_uv) {
      if (!fold_here) GS_TRY(gs_pge_scale_uv(e->U, e->V, e->a1_tab, (long)n * Ph, Ph, e->UsVs, e->UsVs + (long)n * Ph, s));
      e->uvs_valid = true;
    }
    if (rows > 0) {
      // E2 = A1 W1^T + b1.  Preferred: A1 is formed in the product's operand path from U, V (gs_panel_forward_a1) and is not
      // written — feature epochs never read it (the backward regenerates the [A1 > 0] mask the same way); PGE epochs
      // materialise it lazily for dW1 (ensure_a1).
      gs_panel_a1 a1 = pge_a1(e);
      a1.b_ready = pre_split ? 1 : 0;
      // device-side spans (two atomics per workgroup, no stream traffic) on every probed launch; the HIP-event bracket — two stream
      // markers, ~5 us of this chain per launch — on every probe_stride-th launch of its pass
      const bool probe = e->probing && e->probe_n < 256;
      const bool probe_ev = probe && (e->probe_seen[inference ? 1 : 0]++ % e->probe_stride) == 0;
      if (probe_ev) {
        for (int q = 0; q < 2; ++q)
          if (!e->probe_ev[2 * e->probe_n + q]) GS_HIP(hipEventCreate(&e->probe_ev[2 * e->probe_n + q]));
      }
      e->a1_valid = false;
      int tiles = 0;
      if (probe) { e->probe_pass[e->probe_n] = inference ? 1 : 0; e->probe_has_ev[e->probe_n] = probe_ev ? 1 : 0; }
      if (probe_ev) GS_HIP(hipEventRecord(e->probe_ev[2 * e->probe_n], s));
      const int rc = gs_panel_forward_a1(&a1, n, nr, pp + e->pW1, Ph, pp + e->pb1, e->E2, Ph, Ph, Ph, e->statp, &tiles, e->operand_mode,
                                         e->bsplit_main, probe ? e->probe_span + 2 * e->probe_n : nullptr,
                                         e->hoisting ? 2 : inference, s);
      if (rc == GS_OK) {
        // (tiles = the number of [2][Ph] partial-sum slabs the kernel wrote: one per 128-row tile, or one per workgroup — gs_panel2.hip)
      } else if (rc != GS_ERR_UNSUPPORTED) {
        return rc;
      } else {
        GS_TRY(ensure_a1(e, s));
        gs_gemm_desc g = mk(e->A1, Ph, 1, pp + e->pW1, 1, Ph, e->E2, Ph, (int)rows, Ph, Ph);
        g.bias = pp + e->pb1;
        g.colstat = e->statp;            // BN-2 batch statistics come out of the epilogue: no second pass over E2
        g.prec = e->operand_mode; g.scratch = e->bsplit_main;
        if (probe_ev) GS_HIP(hipEventRecord(e->probe_ev[2 * e->probe_n], s));
        GS_TRY(gs_gemm_f32(&g, stream));
        tiles = gs_gemm_tiles_m(&g);
      }
      if (probe_ev) GS_HIP(hipEventRecord(e->probe_ev[2 * e->probe_n + 1], s));
      if (probe) e->probe_n++;
      // (unsharded: the last reduction stage is folded into gs_bn2_from_sums; sharded ranks exchange e->sums in between)
      e->bn2_done = false;
      // (<= 256 partial rows — the persistent kernel's one per workgroup: 8 workgroups sum them as fast as the 512 of the two-stage
      // form; the 128-row kernel's 1 611 rows took them 36 us)
      if (tiles <= 256 && knob_of(e, GS_KNOB_FUSED_CHAINS, g_fused_chains) && knob_of(e, GS_KNOB_FOLDED_REDUCTIONS, g_defer_last_stage) && !sharded_r(e)) {
        // round 4: both reduction stages and the statistics in ONE launch, same grouping of the sums (bitwise the three-launch values)
        hipLaunchKernelGGL(gs_bn2_stats_fold_kernel, dim3(gs_cdiv(Ph, 32)), dim3(1024), 0, s, (const float*)e->statp, tiles, tiles > 64 ? 32 : 1,
                           n2, Ph, e->mu2, e->is2);
        e->bn2_done = true;
        e->bn2_slices = 1;
      } else {
        e->bn2_slices = reduce_parts(e, e->statp, tiles, 2, 2, s, knob_of(e, GS_KNOB_FOLDED_REDUCTIONS, g_defer_last_stage) && !sharded_r(e));
      }
      GS_LAUNCH_CHECK();
    } else {
      GS_HIP(hipMemsetAsync(e->sums, 0, sizeof(double) * 2 * Ph, s));
      e->bn2_slices = 1;
      e->bn2_done = false;
    }
    return GS_OK;
  }
  if (part == 2) {
    if (!e->bn2_done)
    hipLaunchKernelGGL(gs_bn2_from_sums_kernel, dim3(gs_cdiv(Ph, 64)), dim3(64), 0, s, e->bn2_slices > 1 ? e->sums2 : e->sums,
                       e->bn2_slices, n2, Ph, e->mu2, e->is2);
    if (sharded_r(e)) GS_HIP(hipMemsetAsync(e->e3, 0, sizeof(float) * n2, s));
    if (rows > 0) {
      float* e3o = e->e3 + (long)e->r_lo * n;
      if (Ph == 256)
        hipLaunchKernelGGL((gs_pge_layer3_v4_kernel<64>), dim3(gs_cdiv(rows, 4 * 8)), dim3(256), 0, s, e->E2, e->mu2, e->is2, pp + e->pg2,
                           pp + e->pbe2, pp + e->pw2, pp + e->pb2, rows, e3o);
      else if (Ph == 128)
        hipLaunchKernelGGL((gs_pge_layer3_v4_kernel<32>), dim3(gs_cdiv(rows, 4 * 16)), dim3(256), 0, s, e->E2, e->mu2, e->is2, pp + e->pg2,
                           pp + e->pbe2, pp + e->pw2, pp + e->pb2, rows, e3o);
      else
        hipLaunchKernelGGL(gs_pge_layer3_kernel, dim3(gs_cdiv(rows, 4)), dim3(256), 0, s, e->E2, e->mu2, e->is2, pp + e->pg2,
                           pp + e->pbe2, pp + e->pw2, pp + e->pb2, rows, Ph, e3o);
    }
    GS_LAUNCH_CHECK();
    return GS_OK;
  }
  if (part != 3) GS_FAIL(GS_ERR_ARG, "gs_gcond_stage_pge_forward_part: part must be 1, 2 or 3");
  float* adj = inference ? e->io.adj_syn : e->adj_raw;
  float* rn = inference ? e->rnorm_in : e->rnorm;
  // structure + degrees in one launch, then the scaling pass of dense_gcn_norm (gs_dense_gcn_norm_fwd's two launches, bitwise)
  hipLaunchKernelGGL(gs_pge_adj_deg_kernel, dim3(n), dim3(256), 0, s, e->e3, n, adj, rn);
  hipLaunchKernelGGL(gs_pge_norm_scale_kernel, dim3(gs_cdiv(n2, 256)), dim3(256), 0, s, adj, rn, n, inference ? e->adj_norm_in : e->adj_norm);
  GS_LAUNCH_CHECK();
  if (inference) e->fwd_cached = true;     // U, V, A1, E2, BN statistics and e3 of the current (X~, theta_PGE) are in place
  return GS_OK;
}

// adopt the last inference forward as this iteration's training forward (three small device copies)
static int adopt_cached_forward(gs_gcond* e, gs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const long n = e->n;
  GS_HIP(hipMemcpyAsync(e->adj_raw, e->io.adj_syn, sizeof(float) * n * n, hipMemcpyDeviceToDevice, s));
  GS_HIP(hipMemcpyAsync(e->adj_norm, e->adj_norm_in, sizeof(float) * n * n, hipMemcpyDeviceToDevice, s));
  GS_HIP(hipMemcpyAsync(e->rnorm, e->rnorm_in, sizeof(float) * 2 * n, hipMemcpyDeviceToDevice, s));
  return GS_OK;
}

extern "C" int gs_gcond_adopt_cached_forward(gs_gcond* e, gs_stream_t stream) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  GS_TRY(quiesce_hoist(e));
  if (!e->fwd_cached) GS_FAIL(GS_ERR_STATE, "gs_gcond_adopt_cached_forward: no inference forward of the current state is cached");
  return adopt_cached_forward(e, stream);
}

// ---- block structure: dst[b][i][j] = src[pos[b][i]][pos[b][j]] and back -------------------------------------------------
__global__ void gs_block_gather_kernel(const float* __restrict__ src, const int* __restrict__ pos, int B, int nb, int n,
                                       float* __restrict__ dst) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * nb * nb) return;
  const int j = idx % nb, i = (idx / nb) % nb, b = idx / ((long)nb * nb);
  dst[idx] = src[(long)pos[b * nb + i] * n + pos[b * nb + j]];
}

__global__ void gs_block_scatter_kernel(const float* __restrict__ src, const int* __restrict__ pos, int B, int nb, int n,
                                        float* __restrict__ dst) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * nb * nb) return;
  const int j = idx % nb, i = (idx / nb) % nb, b = idx / ((long)nb * nb);
  dst[(long)pos[b * nb + i] * n + pos[b * nb + j]] = src[idx];
}

static int block_gather(gs_gcond* e, const float* src, float* dst, hipStream_t s) {
  const long tot = (long)e->blk_B * e->blk_n * e->blk_n;
  hipLaunchKernelGGL(gs_block_gather_kernel, dim3(gs_cdiv(tot, 256)), dim3(256), 0, s, src, e->blk_pos, e->blk_B, e->blk_n, e->n, dst);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

static int block_scatter(gs_gcond* e, const float* src, float* dst, hipStream_t s) {
  const long tot = (long)e->blk_B * e->blk_n * e->blk_n;
  GS_HIP(hipMemsetAsync(dst, 0, sizeof(float) * (size_t)e->n * e->n, s));
  hipLaunchKernelGGL(gs_block_scatter_kernel, dim3(gs_cdiv(tot, 256)), dim3(256), 0, s, src, e->blk_pos, e->blk_B, e->blk_n, e->n, dst);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// Q = A A for a block-structured A: per block Q_b = A_b A_b (2 B n_b^3 instead of 2 (B n_b)^3 FLOP)
static int block_square(gs_gcond* e, const float* A, float* Q, gs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const int nb = e->blk_n;
  GS_TRY(block_gather(e, A, e->blk_A, s));
  gs_gemm_desc g = mk(e->blk_A, nb, 1, e->blk_A, nb, 1, e->blk_X, nb, nb, nb, nb);
  g.nb1 = e->blk_B; g.sAb1 = g.sBb1 = g.sCb1 = (long)nb * nb;
  GS_TRY(gs_gemm_f32(&g, stream));
  return block_scatter(e, e->blk_X, Q, s);
}

// dA^ = dQ A^T + A^T dQ restricted to the blocks (the entries outside them meet structural zeros of adj + I downstream)
static int block_dadj(gs_gcond* e, const float* A, const float* dQ, float* dAn, gs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const int nb = e->blk_n;
  const long bs = (long)nb * nb;
  GS_TRY(block_gather(e, A, e->blk_A, s));
  GS_TRY(block_gather(e, dQ, e->blk_X, s));
  gs_gemm_desc a = mk(e->blk_X, nb, 1, e->blk_A, 1, nb, e->blk_Y, nb, nb, nb, nb);
  a.nb1 = e->blk_B; a.sAb1 = a.sBb1 = a.sCb1 = bs;
  GS_TRY(gs_gemm_f32(&a, stream));
  gs_gemm_desc b = mk(e->blk_A, 1, nb, e->blk_X, nb, 1, e->blk_Y, nb, nb, nb, nb);
  b.nb1 = e->blk_B; b.sAb1 = b.sBb1 = b.sCb1 = bs; b.beta = 1.f;
  GS_TRY(gs_gemm_f32(&b, stream));
  return block_scatter(e, e->blk_Y, dAn, s);
}

// Y[pos[b][i]][k] = Yb[b][i][k] (times [mask > 0] when a mask is given: the ReLU derivative, like gs_gemm_desc.mask)
__global__ __launch_bounds__(256) void gs_block_scatter_rows_kernel(const float* __restrict__ Yb, const int* __restrict__ pos, long rows,
                                                                    int F, float* __restrict__ Y, long ldy,
                                                                    const float* __restrict__ mask, long ldm) {
  const int lane = threadIdx.x & 63;
  const long r = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= rows) return;
  const long p = pos[r];
  for (int k = lane; k < F; k += 64) {
    float v = Yb[r * F + k];
    if (mask && !(mask[p * ldm + k] > 0.f)) v = 0.f;
    Y[p * ldy + k] = v;
  }
}

// Y = Qm X (or Qm^T X) for a block-structured Qm whose gathered blocks [B][n_b][n_b] are in `Qb`: rows of X gathered per block,
// B small products, rows scattered back.  X, Y: [n][F] with row stride F.
static int block_apply(gs_gcond* e, const float* Qb, bool transpose, const float* X, int F, float* Y, const float* mask, long ldm,
                       gs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const int nb = e->blk_n, B = e->blk_B;
  GS_TRY(gs_gather_rows_f32(X, F, e->blk_pos, e->n, F, e->blk_F1, F, stream));
  gs_gemm_desc g = transpose ? mk(Qb, 1, nb, e->blk_F1, F, 1, e->blk_F2, F, nb, F, nb) : mk(Qb, nb, 1, e->blk_F1, F, 1, e->blk_F2, F, nb, F, nb);
  g.nb1 = B; g.sAb1 = (long)nb * nb; g.sBb1 = g.sCb1 = (long)nb * F;
  GS_TRY(gs_gemm_f32(&g, stream));
  hipLaunchKernelGGL(gs_block_scatter_rows_kernel, dim3(gs_cdiv((long)e->n, 4)), dim3(256), 0, s, e->blk_F2, e->blk_pos, (long)e->n, F, Y,
                     (long)F, mask, ldm);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_gcond_forward_cached(gs_gcond* e) { return e && e->reuse_fwd && e->fwd_cached ? 1 : 0; }
extern "C" int gs_gcond_forward_hoisted(gs_gcond* e) { return e && e->fwd_hoisted && e->hoist_pending ? 1 : 0; }

extern "C" int gs_gcond_stage_pge_forward(gs_gcond* e, int inference, gs_stream_t stream) {
  if (e && (sharded_r(e))) GS_FAIL(GS_ERR_STATE, "gs_gcond_stage_pge_forward: sharded engines must run the _part entry points with all-reduces in between");
  GS_TRY(gs_gcond_stage_pge_forward_part(e, inference, 1, stream));
  GS_TRY(gs_gcond_stage_pge_forward_part(e, inference, 2, stream));
  return gs_gcond_stage_pge_forward_part(e, inference, 3, stream);
}

// match_loss of this rank's Cl classes and D = d loss / d gw_syn (condensation/utils.py:12-106): one launch (a workgroup per class)
// when the chains are fused and the per-class statistics row fits its LDS image, else statistics / finalize / apply
// ---- nlayers != 2 (models/sgc.py:45-51: K propagations): Q = A^^K and d A^ from d Q (oracle/analytic.py: adj_power / adj_power_bwd) ----
// The powers A^^2 .. A^^(K-1) stay in e->Apow + i n'^2 (the backward needs them); n'^3 products of a few hundred rows: latency-bound.
static int adj_powers(gs_gcond* e, const float* A, float* Qout, gs_stream_t stream) {
  const int n = e->n;
  const long n2 = (long)n * n;
  const float* prev = A;
  for (int i = 2; i <= e->K; ++i) {
    float* out = i == e->K ? Qout : e->Apow + (long)i * n2;
    gs_gemm_desc g = mk(prev, n, 1, A, n, 1, out, n, n, n, n);
    GS_TRY(gs_gemm_f32(&g, stream));
    prev = out;
  }
  return GS_OK;
}
// dA = sum over the K positions of A^ in the product: (A^^i)^T dQ (A^^(K-1-i))^T, i = 0 .. K-1 (powers from adj_powers of the SAME A)
static int adj_powers_bwd(gs_gcond* e, const float* A, const float* dQ, float* dA, gs_stream_t stream) {
  const int n = e->n, K = e->K;
  const long n2 = (long)n * n;
  auto pw = [&](int i) -> const float* { return i == 1 ? A : e->Apow + (long)i * n2; };
  {      // i = 0: dQ (A^^(K-1))^T
    gs_gemm_desc g = mk(dQ, n, 1, pw(K - 1), 1, n, dA, n, n, n, n);
    GS_TRY(gs_gemm_f32(&g, stream));
  }
  for (int i = 1; i < K - 1; ++i) {      // (A^^i)^T [dQ (A^^(K-1-i))^T]
    gs_gemm_desc t = mk(dQ, n, 1, pw(K - 1 - i), 1, n, e->Apow_t, n, n, n, n);
    GS_TRY(gs_gemm_f32(&t, stream));
    gs_gemm_desc g = mk(pw(i), 1, n, e->Apow_t, n, 1, dA, n, n, n, n);
    g.beta = 1.f;
    GS_TRY(gs_gemm_f32(&g, stream));
  }
  {      // i = K - 1: (A^^(K-1))^T dQ
    gs_gemm_desc g = mk(pw(K - 1), 1, n, dQ, n, 1, dA, n, n, n, n);
    g.beta = 1.f;
    GS_TRY(gs_gemm_f32(&g, stream));
  }
  return GS_OK;
}

static int match_distance_folded(gs_gcond* e, const float* gwr, const float* gws, float* stats, float* ab, float* Dv, int Cl, const float* s2,
                                 hipStream_t s);
static int match_distance(gs_gcond* e, const float* gwr, const float* gws, float* stats, float* ab, float* Dv, int Cl, hipStream_t s) {
  if (Cl <= 0) { e->dist_shares = 0; return GS_OK; }
  if (!(e->with_bn && e->cfg.dis_metric != 0)) return match_distance_folded(e, gwr, gws, stats, ab, Dv, Cl, e->bn_s2, s);
  // with_bn under 'mse' / 'cos': the distance runs on the gradients of the TRUE parameters (BatchNorm weight / bias included), then
  // D is mapped back onto the folded model (gs_kernels.h: gs_bn_unfold_class_grads_kernel / gs_bn_fold_class_D_kernel)
  const long P = e->P, c0 = e->c_lo;
  const float* th = e->io.model_params;
  float* tr = e->gw_true_r + c0 * P; float* ts = e->gw_true_s + c0 * P;
  const dim3 grid(gs_cdiv(e->h, 32), Cl), block(256);
  hipLaunchKernelGGL(gs_bn_unfold_class_grads_kernel, grid, block, 0, s, gwr, P, th, tr, P, e->d, e->h, e->oW1, e->ob1, e->og, e->obeta, e->P_base);
  hipLaunchKernelGGL(gs_bn_unfold_class_grads_kernel, grid, block, 0, s, gws, P, th, ts, P, e->d, e->h, e->oW1, e->ob1, e->og, e->obeta, e->P_base);
  GS_LAUNCH_CHECK();
  GS_TRY(match_distance_folded(e, tr, ts, stats, ab, Dv, Cl, nullptr, s));
  hipLaunchKernelGGL(gs_bn_fold_class_D_kernel, grid, block, 0, s, Dv, P, th, e->d, e->h, e->oW1, e->ob1, e->og, e->obeta);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

static int match_distance_folded(gs_gcond* e, const float* gwr, const float* gws, float* stats, float* ab, float* Dv, int Cl, const float* s2,
                                 hipStream_t s) {
  const long P = e->P, c0 = e->c_lo;
  if (Cl <= 0) return GS_OK;
  if (knob_of(e, GS_KNOB_FUSED_CHAINS, g_fused_chains) && e->pd.total_cols <= GS_DIST_FUSED_COLS) {
    const int G = e->cfg.dis_metric == 2 || e->pd.nparams < 2 ? 1 : GS_DIST_SPLIT;      // ('cos' couples all columns of a class)
    hipLaunchKernelGGL(gs_dist_fused_kernel, dim3(Cl, G), dim3(1024), 0, s, gwr, gws, P, e->pd, e->cfg.dis_metric, e->io.syn_lo + c0,
                       e->io.syn_hi + c0, e->n, ab, e->loss_c + c0, Dv, s2, e->dist_part + c0 * GS_DIST_SPLIT);
    e->dist_shares = G > 1 ? G : 0;      // (the V kernel of this stage call finishes loss_c from the shares)
    GS_LAUNCH_CHECK();
    return GS_OK;
  }
  e->dist_shares = 0;
  hipLaunchKernelGGL(gs_dist_stats_kernel, dim3(e->pd.nparams, Cl), dim3(256), 0, s, gwr, gws, P, e->pd, stats);
  hipLaunchKernelGGL(gs_dist_finalize_kernel, dim3(Cl), dim3(256), 0, s, stats, e->pd, e->cfg.dis_metric, e->io.syn_lo + c0,
                     e->io.syn_hi + c0, e->n, ab, e->loss_c + c0, s2);
  hipLaunchKernelGGL(gs_dist_apply_kernel, dim3(gs_cdiv((long)Cl * P, 256)), dim3(256), 0, s, gwr, gws, P, e->pd, ab, Dv, Cl);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// ================================================================================================
// stage: synthetic side — gw_syn per class, matching loss, closed-form d loss / d X~ (direct path)
// and d loss / d A^ -> d adj     (gcond_base.py:226-239 + autograd through them).
// Writes this rank's PARTIAL [d_feat | d_adj | loss_c] ("ar_syn"); gs_gcond_stage_syn_finish books the loss.
// ================================================================================================
#include "gs_deep.inc"
#include "gs_gcnk.inc"

// A block structure (MSGC's batch of disconnected graphs) with nlayers != 2: the per-block products are written for A^ A^ only; the
// dense products are the same values on a block-diagonal A^ (B^2 times the work of the per-block form, on matrices of a few hundred
// rows) — the stage runs as if no block structure were set.  Entries of d A^ outside the blocks meet structural zeros downstream.
struct gs_blk_dense_guard {
  gs_gcond* e; const int* keep;
  explicit gs_blk_dense_guard(gs_gcond* e_) : e(e_), keep(e_->blk_pos) { if (e->K != 2) e->blk_pos = nullptr; }
  ~gs_blk_dense_guard() { e->blk_pos = keep; }
};

extern "C" int gs_gcond_stage_syn_match(gs_gcond* e, gs_stream_t stream) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  GS_TRY(quiesce_hoist(e));
  if (e->deep) return deep_syn_match(e, stream);
  if (e->gk) return gcnk_syn_match(e, stream);
  gs_blk_dense_guard blk_guard(e);
  hipStream_t s = (hipStream_t)stream;
  const int C = e->C, Cl = e->c_hi - e->c_lo, n = e->n, d = e->d, h = e->h, F = e->F, mx = e->max_nc;
  const long P = e->P, c0 = e->c_lo;
  const bool part = sharded_c(e);
  const float* X = e->io.feat_syn;
  const float* th = model_th(e);
  const float* A = e->adj_norm;
  const float* Wl = th + e->oWl;
  const float* bl = th + e->obl;
  // per-class views of this rank's classes
  const long* off_lo = e->off_lo + c0; const long* off_loF = e->off_loF + c0; const long* off_loC = e->off_loC + c0;
  const long* off_lon = e->off_lon + c0; const int* ncls = e->ncls + c0;
  float* gws = e->gw_syn + c0 * P; const float* gwr = e->gw_real + c0 * P; float* Dv = e->D + c0 * P;
  if (e->gcn) {
    // ---- GCN condense model: S1 = A X ; H1 = relu(S1 W1 + b1) ; S2 = A H1 ; Z = S2 W2 + b2 (oracle/analytic.py:syn_match_gcn)
    const float* W1 = th + e->oW1;
    float* S2 = e->Ss;                               // [n x h]
    { gs_gemm_desc g = mk(A, n, 1, X, d, 1, e->S1, d, n, d, n); GS_TRY(gs_gemm_f32(&g, stream)); }
    { gs_gemm_desc g = mk(e->S1, d, 1, W1, h, 1, e->H1s, h, n, h, d); g.bias = th + e->ob1; g.relu = 1; GS_TRY(gs_gemm_f32(&g, stream)); }
    { gs_gemm_desc g = mk(A, n, 1, e->H1s, h, 1, S2, h, n, h, n); GS_TRY(gs_gemm_f32(&g, stream)); }
    { gs_gemm_desc g = mk(S2, h, 1, Wl, C, 1, e->Zs, C, n, C, h); GS_TRY(gs_gemm_f32(&g, stream)); }
    hipLaunchKernelGGL(gs_syn_softmax_grad_kernel, dim3(gs_cdiv(n, 4)), dim3(256), 0, s, e->Zs, e->ones, bl, e->io.labels_syn,
                       e->io.syn_lo, e->io.syn_hi, n, C, e->Ps, e->Gs, 0);
    GS_LAUNCH_CHECK();
    { gs_gemm_desc g = mk(e->Gs, C, 1, Wl, 1, C, e->Tmat, h, n, h, C); GS_TRY(gs_gemm_f32(&g, stream)); }   // T = G W2^T (rows partition by class)
    if (part) {
      GS_HIP(hipMemsetAsync(e->loss_c, 0, sizeof(float) * C, s));
      GS_HIP(hipMemsetAsync(e->dSs, 0, sizeof(float) * (long)n * h, s));
      GS_HIP(hipMemsetAsync(e->dAn, 0, sizeof(float) * (long)n * n, s));
      GS_HIP(hipMemsetAsync(e->dS1, 0, sizeof(float) * (long)n * d, s));
      GS_HIP(hipMemsetAsync(e->Rmat, 0, sizeof(float) * (long)n * h, s));
    }
    const float* Dl2 = Dv + e->oWl;
    if (Cl > 0) {
      float* dHp = e->dHp + c0 * n * h; float* A1m = e->A1m + c0 * n * h;
      // dH1p_c = (A[lo:hi]^T T[lo:hi]) * mask ; gW1_c = S1^T dH1p_c ; gb1_c ; gW2_c = S2[lo:hi]^T G[lo:hi] ; gb2_c = colsum G[lo:hi]
      gs_gemm_desc a = mk(A, 1, n, e->Tmat, h, 1, dHp, h, n, h, mx);
      a.nb1 = Cl; a.offA = off_lon; a.offB = off_loF; a.Kb = ncls; a.sCb1 = (long)n * h; a.mask = e->H1s; a.ldmask = h;
      GS_TRY(gs_gemm_f32(&a, stream));
      gs_gemm_desc w3[3];
      w3[0] = mk(e->S1, 1, d, dHp, h, 1, gws + e->oW1, h, d, h, n);
      w3[0].nb1 = Cl; w3[0].sBb1 = (long)n * h; w3[0].sCb1 = P;
      w3[1] = mk(S2, 1, h, e->Gs, C, 1, gws + e->oWl, C, h, C, mx);
      w3[1].nb1 = Cl; w3[1].offA = off_loF; w3[1].offB = off_loC; w3[1].Kb = ncls; w3[1].sCb1 = P;
      w3[2] = mk(e->ones, 0, 0, e->Gs, C, 1, gws + e->obl, C, 1, C, mx);
      w3[2].nb1 = Cl; w3[2].offB = off_loC; w3[2].Kb = ncls; w3[2].sCb1 = P;
      GS_TRY(gs_gemm_group_f32(w3, 3, stream));          // gW1_c, gW2_c, gb2_c: independent
      if (e->cfg.dis_metric != 0) GS_TRY(colsum_batched(e, dHp, (long)n * h, n, nullptr, h, Cl, gws + e->ob1, P, stream));
      float* stats = e->stats + c0 * e->pd.total_cols * 3; float* ab = e->ab + c0 * e->pd.total_cols * 2;
      if (e->join_pending) { GS_HIP(hipStreamWaitEvent(s, e->ev_join, 0)); e->join_pending = false; }
      GS_TRY(match_distance(e, gwr, gws, stats, ab, Dv, Cl, s));
      // A1m_c = (S1 D1_c + Db1_c) * mask ; dS1 slabs_c = dH1p_c D1_c^T ; dA[lo:hi] = T[lo:hi] A1m_c^T ; R[lo:hi] = A[lo:hi] A1m_c
      gs_gemm_desc mx2[2];
      mx2[0] = mk(e->S1, d, 1, Dv + e->oW1, h, 1, A1m, h, n, h, d);
      mx2[0].nb1 = Cl; mx2[0].sBb1 = P; mx2[0].sCb1 = (long)n * h; mx2[0].bias = Dv + e->ob1; mx2[0].sbias_b1 = P;
      mx2[0].mask = e->H1s; mx2[0].ldmask = h;
      mx2[1] = mk(dHp, h, 1, Dv + e->oW1, 1, h, e->slabs_x, d, n, d, h);
      mx2[1].nb1 = Cl; mx2[1].sAb1 = (long)n * h; mx2[1].sBb1 = P; mx2[1].sCb1 = (long)n * d;
      GS_TRY(gs_gemm_group_f32(mx2, 2, stream));
      GS_TRY(gs_reduce_slabs_f32(e->dS1, e->slabs_x, (long)n * d, Cl, (long)n * d, 0.f, stream));
      gs_gemm_desc q4[4];                                   // four products with distinct outputs: one launch
      q4[0] = mk(e->Tmat, h, 1, A1m, 1, h, e->dAn, n, mx, n, h);
      q4[0].nb1 = Cl; q4[0].offA = off_loF; q4[0].Mb = ncls; q4[0].sBb1 = (long)n * h; q4[0].offC = off_lon;
      q4[1] = mk(A, n, 1, A1m, h, 1, e->Rmat, h, mx, h, n);
      q4[1].nb1 = Cl; q4[1].offA = off_lon; q4[1].Mb = ncls; q4[1].sBb1 = (long)n * h; q4[1].offC = off_loF;
      // Ug[lo:hi] = S2[lo:hi] D2_c  (+ R W2 below, + Db2 inside the V kernel)
      q4[2] = mk(S2, h, 1, Dl2, C, 1, e->Ug, C, mx, C, h);
      q4[2].nb1 = Cl; q4[2].offA = off_loF; q4[2].Mb = ncls; q4[2].sBb1 = P; q4[2].offC = off_loC;
      // dS2[lo:hi] = G[lo:hi] D2_c^T
      q4[3] = mk(e->Gs, C, 1, Dl2, 1, C, e->dSs, h, mx, h, C);
      q4[3].nb1 = Cl; q4[3].offA = off_loC; q4[3].Mb = ncls; q4[3].sBb1 = P; q4[3].offC = off_loF;
      GS_TRY(gs_gemm_group_f32(q4, 4, stream));
    } else if (e->join_pending) {
      GS_HIP(hipStreamWaitEvent(s, e->ev_join, 0));
      e->join_pending = false;
    }
    { gs_gemm_desc g = mk(e->Rmat, h, 1, Wl, C, 1, e->Ug, C, n, C, h); g.beta = 1.f; GS_TRY(gs_gemm_f32(&g, stream)); }   // Ug += R W2
    hipLaunchKernelGGL(gs_syn_v_kernel, dim3(gs_cdiv(n, 4)), dim3(256), 0, s, e->Ps, e->Gs, e->Ug, e->ones, e->D, P, e->obl, bl,
                       e->io.labels_syn, e->io.syn_lo, e->io.syn_hi, n, C, e->c_lo, e->c_hi, e->Vs, e->drs,
                       e->fold_finish ? (const float*)e->loss_c : (const float*)nullptr, C, e->loss_sum, e->loss_last, e->steps,
                     e->dist_shares ? (const double*)e->dist_part : (const double*)nullptr, e->dist_shares, e->loss_c);
    GS_LAUNCH_CHECK();
    { gs_gemm_desc g = mk(e->Vs, C, 1, Wl, 1, C, e->dSs, h, n, h, C); g.beta = 1.f; GS_TRY(gs_gemm_f32(&g, stream)); }    // dS2 += V W2^T
    {
      gs_gemm_desc g2[2];
      g2[0] = mk(e->dSs, h, 1, e->H1s, 1, h, e->dAn, n, n, n, h); g2[0].beta = 1.f;                      // dA += dS2 H1^T
      g2[1] = mk(A, 1, n, e->dSs, h, 1, e->dH1m, h, n, h, n); g2[1].mask = e->H1s; g2[1].ldmask = h;     // (A^T dS2)*mask
      GS_TRY(gs_gemm_group_f32(g2, 2, stream));
    }
    { gs_gemm_desc g = mk(e->dH1m, h, 1, W1, 1, h, e->dS1, d, n, d, h); g.beta = 1.f; GS_TRY(gs_gemm_f32(&g, stream)); }   // dS1 += dH1p W1^T
    {
      gs_gemm_desc g2[2];
      g2[0] = mk(e->dS1, d, 1, X, 1, d, e->dAn, n, n, n, d); g2[0].beta = 1.f;     // dA += dS1 X^T
      g2[1] = mk(A, 1, n, e->dS1, d, 1, e->d_feat, d, n, d, n);                    // dX = A^T dS1
      GS_TRY(gs_gemm_group_f32(g2, 2, stream));
    }
    if (e->identity) return GS_OK;
    return gs_dense_gcn_norm_bwd(e->adj_raw, e->rnorm, e->dAn, n, e->d_adj, stream);
  }
  const float* Hin = X;   // [n x F]
  {  // Q = A A ; rsq = Q 1 ; (T = 2) H1 = relu(X W1 + b1) — the two products are independent: one launch
    gs_gemm_desc qh[2];
    qh[0] = mk(A, n, 1, A, n, 1, e->Q, n, n, n, n);
    if (e->T == 2) {
      qh[1] = mk(X, d, 1, th + e->oW1, h, 1, e->H1s, h, n, h, d);
      qh[1].bias = th + e->ob1; qh[1].relu = 1;
      Hin = e->H1s;
    }
    if (e->blk_pos) {
      if (e->K != 2) GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond: nlayers != 2 with a block structure");
      GS_TRY(block_square(e, A, e->Q, stream));
      if (e->T == 2) GS_TRY(gs_gemm_f32(&qh[1], stream));
    } else if (e->K != 2) {      // Q = A^^K
      GS_TRY(adj_powers(e, A, e->Q, stream));
      if (e->T == 2) GS_TRY(gs_gemm_f32(&qh[1], stream));
    } else {
      GS_TRY(gs_gemm_group_f32(qh, e->T == 2 ? 2 : 1, stream));
    }
    hipLaunchKernelGGL(gs_rowsum_kernel, dim3(gs_cdiv(n, 4)), dim3(256), 0, s, e->Q, n, n, e->rsq);
  }
  // dropout (models/sgc.py:42): the synthetic forward of class c draws its own mask over all n' rows, so the hidden activations — and
  // every product behind them — exist once per class: H1c[c] = H1 * multiplier_c.  Only the rows of class c of S = Q H1c[c] enter its
  // loss (gcond_base.py:228-230), so the work stays that of one shared forward; the masked products run class by class on the generic
  // kernels (no shipped configuration uses dropout: this path is built for parity, not tuned).
  const bool drop = e->drop_thr != 0 && e->T == 2;
  float* H1c = drop ? e->H1c + c0 * n * h : nullptr;
  if (drop) {
    if (e->blk_pos) GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond: dropout != 0 with a block structure");
    if (Cl > 0)
      hipLaunchKernelGGL(gs_drop_rows_kernel, dim3(gs_cdiv((long)Cl * n * (h / 4), 256)), dim3(256), 0, s, e->H1s, H1c, n, h, Cl,
                         gs_drop_key0(e->cfg.seed, e->drop_iter, 1), (int)c0, e->drop_thr, e->drop_q);
    GS_LAUNCH_CHECK();
  }
  {  // S = Q Hin ; Z = S Wl ; P, G
    if (drop) {      // S[lo:hi] = Q[lo:hi] H1c[c]
      if (part) GS_HIP(hipMemsetAsync(e->Ss, 0, sizeof(float) * (long)n * F, s));
      if (Cl > 0) {
        gs_gemm_desc g = mk(e->Q, n, 1, H1c, h, 1, e->Ss, F, mx, F, n);
        g.nb1 = Cl; g.offA = off_lon; g.Mb = ncls; g.sBb1 = (long)n * h; g.offC = off_loF;
        GS_TRY(gs_gemm_f32(&g, stream));
      }
    } else if (e->blk_pos) {
      GS_TRY(block_apply(e, e->blk_X, false, Hin, F, e->Ss, nullptr, 0, stream));      // blk_X holds the blocks of Q (block_square)
    } else {
      gs_gemm_desc g = mk(e->Q, n, 1, Hin, F, 1, e->Ss, F, n, F, n);
      GS_TRY(gs_gemm_f32(&g, stream));
    }
    {
      gs_gemm_desc z = mk(e->Ss, F, 1, Wl, C, 1, e->Zs, C, n, C, F);
      GS_TRY(gs_gemm_f32(&z, stream));
      hipLaunchKernelGGL(gs_syn_softmax_grad_kernel, dim3(gs_cdiv(n, 4)), dim3(256), 0, s, e->Zs, e->rsq, bl, e->io.labels_syn,
                         e->io.syn_lo, e->io.syn_hi, n, C, e->Ps, e->Gs, 0);
      GS_LAUNCH_CHECK();
    }
  }
  if (part) {   // rows / classes of other ranks contribute zero from this rank
    GS_HIP(hipMemsetAsync(e->loss_c, 0, sizeof(float) * C, s));
    GS_HIP(hipMemsetAsync(e->dSs, 0, sizeof(float) * (long)n * F, s));
    GS_HIP(hipMemsetAsync(e->dQ, 0, sizeof(float) * (long)n * n, s));
    if (e->T == 2) GS_HIP(hipMemsetAsync(e->d_feat, 0, sizeof(float) * (long)n * d, s));
  }
  const float* Dl = Dv + e->oWl;   // + c*P : d/d gWl_c
  // two-segment / regrouped launches of the second-order chain (VERDICT r03 "next" 5), same knob as the fused inner steps
  const bool merge = knob_of(e, GS_KNOB_FUSED_CHAINS, g_fused_chains) && !e->blk_pos;
  bool ds_done = false;
  if (Cl > 0) {
    float* dH2 = e->T == 2 ? e->dH2 + c0 * n * C : nullptr; float* dHp = e->T == 2 ? e->dHp + c0 * n * h : nullptr;
    float* A1m = e->T == 2 ? e->A1m + c0 * n * h : nullptr; float* Bc = e->T == 2 ? e->Bc + c0 * n * C : nullptr;
    // two chained launches for the per-class products of the two-transform model (gs_match.hip) instead of six batched ones
    const bool mclass = e->T == 2 && knob_of(e, GS_KNOB_CHAINED_PRODUCTS, g_chained_products) && !e->blk_pos && !drop &&
                        gs_match_class_eligible(n, h, d, C);
    {
      // gWl_c, gbl_c and (T = 2) dH2_c = Q[lo:hi]^T G[lo:hi] only read G: one launch
      gs_gemm_desc g3[3];
      g3[0] = mk(e->Ss, 1, F, e->Gs, C, 1, gws + e->oWl, C, F, C, mx);
      g3[0].nb1 = Cl; g3[0].offA = off_loF; g3[0].offB = off_loC; g3[0].Kb = ncls; g3[0].sCb1 = P;
      g3[1] = mk(e->rsq, 0, 1, e->Gs, C, 1, gws + e->obl, C, 1, C, mx);
      g3[1].nb1 = Cl; g3[1].offA = off_lo; g3[1].offB = off_loC; g3[1].Kb = ncls; g3[1].sCb1 = P;
      if (e->T == 2 && !mclass) {
        g3[2] = mk(e->Q, 1, n, e->Gs, C, 1, dH2, C, n, C, mx);
        g3[2].nb1 = Cl; g3[2].offA = off_lon; g3[2].offB = off_loC; g3[2].Kb = ncls; g3[2].sCb1 = (long)n * C;
      }
      GS_TRY(gs_gemm_group_f32(g3, e->T == 2 && !mclass ? 3 : 2, stream));
    }
    if (mclass) {
      // T = G Wl^T once (rows partition by class), then dHp_c = (Q[lo:hi]^T T[lo:hi]) * mask ; gW1_c = X^T dHp_c ; gb1_c in ONE launch
      float* Tm = e->A1m;      // (the A1m_c buffers are not written on this path)
      gs_gemm_desc t = mk(e->Gs, C, 1, Wl, 1, C, Tm, h, n, h, C);
      GS_TRY(gs_gemm_f32(&t, stream));
      gs_match1_args a1;
      a1.Q = e->Q; a1.T = Tm; a1.mask = e->H1s; a1.X = X; a1.lo = off_lo; a1.nc = ncls; a1.dHp = dHp; a1.s_dhp = (long)n * h;
      a1.gW1 = gws + e->oW1; a1.s_gw = P; a1.gb1 = e->cfg.dis_metric != 0 ? gws + e->ob1 : nullptr; a1.n = n; a1.h = h; a1.d = d; a1.ldx = d;
      GS_TRY(gs_match_first(&a1, Cl, s));
    } else if (e->T == 2) {
      // dHp_c = (dH2_c W2^T) * mask ; gW1_c = X^T dHp_c ; gb1_c = colsum
      gs_gemm_desc b = mk(dH2, C, 1, Wl, 1, C, dHp, h, n, h, C);
      b.nb1 = Cl; b.sAb1 = (long)n * C; b.sCb1 = (long)n * h; b.mask = e->H1s; b.ldmask = h;
      if (drop) {      // the classes' rows under each other: one product whose mask rows are the classes' own ([H1c > 0] / (1 - p))
        b = mk(dH2, C, 1, Wl, 1, C, dHp, h, Cl * n, h, C);
        b.mask = H1c; b.ldmask = h; b.alpha = e->drop_q;
      }
      GS_TRY(gs_gemm_f32(&b, stream));
      gs_gemm_desc c = mk(X, 1, d, dHp, h, 1, gws + e->oW1, h, d, h, n);
      c.nb1 = Cl; c.sBb1 = (long)n * h; c.sCb1 = P;
      GS_TRY(gs_gemm_f32(&c, stream));
      if (e->cfg.dis_metric != 0) GS_TRY(colsum_batched(e, dHp, (long)n * h, n, nullptr, h, Cl, gws + e->ob1, P, stream));
    }
    // ---- matching distance and D = d loss / d gw_syn (already scaled by n_c / n') -------------------
    if (e->join_pending) {      // gw_real comes from the real-side branch running on the aux stream
      GS_HIP(hipStreamWaitEvent(s, e->ev_join, 0));
      e->join_pending = false;
    }
    float* stats = e->stats + c0 * e->pd.total_cols * 3; float* ab = e->ab + c0 * e->pd.total_cols * 2;
    GS_TRY(match_distance(e, gwr, gws, stats, ab, Dv, Cl, s));
    // ---- second-order backward, per-class part ------------------------------------------------------
    if (e->T == 2) {
      // A1m_c = (X D1_c + Db1_c) * mask ; dX slabs_c = dHp_c D1_c^T ; Bc = A1m_c W2
      if (mclass) {      // one launch; A1m_c stays in the accumulator registers of the wave that feeds it to Bc
        gs_match2_args a2;
        a2.X = X; a2.D = Dv; a2.mask = e->H1s; a2.Wl = Wl; a2.dHp = dHp; a2.sD = P; a2.oW1 = e->oW1; a2.ob1 = e->ob1; a2.s_dhp = (long)n * h;
        a2.Bc = Bc; a2.s_bc = (long)n * C; a2.slabs = e->skip_dfeat ? nullptr : e->slabs_x; a2.s_slab = (long)n * d; a2.n = n; a2.h = h; a2.d = d;
        a2.ldx = d; a2.C = C;
        GS_TRY(gs_match_second(&a2, Cl, s));
        if (!e->skip_dfeat) GS_TRY(gs_reduce_slabs_f32(e->d_feat, e->slabs_x, (long)n * d, Cl, (long)n * d, 0.f, stream));
      } else {
        gs_gemm_desc ax[2];
        ax[0] = mk(X, d, 1, Dv + e->oW1, h, 1, A1m, h, n, h, d);
        ax[0].nb1 = Cl; ax[0].sBb1 = P; ax[0].sCb1 = (long)n * h; ax[0].bias = Dv + e->ob1; ax[0].sbias_b1 = P; ax[0].mask = e->H1s;
        ax[0].ldmask = h;
        if (drop) ax[0].mask = nullptr;      // (masked per class right behind the launch)
        ax[1] = mk(dHp, h, 1, Dv + e->oW1, 1, h, e->slabs_x, d, n, d, h);
        ax[1].nb1 = Cl; ax[1].sAb1 = (long)n * h; ax[1].sBb1 = P; ax[1].sCb1 = (long)n * d;
        GS_TRY(gs_gemm_group_f32(ax, 2, stream));          // independent of each other
        if (drop) {
          hipLaunchKernelGGL(gs_drop_mask_scale_kernel, dim3(gs_cdiv((long)Cl * n * h, 256)), dim3(256), 0, s, A1m, H1c, (long)Cl * n * h,
                             e->drop_q);
          GS_LAUNCH_CHECK();
        }
        GS_TRY(gs_reduce_slabs_f32(e->d_feat, e->slabs_x, (long)n * d, Cl, (long)n * d, 0.f, stream));
        gs_gemm_desc b = mk(A1m, h, 1, Wl, C, 1, Bc, C, n, C, h);
        b.nb1 = Cl; b.sAb1 = (long)n * h; b.sCb1 = (long)n * C;
        GS_TRY(gs_gemm_f32(&b, stream));
      }
      // dQ[lo:hi] = G[lo:hi] Bc_c^T
      gs_gemm_desc qu[2];
      qu[0] = mk(e->Gs, C, 1, Bc, 1, C, e->dQ, n, mx, n, C);
      qu[0].nb1 = Cl; qu[0].offA = off_loC; qu[0].Mb = ncls; qu[0].sBb1 = (long)n * C; qu[0].offC = off_lon;
      // Ug[lo:hi] = Q[lo:hi] Bc_c + S[lo:hi] D2_c
      qu[1] = mk(e->Q, n, 1, Bc, C, 1, e->Ug, C, mx, C, n);
      qu[1].nb1 = Cl; qu[1].offA = off_lon; qu[1].Mb = ncls; qu[1].sBb1 = (long)n * C; qu[1].offC = off_loC;
      gs_gemm_desc u2 = mk(e->Ss, F, 1, Dl, C, 1, e->Ug, C, mx, C, F);
      u2.nb1 = Cl; u2.offA = off_loF; u2.Mb = ncls; u2.sBb1 = P; u2.offC = off_loC; u2.beta = 1.f;
      if (merge) {
        // one launch: dQ ; Ug = Q Bc followed by Ug += S D2 in the workgroup that owns the tile ; dS[lo:hi] = G Dl_c^T (reads only G
        // and D: it used to sit behind the V kernel) — bitwise the launch-per-product values (gs_gemm_group2_f32)
        gs_gemm_desc m3[3], s3[3];
        m3[0] = qu[0]; m3[1] = qu[1];
        m3[2] = mk(e->Gs, C, 1, Dl, 1, C, e->dSs, F, mx, F, C);
        m3[2].nb1 = Cl; m3[2].offA = off_loC; m3[2].Mb = ncls; m3[2].sBb1 = P; m3[2].offC = off_loF;
        memset(s3, 0, sizeof(s3));
        s3[1] = u2;
        if (e->tiles_ok) {      // only the M tiles that exist (class sizes differ by an order of magnitude)
          const int t0 = e->tile_start[e->c_lo], t1 = e->tile_start[e->c_hi];
          GS_TRY(gs_gemm_group2_tiles_f32(m3, s3, 3, e->tile_b + t0, e->tile_my + t0, t1 - t0, e->c_lo, stream));
        } else {
          GS_TRY(gs_gemm_group2_f32(m3, s3, 3, stream));
        }
        ds_done = true;
      } else {
        GS_TRY(gs_gemm_group_f32(qu, 2, stream));
        GS_TRY(gs_gemm_f32(&u2, stream));
      }
    } else {
      gs_gemm_desc u = mk(e->Ss, F, 1, Dl, C, 1, e->Ug, C, mx, C, F);
      u.nb1 = Cl; u.offA = off_loF; u.Mb = ncls; u.sBb1 = P; u.offC = off_loC;
      if (merge) {      // {Ug = S D2, dS[lo:hi] = G Dl_c^T}: independent, one launch
        gs_gemm_desc m2[2];
        m2[0] = u;
        m2[1] = mk(e->Gs, C, 1, Dl, 1, C, e->dSs, F, mx, F, C);
        m2[1].nb1 = Cl; m2[1].offA = off_loC; m2[1].Mb = ncls; m2[1].sBb1 = P; m2[1].offC = off_loF;
        GS_TRY(gs_gemm_group_f32(m2, 2, stream));
        ds_done = true;
      } else {
        GS_TRY(gs_gemm_f32(&u, stream));
      }
    }
  }
  hipLaunchKernelGGL(gs_syn_v_kernel, dim3(gs_cdiv(n, 4)), dim3(256), 0, s, e->Ps, e->Gs, e->Ug, e->rsq, e->D, P, e->obl, bl,
                     e->io.labels_syn, e->io.syn_lo, e->io.syn_hi, n, C, e->c_lo, e->c_hi, e->Vs, e->drs,
                     e->fold_finish ? (const float*)e->loss_c : (const float*)nullptr, C, e->loss_sum, e->loss_last, e->steps,
                     e->dist_shares ? (const double*)e->dist_part : (const double*)nullptr, e->dist_shares, e->loss_c);
  GS_LAUNCH_CHECK();
  {  // dS[lo:hi] = G[lo:hi] Dl_c^T ;  dS += V Wl^T
    if (Cl > 0 && !ds_done) {
      gs_gemm_desc a = mk(e->Gs, C, 1, Dl, 1, C, e->dSs, F, mx, F, C);
      a.nb1 = Cl; a.offA = off_loC; a.Mb = ncls; a.sBb1 = P; a.offC = off_loF;
      GS_TRY(gs_gemm_f32(&a, stream));
    }
    gs_gemm_desc b = mk(e->Vs, C, 1, Wl, 1, C, e->dSs, F, n, F, C);
    b.beta = 1.f;
    GS_TRY(gs_gemm_f32(&b, stream));
  }
  if (e->T == 1) {
    gs_gemm_desc ab1[2];
    ab1[0] = mk(e->Q, 1, n, e->dSs, d, 1, e->d_feat, d, n, d, n);   // dX = Q^T dS
    ab1[1] = mk(e->dSs, d, 1, X, 1, d, e->dQ, n, n, n, d);          // dQ = dS X^T
    if (e->blk_pos) {
      GS_TRY(block_apply(e, e->blk_X, true, e->dSs, d, e->d_feat, nullptr, 0, stream));
      GS_TRY(gs_gemm_f32(&ab1[1], stream));
    } else {
      GS_TRY(gs_gemm_group_f32(ab1, 2, stream));
    }
  } else {
    gs_gemm_desc ab2[2];
    ab2[0] = mk(e->Q, 1, n, e->dSs, h, 1, e->dH1m, h, n, h, n);     // (Q^T dS) * mask
    ab2[0].mask = e->H1s; ab2[0].ldmask = h;
    ab2[1] = mk(e->dSs, h, 1, e->H1s, 1, h, e->dQ, n, n, n, h);     // dQ += dS H1^T
    ab2[1].beta = 1.f;
    if (drop) {      // class by class: dQ[lo:hi] += dS[lo:hi] H1c[c]^T ; dH1m = sum_c (Q[lo:hi]^T dS[lo:hi]) * [H1c[c] > 0] / (1 - p)
      float* tmp = e->A1m + c0 * n * h;      // (the A1m_c buffers have been consumed)
      if (Cl > 0) {
        gs_gemm_desc dq = mk(e->dSs, h, 1, H1c, 1, h, e->dQ, n, mx, n, h);
        dq.nb1 = Cl; dq.offA = off_loF; dq.Mb = ncls; dq.sBb1 = (long)n * h; dq.offC = off_lon; dq.beta = 1.f;
        GS_TRY(gs_gemm_f32(&dq, stream));
      }
      if (!e->skip_dfeat) {
        if (Cl > 0) {
          gs_gemm_desc t = mk(e->Q, 1, n, e->dSs, h, 1, tmp, h, n, h, mx);
          t.nb1 = Cl; t.offA = off_lon; t.offB = off_loF; t.Kb = ncls; t.sCb1 = (long)n * h;
          GS_TRY(gs_gemm_f32(&t, stream));
        }
        hipLaunchKernelGGL(gs_drop_mask_reduce_kernel, dim3(gs_cdiv((long)n * h, 256)), dim3(256), 0, s, tmp, H1c, (long)n * h, Cl, e->drop_q,
                           e->dH1m);
        GS_LAUNCH_CHECK();
        gs_gemm_desc c = mk(e->dH1m, h, 1, th + e->oW1, 1, h, e->d_feat, d, n, d, h);   // dX += dH1m W1^T
        c.beta = 1.f;
        GS_TRY(gs_gemm_f32(&c, stream));
      }
    } else
    if (e->skip_dfeat) {      // (PGE epochs of the fused step: d loss / d X~ is not consumed — only dQ is needed from here on)
      GS_TRY(gs_gemm_f32(&ab2[1], stream));
    } else {
    if (e->blk_pos) {
      GS_TRY(block_apply(e, e->blk_X, true, e->dSs, h, e->dH1m, e->H1s, h, stream));
      GS_TRY(gs_gemm_f32(&ab2[1], stream));
    } else {
      GS_TRY(gs_gemm_group_f32(ab2, 2, stream));
    }
    gs_gemm_desc c = mk(e->dH1m, h, 1, th + e->oW1, 1, h, e->d_feat, d, n, d, h);   // dX += dH1m W1^T
    c.beta = 1.f;
    GS_TRY(gs_gemm_f32(&c, stream));
    }
  }
  hipLaunchKernelGGL(gs_add_rowvec_kernel, dim3(gs_cdiv((long)n * n, 256)), dim3(256), 0, s, e->dQ, e->drs, n, n);
  GS_LAUNCH_CHECK();
  if (e->blk_pos) {
    GS_TRY(block_dadj(e, A, e->dQ, e->dAn, stream));
  } else if (e->K != 2) {
    GS_TRY(adj_powers_bwd(e, A, e->dQ, e->dAn, stream));
  } else {  // dA^ = dQ A^T + A^T dQ ; then through dense_gcn_norm (both linear in dQ: partial sums stay partial sums)
    gs_gemm_desc a = mk(e->dQ, n, 1, A, 1, n, e->dAn, n, n, n, n);
    gs_gemm_desc b = mk(A, 1, n, e->dQ, n, 1, e->dAn, n, n, n, n);
    b.beta = 1.f;
    if (merge) {
      GS_TRY(gs_gemm_group2_f32(&a, &b, 1, stream));
    } else {
      GS_TRY(gs_gemm_f32(&a, stream));
      GS_TRY(gs_gemm_f32(&b, stream));
    }
  }
  if (e->identity) return GS_OK;     // adj_syn is the constant identity: nothing upstream of A^
  return gs_dense_gcn_norm_bwd(e->adj_raw, e->rnorm, e->dAn, n, e->d_adj, stream);
}

extern "C" int gs_gcond_stage_syn_finish(gs_gcond* e, gs_stream_t stream) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  hipLaunchKernelGGL(gs_loss_accum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, e->loss_c, e->C, e->loss_sum, e->loss_last,
                     e->steps);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// ================================================================================================
// stage: PGE backward: d adj -> d theta_PGE and d X~ (added to d_feat), in three parts; ranks exchange
// (1 -> 2) the BN-2 backward column sums "ar_l3" and (2 -> 3) the marginals "ar_marg" (+ "ar_w1" in PGE epochs).
// ================================================================================================
extern "C" int gs_gcond_stage_pge_backward_part(gs_gcond* e, int need_param_grads, int need_feat_grad, int part,
                                                gs_stream_t stream) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  GS_TRY(quiesce_hoist(e));
  hipStream_t s = (hipStream_t)stream;
  const int n = e->n, d = e->d, Ph = e->Ph, nr = e->r_hi - e->r_lo;
  const long n2 = (long)n * n, rows = (long)nr * n;
  const float* X = e->io.feat_syn;
  const float* pp = e->io.pge_params;
  float* g = e->d_pge;
  const float* de3l = e->de3 + (long)e->r_lo * n;
  const int nblk = gs_cdiv(rows > 0 ? rows : 1, e->rows_per_block);
  if (e->pge_chunks > 1 && part != 3) return part == 1 ? pge_backward_chunked(e, need_param_grads, stream) : GS_OK;
  if (part == 1) {
    hipLaunchKernelGGL(gs_pge_adj_bwd_kernel, dim3(gs_cdiv(n2, 256)), dim3(256), 0, s, e->d_adj, e->adj_raw, n, e->de3);
    if (rows > 0) {
      if (Ph == 256)
        hipLaunchKernelGGL((gs_pge_l3_bwd_stats_v4_kernel<64>), dim3(nblk), dim3(256), 0, s, e->E2, de3l, e->mu2, e->is2, pp + e->pg2,
                           pp + e->pbe2, pp + e->pw2, rows, e->rows_per_block, e->parts);
      else if (Ph == 128)
        hipLaunchKernelGGL((gs_pge_l3_bwd_stats_v4_kernel<32>), dim3(nblk), dim3(256), 0, s, e->E2, de3l, e->mu2, e->is2, pp + e->pg2,
                           pp + e->pbe2, pp + e->pw2, rows, e->rows_per_block, e->parts);
      else
        hipLaunchKernelGGL(gs_pge_l3_bwd_stats_kernel, dim3(nblk), dim3(256), 0, s, e->E2, de3l, e->mu2, e->is2, pp + e->pg2,
                           pp + e->pbe2, pp + e->pw2, rows, Ph, e->rows_per_block, e->parts);
      // (unsharded + fused: both reduction stages ride in part 2's prep launch — gs_pge_bwd_prep_fold_kernel)
      e->l3_direct = nblk <= 256 && knob_of(e, GS_KNOB_FUSED_CHAINS, g_fused_chains) && knob_of(e, GS_KNOB_FOLDED_REDUCTIONS, g_defer_last_stage) && !sharded_r(e);      // (few partial rows only: 3 221 at n' = 454 took the 8 workgroups 100 us)
      if (e->l3_direct) e->l3_slices = 1;
      else
      e->l3_slices = reduce_parts(e, e->parts, nblk, 3, 3, s, knob_of(e, GS_KNOB_FOLDED_REDUCTIONS, g_defer_last_stage) && !sharded_r(e));
    } else {
      GS_HIP(hipMemsetAsync(e->sums, 0, sizeof(double) * 3 * Ph, s));
      e->l3_slices = 1;
      e->l3_direct = false;
    }
    GS_LAUNCH_CHECK();
    return GS_OK;
  }
  if (part == 2) {
    // one launch: the layer-3 statistics as floats (dw2, dbe2, dg2), the constants table of the generated dE2 operand where
    // that form applies (feature epochs), and W1^T
    const bool want_gen = (!need_param_grads || knob_of(e, GS_KNOB_DW1_GENERATED, g_dw1_generated)) && Ph <= 256 && rows > 0;
    // (mode 3: the split image of W1^T for the marginals product comes out of the prep launch)
    const bool pre_split = knob_of(e, GS_KNOB_FUSED_CHAINS, g_fused_chains) && e->operand_mode == 3 && rows > 0 && (Ph & 31) == 0;
    if (e->l3_direct && rows > 0) {
      const int nt = gs_cdiv(Ph, 32);
      hipLaunchKernelGGL(gs_pge_bwd_prep_fold_kernel, dim3(nt + nt * nt), dim3(1024), 0, s, (const double*)e->parts, nblk, nblk > 64 ? 32 : 1, Ph,
                         g + e->pw2, g + e->pbe2, g + e->pg2, e->mu2, e->is2, pp + e->pg2, pp + e->pbe2, pp + e->pw2, 1.f / (float)n2,
                         want_gen ? e->gen_tab : (float*)nullptr, pp + e->pW1, e->W1T, pre_split ? (__bf16*)e->bsplit_main : (__bf16*)nullptr);
    } else
    hipLaunchKernelGGL(gs_pge_bwd_prep_kernel, dim3(gs_cdiv(Ph, 32), gs_cdiv(Ph, 32)), dim3(32, 8), 0, s,
                       e->l3_slices > 1 ? e->sums2 : e->sums, e->l3_slices, Ph, g + e->pw2, g + e->pbe2, g + e->pg2, e->mu2, e->is2,
                       pp + e->pg2, pp + e->pbe2, pp + e->pw2, 1.f / (float)n2, want_gen ? e->gen_tab : (float*)nullptr, pp + e->pW1,
                       e->W1T, pre_split ? (__bf16*)e->bsplit_main : (__bf16*)nullptr);
    if (sharded_r(e)) GS_HIP(hipMemsetAsync(e->ar_marg, 0, sizeof(float) * 2 * (long)n * Ph, s));
    if (need_param_grads) {
      GS_TRY(sum_f32(e, e->de3, n2, g + e->pb2, s));   // db2 = sum de3 (replicated)
      if (rows == 0) GS_HIP(hipMemsetAsync(g + e->pW1, 0, sizeof(float) * ((long)Ph * Ph + Ph), s));
    }
    GS_LAUNCH_CHECK();
    if (rows > 0) {
      // Feature epochs need dE2 only as the first operand of the marginals product: the row-panel kernel then forms it in
      // registers from E2 (generated A operand) and the 422 MB bn2_bwd pass is skipped.  PGE epochs (dW1, db1 read dE2 too)
      // and shapes the generated form does not cover materialise it.
      const int nrb0 = gs_cdiv(nr, 16), ncb0 = gs_cdiv(n, 8);
      bool fused_done = false;
      if (want_gen) {
        gs_panel_gen gen = {e->mu2, e->is2, pp + e->pg2, pp + e->pbe2, pp + e->pw2, g + e->pbe2, g + e->pg2, de3l, (long)n2, e->gen_tab, 1};
        float* partV = e->dY1;
        float* partU = e->dY1 + (size_t)ncb0 * nr * Ph;
        gs_panel_a1 a1 = pge_a1(e);
        a1.b_ready = pre_split ? 1 : 0;
        const int rc = gs_panel_marginals(e->E2, Ph, e->W1T, Ph, e->a1_valid ? e->A1 : nullptr, Ph, n, nr, Ph, Ph, partU, partV,
                                          e->operand_mode, e->bsplit_main, &gen, &a1, s);
        if (rc == GS_OK) {
          GS_TRY(gs_reduce_slabs_pair(e->rowsum_c, partU, (long)n * Ph, nrb0, (long)n * Ph, e->colsum_r + (long)e->r_lo * Ph, partV, (long)nr * Ph,
                                      ncb0, (long)nr * Ph, stream));      // both marginals' partial sums: one launch
          fused_done = true;
        } else if (rc != GS_ERR_UNSUPPORTED) {
          return rc;
        }
      }
      if (fused_done && need_param_grads) {
        // PGE epochs: dW1 = dE2^T A1 and db1 = colsum(dE2) from the same generated operands (gs_edge.hip) — the dE2 buffer is
        // free (nothing materialised): it holds the folded layer-1 terms and the slabs
        float* uv = e->dE2;
        float* slabs = e->dE2 + 2L * n * Ph;
        int nsplit = 0;
        const int rc = gs_dw1_generated(e->E2, de3l, e->gen_tab, e->U, e->V, e->a1_tab, n, e->r_lo, nr, Ph, e->operand_mode, uv, slabs,
                                        n2 * Ph - 2L * n * Ph, &nsplit, e->uvs_valid ? e->UsVs : nullptr, s);
        if (rc == GS_OK) {
          GS_TRY(gs_reduce_slabs_f32(g + e->pW1, slabs, (long)Ph * Ph + Ph, nsplit, 65536 + 256, 0.f, stream));   // dW1 | db1
          return GS_OK;
        }
        if (rc != GS_ERR_UNSUPPORTED) return rc;
      } else if (fused_done) {
        return GS_OK;
      }
      hipLaunchKernelGGL(gs_pge_bn2_bwd_v4_kernel, dim3(gs_cdiv(gs_cdiv(rows, 8) * (long)(Ph / 4), 256)), dim3(256), 0, s, e->E2, de3l, e->mu2,
                         e->is2, pp + e->pg2, pp + e->pbe2, pp + e->pw2, g + e->pbe2, g + e->pg2, rows, n2, Ph, e->dE2);
      GS_LAUNCH_CHECK();
      if (need_param_grads) {
        // dW1 = dE2^T A1  (K = local pair rows split into slabs, fixed-order reduce) ; db1 = colsum(dE2)   [partials]
        GS_TRY(ensure_a1(e, s));
        const int nsplit = gs_cdiv(rows, e->kchunk_w);
        gs_gemm_desc w = mk(e->dE2, 1, Ph, e->A1, Ph, 1, e->slabs_w, Ph, Ph, Ph, (int)rows);
        w.nb2 = nsplit; w.kchunk = e->kchunk_w; w.sCb2 = (long)Ph * Ph;
        GS_TRY(gs_gemm_f32(&w, stream));
        GS_TRY(gs_reduce_slabs_f32(g + e->pW1, e->slabs_w, (long)Ph * Ph, nsplit, (long)Ph * Ph, 0.f, stream));
        hipLaunchKernelGGL(gs_col_stats_kernel, dim3(nblk), dim3(256), 0, s, e->dE2, rows, Ph, e->rows_per_block, e->parts);
        reduce_parts(e, e->parts, nblk, 2, 1, s);
        hipLaunchKernelGGL(gs_sums_to_float_kernel, dim3(gs_cdiv(Ph, 64)), dim3(64), 0, s, e->sums, 1, Ph, g + e->pb1,
                           (float*)nullptr, (float*)nullptr);
        GS_LAUNCH_CHECK();
      }
      if (!fused_done) {  // dY1 = (dE2 W1) * [A1 > 0] is needed only through its two marginals (W1^T from the prep launch above: both
         // operands k-contiguous).  Preferred route: the row-panel kernel's fused-marginals form (pair-patch tiles,
         // gs_panel.hip) — dY1 is never materialised, its buffer holds the partial sums instead.
        const int nrb = gs_cdiv(nr, 16), ncb = gs_cdiv(n, 8);
        float* partV = e->dY1;                                   // [ncb][nr][Ph]
        float* partU = e->dY1 + (size_t)ncb * nr * Ph;           // [nrb][n][Ph]
        int rc = GS_ERR_UNSUPPORTED;
        if (((size_t)ncb * nr + (size_t)nrb * n) * Ph <= (size_t)n2 * Ph) {
          gs_panel_a1 a1 = pge_a1(e);
          a1.b_ready = pre_split ? 1 : 0;
          rc = gs_panel_marginals(e->dE2, Ph, e->W1T, Ph, e->a1_valid ? e->A1 : nullptr, Ph, n, nr, Ph, Ph, partU, partV,
                                  e->operand_mode, e->bsplit_main, nullptr, &a1, s);
        }
        if (rc == GS_OK) {
          GS_TRY(gs_reduce_slabs_f32(e->rowsum_c, partU, (long)n * Ph, nrb, (long)n * Ph, 0.f, stream));
          GS_TRY(gs_reduce_slabs_f32(e->colsum_r + (long)e->r_lo * Ph, partV, (long)nr * Ph, ncb, (long)nr * Ph, 0.f, stream));
        } else if (rc != GS_ERR_UNSUPPORTED) {
          return rc;
        } else {
          GS_TRY(ensure_a1(e, s));
          gs_gemm_desc a = mk(e->dE2, Ph, 1, e->W1T, 1, Ph, e->dY1, Ph, (int)rows, Ph, Ph);
          a.mask = e->A1; a.ldmask = Ph;
          a.prec = e->operand_mode; a.scratch = e->bsplit_main;
          GS_TRY(gs_gemm_f32(&a, stream));
          // both marginals in one pass over dY1; the partial sums live in `parts` (free here: its producers ran earlier on
          // this stream), reduced in a fixed order
          const int nrb2 = gs_cdiv(nr, GS_MARG_RR), ncb2 = gs_cdiv(n, GS_MARG_CC);
          const size_t need = ((size_t)nrb2 * n + (size_t)ncb2 * nr) * Ph * sizeof(float);
          const size_t have = (size_t)e->nblk_stats * 3 * Ph * sizeof(double);
          if (Ph % 4 == 0 && Ph <= 1024 && need <= have) {
            float* pU = reinterpret_cast<float*>(e->parts);
            float* pV = pU + (size_t)nrb2 * n * Ph;
            hipLaunchKernelGGL(gs_pge_marginals_fused_kernel, dim3(ncb2, nrb2), dim3(256), 0, s, e->dY1, n, Ph, nr, pU, pV);
            GS_LAUNCH_CHECK();
            GS_TRY(gs_reduce_slabs_f32(e->rowsum_c, pU, (long)n * Ph, nrb2, (long)n * Ph, 0.f, stream));
            GS_TRY(gs_reduce_slabs_f32(e->colsum_r + (long)e->r_lo * Ph, pV, (long)nr * Ph, ncb2, (long)nr * Ph, 0.f, stream));
          } else {
            hipLaunchKernelGGL(gs_pge_marginal_r_kernel, dim3(nr), dim3(256), 0, s, e->dY1, n, Ph, e->r_lo, e->colsum_r);
            hipLaunchKernelGGL(gs_pge_marginal_c_kernel, dim3(n), dim3(256), 0, s, e->dY1, n, Ph, nr, e->rowsum_c);
            GS_LAUNCH_CHECK();
          }
        }
      }
    }
    return GS_OK;
  }
  if (part != 3) GS_FAIL(GS_ERR_ARG, "gs_gcond_stage_pge_backward_part: part must be 1, 2 or 3");
  if (e->pge_chunks == 1) {   // (the chunked backward has already left dU, dV and the BN-1 parameter gradients)
  if (knob_of(e, GS_KNOB_FUSED_CHAINS, g_fused_chains)) {      // dU | dV out of the same launch (bitwise the two-launch values)
    hipLaunchKernelGGL(gs_pge_bn1_bwd_cols_kernel, dim3(gs_cdiv(Ph, 32)), dim3(1024), 0, s, e->U, e->V, e->rowsum_c, e->colsum_r,
                       pp + e->pb0, e->mu1, e->is1, n, Ph, g + e->pg1, g + e->pbe1, pp + e->pg1, (const float*)e->sumU, (const float*)e->sumV,
                       e->dU, e->dV);
  } else {
  hipLaunchKernelGGL(gs_pge_bn1_bwd_cols_kernel, dim3(gs_cdiv(Ph, 32)), dim3(1024), 0, s, e->U, e->V, e->rowsum_c, e->colsum_r,
                     pp + e->pb0, e->mu1, e->is1, n, Ph, g + e->pg1, g + e->pbe1);
  hipLaunchKernelGGL(gs_pge_bn1_bwd_uv_kernel, dim3(gs_cdiv((long)n * Ph, 256)), dim3(256), 0, s, e->U, e->V, e->rowsum_c,
                     e->colsum_r, pp + e->pb0, e->mu1, e->is1, pp + e->pg1, e->sumU, e->sumV, g + e->pg1, g + e->pbe1, n, Ph,
                     e->dU, e->dV);
  }
  }
  GS_LAUNCH_CHECK();
  if (need_param_grads) {
    hipLaunchKernelGGL(gs_colsum_small_kernel, dim3(gs_cdiv(Ph, 64)), dim3(256), 0, s, e->dU, n, Ph, g + e->pb0);
    GS_LAUNCH_CHECK();
    gs_gemm_desc w0[2];
    w0[0] = mk(e->dU, 1, Ph, X, d, 1, g + e->pW0, 2L * d, Ph, d, n);       // dW0[:, :d] = dU^T X
    w0[1] = mk(e->dV, 1, Ph, X, d, 1, g + e->pW0 + d, 2L * d, Ph, d, n);   // dW0[:, d:] = dV^T X
    GS_TRY(gs_gemm_group_f32(w0, 2, stream));
  }
  if (need_feat_grad) {   // d_feat += dU W0[:, :d] + dV W0[:, d:]
    gs_gemm_desc a = mk(e->dU, Ph, 1, pp + e->pW0, 2L * d, 1, e->d_feat, d, n, d, Ph);
    a.beta = 1.f;
    gs_gemm_desc b = mk(e->dV, Ph, 1, pp + e->pW0 + d, 2L * d, 1, e->d_feat, d, n, d, Ph);
    b.beta = 1.f;
    if (knob_of(e, GS_KNOB_FUSED_CHAINS, g_fused_chains)) {      // both products by the workgroup that owns the tile: one launch, same bits
      GS_TRY(gs_gemm_group2_f32(&a, &b, 1, stream));
    } else {
      GS_TRY(gs_gemm_f32(&a, stream));
      GS_TRY(gs_gemm_f32(&b, stream));
    }
  }
  return GS_OK;
}

extern "C" int gs_gcond_stage_pge_backward(gs_gcond* e, int need_param_grads, int need_feat_grad, gs_stream_t stream) {
  if (e && sharded_r(e)) GS_FAIL(GS_ERR_STATE, "gs_gcond_stage_pge_backward: sharded engines must run the _part entry points");
  for (int part = 1; part <= 3; ++part) GS_TRY(gs_gcond_stage_pge_backward_part(e, need_param_grads, need_feat_grad, part, stream));
  return GS_OK;
}

// ================================================================================================
// stage: optimiser step (gcond.py:58-61: PGE for epoch % 50 < 10, else X~)
// ================================================================================================
extern "C" int gs_gcond_stage_update(gs_gcond* e, int epoch, gs_stream_t stream) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  GS_TRY(quiesce_hoist(e));
  if (epoch % 50 < 10)
    return gs_adam_step_f32(e->io.pge_params, e->d_pge, e->m_pge, e->v_pge, e->Ppge, e->cfg.lr_adj, 0.9f, 0.999f, 1e-8f, 0.f,
                            ++e->step_pge, stream);
  return gs_adam_step_f32(e->io.feat_syn, e->d_feat, e->m_feat, e->v_feat, (long)e->n * e->d, e->cfg.lr_feat, 0.9f, 0.999f,
                          1e-8f, 0.f, ++e->step_feat, stream);
}

// ================================================================================================
// stage: inner_loop SGC steps on (X~, A^) after PGE.inference  (gcond.py:63-72).  Replicated on every rank.
// ================================================================================================
// measurement / test knob: 1 (default) = the fused launch chains of gs_chain.hip (inner SGC steps: logits + softmax + NLL gradient in
// one launch, products re-associated through the C-wide operand, Adam in the gradient products' epilogue; synthetic match: logits +
// softmax in one launch); 0 = one launch per product / element-wise pass (the first form)
extern "C" int gs_gcond_set_edge_src(int on) {
  g_edge_src = on != 0;
  return GS_OK;
}
extern "C" int gs_gcond_set_inner_persist(int on) {
  g_inner_persist = on < 0 ? 0 : (on > 2 ? 2 : on);
  return GS_OK;
}
extern "C" int gs_gcond_set_fused_chains(int on) {
  g_fused_chains = on != 0;
  return GS_OK;
}

static gs_adam_ep model_adam(gs_gcond* e) {
  const int step = ++e->step_model;
  const double bc1 = 1.0 - pow(0.9, step), bc2 = 1.0 - pow(0.999, step);
  gs_adam_ep ep;
  ep.theta = e->io.model_params; ep.m = e->m_model; ep.v = e->v_model; ep.gbase = e->g_model;
  ep.step_size = (float)((double)e->cfg.lr_model / bc1); ep.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  ep.b1 = 0.9f; ep.b2 = 0.999f; ep.eps = 1e-8f;
  return ep;
}

// The inner SGC steps as fused launches (gs_chain.hip; algebra in its header).  SGC without BatchNorm / block structure, C <= 64.
static int inner_steps_fused(gs_gcond* e, gs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const int C = e->C, n = e->n, d = e->d, h = e->h, F = e->F;
  const float* X = e->io.feat_syn;
  const float* th = e->io.model_params;
  const float* A = e->adj_norm_in;
  float* g = e->g_model;
  // knob: 0 = the launch chain, 2 = always the persistent launch, 1 (default) = the persistent launch unless a hoisted training forward
  // is in flight beside this stage (measured, profiles/r06: alone on the chip the one launch wins — 1 329 against 1 344 us per iteration
  // with forward reuse — beside the forward's persistent grid its 48 resident workgroups lose to the chain's short launches: 1 458 / 1 416)
  const int ipk = knob_of(e, GS_KNOB_INNER_PERSIST, g_inner_persist);
  if (e->T == 2 && e->K == 2 && e->ip_ws && !e->drop_thr && (ipk == 2 || (ipk == 1 && !e->hoist_pending)) &&
      gs_inner_persist_eligible(n, d, h, C, e->cfg.inner_loop)) {
    // ONE persistent launch for all inner steps (gs_inner.hip): Q = A^ A^ and its transpose land in the launch's padded operands
    gs_gemm_desc q2[2];
    q2[0] = mk(A, n, 1, A, n, 1, e->ipw.Q, e->ipw.npad, n, n, n);
    q2[1] = mk(A, 1, n, A, 1, n, e->ipw.QT, e->ipw.npad, n, n, n);       // Q^T = A^^T A^^T
    GS_TRY(gs_gemm_group_f32(q2, 2, stream));
    float ss[16], isb[16];
    for (int it = 0; it < e->cfg.inner_loop; ++it) {
      const gs_adam_ep ep = model_adam(e);
      ss[it] = ep.step_size; isb[it] = ep.inv_sqrt_bc2;
    }
    return gs_inner_persist_launch(e->ipw, n, d, h, C, e->cfg.inner_loop, X, d, e->io.labels_syn, e->io.model_params, e->m_model, e->v_model, g,
                                   e->oW1, e->ob1, e->oWl, e->obl, ss, isb, 0.9f, 0.999f, 1e-8f, &e->ip_counter, s);
  }
  {
    if (e->K != 2) {
      GS_TRY(adj_powers(e, A, e->Qin, stream));      // (the training powers are not needed any more: the backward has run)
    } else {
      gs_gemm_desc q = mk(A, n, 1, A, n, 1, e->Qin, n, n, n, n);
      GS_TRY(gs_gemm_f32(&q, stream));
    }
    hipLaunchKernelGGL(gs_rowsum_kernel, dim3(gs_cdiv(n, 4)), dim3(256), 0, s, e->Qin, n, n, e->rsq_in);
    GS_LAUNCH_CHECK();
    if (e->T == 1) {   // S = Q X does not depend on theta: once per outer iteration
      gs_gemm_desc sg = mk(e->Qin, n, 1, X, d, 1, e->S_in, d, n, d, n);
      GS_TRY(gs_gemm_f32(&sg, stream));
    }
  }
  gs_zg_args zg;
  zg.n = n; zg.C = C; zg.rs = e->rsq_in; zg.bias = th + e->obl; zg.labels = e->io.labels_syn; zg.syn_lo = e->io.syn_lo;
  zg.syn_hi = e->io.syn_hi; zg.divide_by_n = 1; zg.P = nullptr; zg.G = e->Gs;
  for (int it = 0; it < e->cfg.inner_loop; ++it) {
    if (e->T == 1) {
      zg.A = e->S_in; zg.sAm = d; zg.sAk = 1; zg.B = th + e->oWl; zg.sBk = C; zg.sBn = 1; zg.K = d;
      GS_TRY(gs_chain_zg(&zg, s));                                                       // Z = S Wl + rs (x) bl ; G
      gs_gemm_desc g2[2];
      g2[0] = mk(e->S_in, 1, F, e->Gs, C, 1, g + e->oWl, C, F, C, n);                    // gWl = S^T G
      g2[1] = mk(e->rsq_in, 0, 1, e->Gs, C, 1, g + e->obl, C, 1, C, n);                  // gbl = rs^T G
      const gs_adam_ep ep = model_adam(e);
      GS_TRY(gs_chain_group(g2, 2, &ep, s));                                             // (+ Adam on Wl, bl)
      continue;
    }
    float* Tm = e->Zs;      // [n x C] T = H1 Wl
    float* R = e->Ug;       // [n x C] R = Q^T G   (the matching stage's Ug: free here)
    gs_gemm_desc a = mk(X, d, 1, th + e->oW1, h, 1, e->H1s, h, n, h, d);
    a.bias = th + e->ob1; a.relu = 1;
    GS_TRY(gs_gemm_f32(&a, stream));                                                     // H1 = relu(X W1 + b1)
    if (e->drop_thr) {      // models/sgc.py:42 in the inner step's forward (gcond.py:67-72: the model is in training mode): a fresh mask per step
      hipLaunchKernelGGL(gs_drop_rows_kernel, dim3(gs_cdiv((long)n * (h / 4), 256)), dim3(256), 0, s, e->H1s, e->H1s, n, h, 1,
                         gs_drop_key0(e->cfg.seed, e->drop_iter, 2), it, e->drop_thr, e->drop_q);
      GS_LAUNCH_CHECK();
    }
    gs_gemm_desc t = mk(e->H1s, h, 1, th + e->oWl, C, 1, Tm, C, n, C, h);
    GS_TRY(gs_gemm_f32(&t, stream));                                                     // T = H1 Wl
    // (H1 -> T and R -> dH1 chained through the accumulator registers as in gs_match.hip were measured: 15 workgroups carry each
    // pair, 15.1 / 13.6 us against 5.0 + 5.4 / 8.0 + 4.8 us for the launches of 120 / 30 tiles — profiles/r04/inner_step_chained_pairs_not_kept.txt)
    // (logits and softmax stay two launches here: fused, the 15 row-block workgroups of n' = 454 carry the whole K = n' product —
    // 14.6 us against 6.6 + 4.4 for the 30-tile product and the wave-per-row softmax)
    float* Zl = e->Ps;      // [n x C] logits (the matching stage's softmax buffer: free here)
    gs_gemm_desc z = mk(e->Qin, n, 1, Tm, C, 1, Zl, C, n, C, n);
    GS_TRY(gs_gemm_f32(&z, stream));                                                     // Z = Q T
    hipLaunchKernelGGL(gs_syn_softmax_grad_kernel, dim3(gs_cdiv(n, 4)), dim3(256), 0, s, Zl, e->rsq_in, th + e->obl, e->io.labels_syn,
                       e->io.syn_lo, e->io.syn_hi, n, C, (float*)nullptr, e->Gs, 1);   // + rs (x) bl ; G
    GS_LAUNCH_CHECK();
    gs_gemm_desc r = mk(e->Qin, 1, n, e->Gs, C, 1, R, C, n, C, n);
    GS_TRY(gs_gemm_f32(&r, stream));                                                     // R = Q^T G
    gs_gemm_desc dh = mk(R, C, 1, th + e->oWl, 1, C, e->dH1m, h, n, h, C);
    dh.mask = e->H1s; dh.ldmask = h; dh.alpha = e->drop_q;      // ([H1 > 0] covers relu and the mask; 1 / (1 - p) or 1)
    GS_TRY(gs_gemm_f32(&dh, stream));                                                    // dH1 = (R Wl^T) * [H1 > 0]
    gs_gemm_desc g4[4];
    g4[0] = mk(X, 1, d, e->dH1m, h, 1, g + e->oW1, h, d, h, n);                          // gW1 = X^T dH1
    g4[1] = mk(e->ones, 0, 1, e->dH1m, h, 1, g + e->ob1, h, 1, h, n);                    // gb1 = 1^T dH1
    g4[2] = mk(e->H1s, 1, h, R, C, 1, g + e->oWl, C, h, C, n);                           // gWl = H1^T R
    g4[3] = mk(e->rsq_in, 0, 1, e->Gs, C, 1, g + e->obl, C, 1, C, n);                    // gbl = rs^T G
    const gs_adam_ep ep = model_adam(e);
    GS_TRY(gs_chain_group(g4, 4, &ep, s));                                               // (+ Adam on theta)
  }
  return GS_OK;
}

extern "C" int gs_gcond_stage_inner_steps(gs_gcond* e, gs_stream_t stream) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  hipStream_t s = (hipStream_t)stream;
  const int C = e->C, n = e->n, d = e->d, h = e->h, F = e->F;
  if (e->cfg.inner_loop == 0) return GS_OK;
  if (e->deep) return deep_inner_steps(e, stream);
  if (e->gk) return gcnk_inner_steps(e, stream);
  gs_blk_dense_guard blk_guard(e);
  if (knob_of(e, GS_KNOB_FUSED_CHAINS, g_fused_chains) && !e->gcn && !e->with_bn && !e->blk_pos && C <= 64) return inner_steps_fused(e, stream);
  const float* X = e->io.feat_syn;
  float* th_true = e->io.model_params;                 // Adam steps these ...
  const float* th = model_th(e);                       // ... the products read these (the BatchNorm-folded image under with_bn)
  const float* A = e->adj_norm_in;
  if (e->gcn) {   // oracle/analytic.py:gcn_dense_grad
    { gs_gemm_desc g = mk(A, n, 1, X, d, 1, e->S1, d, n, d, n); GS_TRY(gs_gemm_f32(&g, stream)); }      // S1 = A X: once per outer iteration
    for (int it = 0; it < e->cfg.inner_loop; ++it) {
      { gs_gemm_desc g = mk(e->S1, d, 1, th + e->oW1, h, 1, e->H1s, h, n, h, d); g.bias = th + e->ob1; g.relu = 1; GS_TRY(gs_gemm_f32(&g, stream)); }
      { gs_gemm_desc g = mk(A, n, 1, e->H1s, h, 1, e->S_in, h, n, h, n); GS_TRY(gs_gemm_f32(&g, stream)); }
      { gs_gemm_desc g = mk(e->S_in, h, 1, th + e->oWl, C, 1, e->Zs, C, n, C, h); GS_TRY(gs_gemm_f32(&g, stream)); }
      hipLaunchKernelGGL(gs_syn_softmax_grad_kernel, dim3(gs_cdiv(n, 4)), dim3(256), 0, s, e->Zs, e->ones, th + e->obl,
                         e->io.labels_syn, e->io.syn_lo, e->io.syn_hi, n, C, (float*)nullptr, e->Gs, 1);
      GS_LAUNCH_CHECK();
      {
        gs_gemm_desc g3[3];
        g3[0] = mk(e->S_in, 1, h, e->Gs, C, 1, e->g_model + e->oWl, C, h, C, n);         // gW2 = S2^T G
        g3[1] = mk(e->ones, 0, 1, e->Gs, C, 1, e->g_model + e->obl, C, 1, C, n);         // gb2 = 1^T G
        g3[2] = mk(e->Gs, C, 1, th + e->oWl, 1, C, e->dSs, h, n, h, C);                  // dS2 = G W2^T
        GS_TRY(gs_gemm_group_f32(g3, 3, stream));
      }
      { gs_gemm_desc g = mk(A, 1, n, e->dSs, h, 1, e->dH1m, h, n, h, n); g.mask = e->H1s; g.ldmask = h; GS_TRY(gs_gemm_f32(&g, stream)); }
      {
        gs_gemm_desc g2[2];
        g2[0] = mk(e->S1, 1, d, e->dH1m, h, 1, e->g_model + e->oW1, h, d, h, n);         // gW1 = S1^T dH1p
        g2[1] = mk(e->ones, 0, 1, e->dH1m, h, 1, e->g_model + e->ob1, h, 1, h, n);       // gb1 = 1^T dH1p
        GS_TRY(gs_gemm_group_f32(g2, 2, stream));
      }
      GS_TRY(gs_adam_step_f32(th_true, e->g_model, e->m_model, e->v_model, e->P, e->cfg.lr_model, 0.9f, 0.999f, 1e-8f, 0.f,
                              ++e->step_model, stream));
    }
    return GS_OK;
  }
  {
    if (e->blk_pos) {
      GS_TRY(block_square(e, A, e->Qin, stream));
    } else if (e->K != 2) {
      GS_TRY(adj_powers(e, A, e->Qin, stream));
    } else {
      gs_gemm_desc g = mk(A, n, 1, A, n, 1, e->Qin, n, n, n, n);
      GS_TRY(gs_gemm_f32(&g, stream));
    }
    hipLaunchKernelGGL(gs_rowsum_kernel, dim3(gs_cdiv(n, 4)), dim3(256), 0, s, e->Qin, n, n, e->rsq_in);
    GS_LAUNCH_CHECK();
    if (e->T == 1) {   // S = Q X does not depend on theta: once per outer iteration
      if (e->blk_pos) {
        GS_TRY(block_apply(e, e->blk_X, false, X, d, e->S_in, nullptr, 0, stream));
      } else {
        gs_gemm_desc sg = mk(e->Qin, n, 1, X, d, 1, e->S_in, d, n, d, n);
        GS_TRY(gs_gemm_f32(&sg, stream));
      }
    }
  }
  for (int it = 0; it < e->cfg.inner_loop; ++it) {
    if (e->T == 2) {
      gs_gemm_desc a = mk(X, d, 1, th + e->oW1, h, 1, e->H1s, h, n, h, d);
      a.bias = th + e->ob1; a.relu = 1;
      GS_TRY(gs_gemm_f32(&a, stream));
      if (e->drop_thr) {
        if (e->blk_pos) GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond: dropout != 0 with a block structure");
        hipLaunchKernelGGL(gs_drop_rows_kernel, dim3(gs_cdiv((long)n * (h / 4), 256)), dim3(256), 0, s, e->H1s, e->H1s, n, h, 1,
                           gs_drop_key0(e->cfg.seed, e->drop_iter, 2), it, e->drop_thr, e->drop_q);
        GS_LAUNCH_CHECK();
      }
      if (e->blk_pos) {
        GS_TRY(block_apply(e, e->blk_X, false, e->H1s, h, e->S_in, nullptr, 0, stream));
      } else {
        gs_gemm_desc b = mk(e->Qin, n, 1, e->H1s, h, 1, e->S_in, h, n, h, n);
        GS_TRY(gs_gemm_f32(&b, stream));
      }
    }
    gs_gemm_desc z = mk(e->S_in, F, 1, th + e->oWl, C, 1, e->Zs, C, n, C, F);
    GS_TRY(gs_gemm_f32(&z, stream));
    hipLaunchKernelGGL(gs_syn_softmax_grad_kernel, dim3(gs_cdiv(n, 4)), dim3(256), 0, s, e->Zs, e->rsq_in, th + e->obl,
                       e->io.labels_syn, e->io.syn_lo, e->io.syn_hi, n, C, (float*)nullptr, e->Gs, 1);
    GS_LAUNCH_CHECK();
    // three products that only read G share a launch, then the two that only read dHp (gs_gemm_group_f32)
    gs_gemm_desc g3[3];
    g3[0] = mk(e->S_in, 1, F, e->Gs, C, 1, e->g_model + e->oWl, C, F, C, n);             // gWl = S^T G
    g3[1] = mk(e->rsq_in, 0, 1, e->Gs, C, 1, e->g_model + e->obl, C, 1, C, n);           // gbl = rs^T G
    g3[2] = mk(e->Gs, C, 1, th + e->oWl, 1, C, e->dSs, h, n, h, C);                      // dS = G W2^T   (T = 2)
    GS_TRY(gs_gemm_group_f32(g3, e->T == 2 ? 3 : 2, stream));
    if (e->T == 2) {
      if (e->blk_pos) {
        GS_TRY(block_apply(e, e->blk_X, true, e->dSs, h, e->dH1m, e->H1s, h, stream));  // (Q^T dS) * mask
      } else {
        gs_gemm_desc dh = mk(e->Qin, 1, n, e->dSs, h, 1, e->dH1m, h, n, h, n);           // (Q^T dS) * mask
        dh.mask = e->H1s; dh.ldmask = h; dh.alpha = e->drop_q;
        GS_TRY(gs_gemm_f32(&dh, stream));
      }
      gs_gemm_desc g2[2];
      g2[0] = mk(X, 1, d, e->dH1m, h, 1, e->g_model + e->oW1, h, d, h, n);               // gW1 = X^T dHp
      g2[1] = mk(e->ones, 0, 1, e->dH1m, h, 1, e->g_model + e->ob1, h, 1, h, n);         // gb1 = 1^T dHp
      GS_TRY(gs_gemm_group_f32(g2, 2, stream));
    }
    if (e->with_bn) {      // gradients of the folded model -> gradients of [W1, b1, W2, b2, bn.weight, bn.bias]
      hipLaunchKernelGGL(gs_bn_unfold_grads_kernel, dim3(gs_cdiv(h, 32)), dim3(256), 0, s, e->g_model, th_true, e->dH1m, X, (long)d, n, d, h,
                         e->oW1, e->ob1, e->og, e->obeta);
      GS_LAUNCH_CHECK();
    }
    GS_TRY(gs_adam_step_f32(th_true, e->g_model, e->m_model, e->v_model, e->P, e->cfg.lr_model, 0.9f, 0.999f, 1e-8f, 0.f,
                            ++e->step_model, stream));
    GS_TRY(refresh_folded_model(e, s));
  }
  return GS_OK;
}

extern "C" int gs_gcond_stage_inner(gs_gcond* e, gs_stream_t stream) {
  GS_TRY(gs_gcond_stage_pge_forward(e, 1, stream));
  return gs_gcond_stage_inner_steps(e, stream);
}

__global__ void gs_eye_kernel(float* __restrict__ a, int n) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (long)n * n) a[idx] = (idx / n == idx % n) ? 1.f : 0.f;
}

extern "C" int gs_gcond_set_identity_structure(gs_gcond* e, int on, gs_stream_t stream) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  GS_TRY(quiesce_hoist(e));
  e->identity = on != 0;
  e->fwd_cached = false;
  if (e->identity) {   // gcondx.py:30 / doscondx.py:30: self.adj_syn = torch.eye(n) is used as is (not re-normalised)
    hipStream_t s = (hipStream_t)stream;
    const int n = e->n;
    dim3 g(gs_cdiv((long)n * n, 256));
    hipLaunchKernelGGL(gs_eye_kernel, g, dim3(256), 0, s, e->adj_norm, n);
    hipLaunchKernelGGL(gs_eye_kernel, g, dim3(256), 0, s, e->adj_norm_in, n);
    hipLaunchKernelGGL(gs_eye_kernel, g, dim3(256), 0, s, e->io.adj_syn, n);
    GS_LAUNCH_CHECK();
  }
  return GS_OK;
}

// The synthetic graph is B disconnected blocks of n_block rows (MSGC: batch_adj structures over the same nodes, laid out as one
// block-diagonal graph whose rows are grouped by class): pos [B][n_block] (device, int32) = row of node i of block b, a bijection
// onto [0, n_syn).  The three n_syn^3 products of an iteration (A^ A^ for train_class and for the inner steps, dQ A^^T + A^^T dQ)
// then run per block: 2 B n_block^3 FLOP each instead of 2 (B n_block)^3 — 767 -> 3 GFLOP at ogbn-arxiv r = 0.5 %, B = 16.
// So do the products of A^ A^ with [n_syn x F] matrices that do not go class by class (rows gathered per block).
// Values: the same sums with the structural zeros left out (other summation order).  scratch: 3 B n_block^2 + 2 n_syn max(d, hidden)
// floats, caller-owned;
// pos = NULL returns to dense products.  SGC only.
extern "C" int gs_gcond_set_block_structure(gs_gcond* e, const int* pos, int B, int n_block, float* scratch) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  GS_TRY(quiesce_hoist(e));
  if (!pos) { e->blk_pos = nullptr; e->blk_B = e->blk_n = 0; return GS_OK; }
  if (e->gcn) GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond_set_block_structure: built for the SGC condensation model");
  if (B < 1 || n_block < 1 || (long)B * n_block != e->n || !scratch) GS_FAIL(GS_ERR_ARG, "gs_gcond_set_block_structure: bad argument");
  e->blk_pos = pos; e->blk_B = B; e->blk_n = n_block;
  const size_t bs = (size_t)B * n_block * n_block;
  const size_t fs = (size_t)e->n * (size_t)(e->d > e->h ? e->d : e->h);
  e->blk_A = scratch; e->blk_X = scratch + bs; e->blk_Y = scratch + 2 * bs; e->blk_F1 = scratch + 3 * bs; e->blk_F2 = e->blk_F1 + fs;
  return GS_OK;
}

// Structure supplied by the caller (MSGC, condensation/msgc.py:98,109: the adjacency comes from an edge-list generator outside
// the engine, graphslim_amd/structure.py): `adj` [n'][n'] is the raw symmetric structure WITHOUT self loops.  inference = 0: it
// becomes the structure train_class differentiates — gs_gcond_stage_syn_match then leaves d loss / d adj in buffer "d_adj" and
// the direct feature gradient in "d_feat"; inference = 1: it becomes the structure of the inner model steps
// (gs_gcond_stage_inner_steps) and is copied to io.adj_syn.  The PGE stages are not run in this mode; the caller drives the
// iteration stage by stage and owns the optimiser steps of X~ and of its generator.
extern "C" int gs_gcond_set_structure(gs_gcond* e, const float* adj, int inference, gs_stream_t stream) {
  if (!e || !adj) GS_FAIL(GS_ERR_ARG, "gs_gcond_set_structure: null argument");
  GS_TRY(quiesce_hoist(e));
  if (e->identity) GS_FAIL(GS_ERR_STATE, "gs_gcond_set_structure: the engine is in identity-structure mode");
  hipStream_t s = (hipStream_t)stream;
  const size_t bytes = sizeof(float) * (size_t)e->n * e->n;
  e->fwd_cached = false;
  if (inference) {
    if (adj != e->io.adj_syn) GS_HIP(hipMemcpyAsync(e->io.adj_syn, adj, bytes, hipMemcpyDeviceToDevice, s));
    return gs_dense_gcn_norm_fwd(e->io.adj_syn, e->n, e->adj_norm_in, e->rnorm_in, stream);
  }
  GS_HIP(hipMemcpyAsync(e->adj_raw, adj, bytes, hipMemcpyDeviceToDevice, s));
  return gs_dense_gcn_norm_fwd(e->adj_raw, e->n, e->adj_norm, e->rnorm, stream);
}

// Sampling + real-side class gradients of this rank's classes, forked onto the engine's second stream when overlap is on
// (the join is taken by the first stage that reads gw_real); also samples the NEXT iteration's blocks behind them.
// ---- exchanges of the sharded iteration (tags: gs_comm_read_timing) -------------------------------------------------
static inline int xchg(gs_comm* comm, void* buf, long count, int f64, int tag, gs_stream_t stream) {
  return comm ? gs_comm_allreduce(comm, buf, count, f64, tag, stream) : GS_OK;
}
// the PGE exchanges (BN-2 sums, pair scores, BN-2 backward sums, marginals | dW1) exist only when the PGE's pair rows are sharded:
// a rank that holds every pair row (gs_gcond_set_shard with r = [0, n'): "replicated PGE") already has the complete values, and
// ONE exchange per iteration is left (the class-partial [d_feat | d_adj | loss_c] of the matching stage)
// (a single-rank communicator still takes every exchange: the identity, but it exercises the transport on each buffer)
static inline int xchg_pge(const gs_gcond* e, gs_comm* comm, void* buf, long count, int f64, int tag, gs_stream_t stream) {
  return (sharded_r(e) || (comm && gs_comm_world(comm) == 1)) ? xchg(comm, buf, count, f64, tag, stream) : GS_OK;
}

static int real_async(gs_gcond* e, gs_stream_t stream, bool fork_recorded);
extern "C" int gs_gcond_stage_real_async(gs_gcond* e, gs_stream_t stream) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  return real_async(e, stream, false);
}

// fork_recorded: the caller has already recorded ev_fork on `stream` (and enqueued independent work behind it: the outer step
// issues the first part of the PGE forward BEFORE this branch's ~20 launches, see outer_step_impl)
static int real_async(gs_gcond* e, gs_stream_t stream, bool fork_recorded) {
  if (e->overlap && e->overlap_sampling_only) {
    // level 2: the real-side gradients stay on the caller's stream; the second stream only samples the NEXT iteration's blocks
    // (into the other block buffer) while they run
    hipStream_t s = (hipStream_t)stream;
    GS_TRY(gs_gcond_stage_sample(e, stream));              // adopts the set sampled during the previous iteration
    if (!e->cfg.external_blocks) {
      GS_HIP(hipEventRecord(e->ev_fork, s));
      GS_HIP(hipStreamWaitEvent(e->aux, e->ev_fork, 0));
      GS_TRY(sample_into(e, e->blk_cur ^ 1, (gs_stream_t)e->aux));
      GS_HIP(hipEventRecord(e->ev_prefetch, e->aux));
      e->prefetched = true;
    }
    GS_TRY(gs_gcond_stage_real_grads(e, stream));
  } else if (e->overlap) {
    hipStream_t s = (hipStream_t)stream;
    if (!fork_recorded) GS_HIP(hipEventRecord(e->ev_fork, s));
    GS_HIP(hipStreamWaitEvent(e->aux, e->ev_fork, 0));
    GS_TRY(gs_gcond_stage_sample(e, (gs_stream_t)e->aux));
    GS_TRY(gs_gcond_stage_real_grads(e, (gs_stream_t)e->aux));
    GS_HIP(hipEventRecord(e->ev_join, e->aux));
    e->join_pending = true;
    if (!e->cfg.external_blocks) {   // sample the NEXT iteration's blocks behind this one's gradients (same RNG sequence)
      GS_TRY(sample_into(e, e->blk_cur ^ 1, (gs_stream_t)e->aux));
      GS_HIP(hipEventRecord(e->ev_prefetch, e->aux));
      e->prefetched = true;
    }
  } else {
    GS_TRY(gs_gcond_stage_sample(e, stream));
    GS_TRY(gs_gcond_stage_real_grads(e, stream));
  }
  return GS_OK;
}


// PGE forward with the two exchanges between its parts when `comm` is given (BN-2 column sums, pair scores)
static int pge_forward_x(gs_gcond* e, int inference, gs_comm* comm, gs_stream_t stream) {
  if (!comm) return gs_gcond_stage_pge_forward(e, inference, stream);
  GS_TRY(gs_gcond_stage_pge_forward_part(e, inference, 1, stream));
  GS_TRY(xchg_pge(e, comm, e->sums, 2L * e->Ph, 1, 0, stream));
  GS_TRY(gs_gcond_stage_pge_forward_part(e, inference, 2, stream));
  GS_TRY(xchg_pge(e, comm, e->e3, (long)e->n * e->n, 0, 1, stream));
  return gs_gcond_stage_pge_forward_part(e, inference, 3, stream);
}

// PGE backward with its two exchanges (BN-2 backward sums; BN-1 marginals — and dW1 | db1 behind them in ONE buffer when the PGE
// is stepped: the weight-gradient partials are copied behind the marginals, exchanged with them and copied back)
static int pge_backward_x(gs_gcond* e, int need_param_grads, int need_feat_grad, gs_comm* comm, gs_stream_t stream) {
  if (!comm) return gs_gcond_stage_pge_backward(e, need_param_grads, need_feat_grad, stream);
  hipStream_t s = (hipStream_t)stream;
  const long Ph = e->Ph, nm = 2L * e->n * Ph, nw = Ph * Ph + Ph;
  GS_TRY(gs_gcond_stage_pge_backward_part(e, need_param_grads, need_feat_grad, 1, stream));
  GS_TRY(xchg_pge(e, comm, e->sums, 3L * Ph, 1, 3, stream));
  GS_TRY(gs_gcond_stage_pge_backward_part(e, need_param_grads, need_feat_grad, 2, stream));
  const bool w1 = need_param_grads && gs_comm_world(comm) > 1 && sharded_r(e);
  if (w1) GS_HIP(hipMemcpyAsync(e->ar_marg + nm, e->d_pge + e->pW1, sizeof(float) * nw, hipMemcpyDeviceToDevice, s));
  GS_TRY(xchg_pge(e, comm, e->ar_marg, nm + (w1 ? nw : 0), 0, 4, stream));
  if (w1) GS_HIP(hipMemcpyAsync(e->d_pge + e->pW1, e->ar_marg + nm, sizeof(float) * nw, hipMemcpyDeviceToDevice, s));
  return gs_gcond_stage_pge_backward_part(e, need_param_grads, need_feat_grad, 3, stream);
}

// One outer iteration with the update rule chosen by the caller (the sibling methods share train_class and differ only here):
//   GCond    (gcond.py:47-72)   : step_pge = epoch%50<10, step_feat = !step_pge, inference + inner steps
//   DosCond  (doscond.py:40-50) : both optimisers stepped, no inference, no inner steps
//   GCondX   (gcondx.py:44-66)  : identity structure, feat stepped unless ol%5<1, inner steps on A^ = I
//   DosCondX (doscondx.py:42-49): identity structure, feat stepped, nothing else
// comm != NULL: this rank's share of a sharded iteration (gs_gcond_set_shard), the all-reduce(sum) exchanges issued here on
// `stream` between the parts — the whole iteration is this ONE host call per rank.
static int outer_step_body(gs_gcond* e, int step_pge, int step_feat, int run_inference, int run_inner, gs_comm* comm, gs_stream_t stream);
static int outer_step_impl(gs_gcond* e, int step_pge, int step_feat, int run_inference, int run_inner, gs_comm* comm,
                           gs_stream_t stream) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  e->in_step = true;
  const int rc = outer_step_body(e, step_pge, step_feat, run_inference, run_inner, comm, stream);
  e->in_step = false;
  if (rc == GS_OK) ++e->drop_iter;      // the next iteration's dropout masks (every stage of one iteration reads the same counter)
  else (void)quiesce_hoist(e);
  return rc;
}
static int outer_step_body(gs_gcond* e, int step_pge, int step_feat, int run_inference, int run_inner, gs_comm* comm,
                           gs_stream_t stream) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  if (!comm && (sharded_c(e) || sharded_r(e)))
    GS_FAIL(GS_ERR_STATE, "gs_gcond_outer_step: a sharded engine needs a communicator (gs_gcond_outer_step_sharded) or is driven stage "
                          "by stage with all-reduces in between");
  if (comm && gs_comm_world(comm) > 1 && e->pge_chunks > 1)
    GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gcond_outer_step_sharded: the chunked PGE (pge_chunks > 1) cannot be sharded over pair rows");
  if (e->identity) { step_pge = 0; run_inference = 0; }
  const bool recompute_fwd = !e->identity && !(e->reuse_fwd && e->fwd_cached);
  // the training forward of this iteration was issued by the previous call beside its inner steps (see the end of this function)
  const bool hoisted = recompute_fwd && e->fwd_hoisted && e->hoist_pending && !comm;
  if (e->hoist_pending && !hoisted) {      // (not what this call needs: wait for it, then compute as usual)
    GS_HIP(hipStreamWaitEvent((hipStream_t)stream, e->ev_hoist, 0));
    e->hoist_pending = false;
  }
  e->fwd_hoisted = false;
  if (hoisted) {
    // the real-side branch forks behind the previous call's inner steps (it reads theta); the main stream takes up the forward's
    // results where the third stream left them.  (The edge product plans for the whole chip: by the time it runs the hoisted
    // forward has finished — it started an inner loop earlier.)
    e->real_beside = hoist_mode(e) == 2;     // (2: the 7/8 slab plan of the un-hoisted step — measurement: 1 467 against 1 449-1 458 us)
    const int rc_real = gs_gcond_stage_real_async(e, stream);
    e->real_beside = false;
    GS_TRY(rc_real);
    GS_HIP(hipStreamWaitEvent((hipStream_t)stream, e->ev_hoist, 0));
    e->hoist_pending = false;
  } else if (recompute_fwd && e->overlap && !e->overlap_sampling_only) {
    // Issue order: the two branches are independent, and the host needs ~3 us per launch.  A class shard's real-side branch is
    // ~20 short launches: enqueued first (round-3 trace of rank 2 of 8) they left the main queue EMPTY for 195 us while the host
    // was still feeding the second stream.  So the fork point is recorded, the first part of the PGE forward (its long product)
    // goes to the main queue, THEN the real-side branch to the second stream, then the rest of the forward.
    hipStream_t s = (hipStream_t)stream;
    // (knob FORK_LATE: the real-side branch starts only when the forward's long product has finished — that product then has the
    // chip to itself, e.g. as a persistent grid — instead of beside it)
    const bool late = knob_of(e, GS_KNOB_FORK_LATE, 0) != 0;
    if (!late) GS_HIP(hipEventRecord(e->ev_fork, s));
    GS_TRY(gs_gcond_stage_pge_forward_part(e, 0, 1, stream));
    if (late) GS_HIP(hipEventRecord(e->ev_fork, s));
    GS_TRY(xchg_pge(e, comm, e->sums, 2L * e->Ph, 1, 0, stream));
    e->real_beside = true;
    const int rc_real = real_async(e, stream, true);
    e->real_beside = false;
    GS_TRY(rc_real);
    GS_TRY(gs_gcond_stage_pge_forward_part(e, 0, 2, stream));
    GS_TRY(xchg_pge(e, comm, e->e3, (long)e->n * e->n, 0, 1, stream));
    GS_TRY(gs_gcond_stage_pge_forward_part(e, 0, 3, stream));
  } else {
    GS_TRY(gs_gcond_stage_real_async(e, stream));
    if (!e->identity) {
      if (!recompute_fwd)
        GS_TRY(adopt_cached_forward(e, stream));
      else
        GS_TRY(pge_forward_x(e, 0, comm, stream));
    }
  }
  // (unsharded: the loss bookkeeping rides in the matching stage's V kernel — every class loss is final there; sharded ranks sum the
  // class partials first)
  e->fold_finish = !comm && !sharded_c(e) && e->c_hi > e->c_lo && knob_of(e, GS_KNOB_FUSED_CHAINS, g_fused_chains);
  // (two-transform model, chained class products: the pieces of the stage that only feed d loss / d X~ are skipped when X~ is not
  // stepped — zero_grad() discards that gradient unread, gcond.py:54-61)
  e->skip_dfeat = !step_feat && !e->gcn && e->T == 2 && !e->blk_pos && knob_of(e, GS_KNOB_FUSED_CHAINS, g_fused_chains) &&
                  knob_of(e, GS_KNOB_CHAINED_PRODUCTS, g_chained_products) && gs_match_class_eligible(e->n, e->h, e->d, e->C);
  const bool folded = e->fold_finish;
  const int rc_match = gs_gcond_stage_syn_match(e, stream);
  e->fold_finish = false;
  e->skip_dfeat = false;
  GS_TRY(rc_match);
  GS_TRY(xchg(comm, e->ar_syn, (long)e->n * e->d + (long)e->n * e->n + e->C, 0, 2, stream));
  if (!folded) GS_TRY(gs_gcond_stage_syn_finish(e, stream));
  // only the gradient of a group that is stepped is consumed (zero_grad() discards the rest unread, gcond.py:54-61)
  if (!e->identity && (step_pge || step_feat)) GS_TRY(pge_backward_x(e, step_pge, step_feat, comm, stream));
  if (step_pge)
    GS_TRY(gs_adam_step_f32(e->io.pge_params, e->d_pge, e->m_pge, e->v_pge, e->Ppge, e->cfg.lr_adj, 0.9f, 0.999f, 1e-8f, 0.f,
                            ++e->step_pge, stream));
  if (step_feat)
    GS_TRY(gs_adam_step_f32(e->io.feat_syn, e->d_feat, e->m_feat, e->v_feat, (long)e->n * e->d, e->cfg.lr_feat, 0.9f, 0.999f,
                            1e-8f, 0.f, ++e->step_feat, stream));
  if (step_pge || step_feat) e->fwd_cached = false;
  if (run_inference) GS_TRY(pge_forward_x(e, 1, comm, stream));
  if (run_inner) {
    if (!e->identity && !run_inference) GS_FAIL(GS_ERR_ARG, "gs_gcond_outer_step_ex: inner steps need the inference pass");
    // Hoist: gcond.py:48 of the NEXT iteration, `adj_syn = pge(feat_syn)`, reads (X~, theta_PGE) only — both final now — while the
    // inner steps (gcond.py:67-72) are ~20 dependent launches on a few compute units.  The same function of the same inputs into the same
    // buffers as at the top of the next call, on the persistent kernel with a capped grid (another fixed partition of the BatchNorm-2
    // column sums: rounding-level differences from the un-hoisted pass, repeat runs bitwise equal), issued on the lowest-priority stream so that the inner
    // steps' small launches take compute units as they free up.  Not with forward reuse (nothing to recompute), not sharded (the
    // forward's exchanges stay in the next call), not without the second stream.
    const bool hoist = hoist_mode(e) && !comm && !sharded_c(e) && !sharded_r(e) && !e->identity && run_inference && !e->reuse_fwd &&
                       e->overlap && !e->overlap_sampling_only && e->low && e->cfg.inner_loop > 0;
    if (hoist) {
      hipStream_t s = (hipStream_t)stream;
      GS_HIP(hipEventRecord(e->ev_inf, s));
      GS_HIP(hipStreamWaitEvent(e->low, e->ev_inf, 0));
      const bool cached = e->fwd_cached;
      e->hoisting = true;
      const int rc_fwd = gs_gcond_stage_pge_forward(e, 0, (gs_stream_t)e->low);
      e->hoisting = false;
      if (rc_fwd != GS_OK) {      // launches already issued on `low` may still be writing U / V / E2 / adj_raw: drain them before the caller retries
        (void)hipStreamSynchronize(e->low);
        return rc_fwd;
      }
      e->fwd_cached = cached;
      GS_HIP(hipEventRecord(e->ev_hoist, e->low));
      e->hoist_pending = true;
      e->fwd_hoisted = true;
    }
    GS_TRY(gs_gcond_stage_inner_steps(e, stream));
  }
  return GS_OK;
}

extern "C" int gs_gcond_outer_step_ex(gs_gcond* e, int step_pge, int step_feat, int run_inference, int run_inner,
                                      gs_stream_t stream) {
  return outer_step_impl(e, step_pge, step_feat, run_inference, run_inner, nullptr, stream);
}

extern "C" int gs_gcond_outer_step(gs_gcond* e, int epoch, gs_stream_t stream) {
  const int pge_epoch = (epoch % 50 < 10);     // gcond.py:58-61
  return outer_step_impl(e, pge_epoch, !pge_epoch, 1, 1, nullptr, stream);
}

extern "C" int gs_gcond_outer_step_sharded_ex(gs_gcond* e, int step_pge, int step_feat, int run_inference, int run_inner,
                                              gs_comm* comm, gs_stream_t stream) {
  if (!comm) GS_FAIL(GS_ERR_ARG, "gs_gcond_outer_step_sharded: null communicator");
  return outer_step_impl(e, step_pge, step_feat, run_inference, run_inner, comm, stream);
}

extern "C" int gs_gcond_outer_step_sharded(gs_gcond* e, int epoch, gs_comm* comm, gs_stream_t stream) {
  const int pge_epoch = (epoch % 50 < 10);
  return gs_gcond_outer_step_sharded_ex(e, pge_epoch, !pge_epoch, 1, 1, comm, stream);
}

// In-situ timing of the dominant kernel: while on, HIP events are recorded on the launch stream around every PGE layer-2
// GEMM (training forward and inference pass), up to 256 launches; reading synchronises the events and returns the summed
// duration and the launch count since the probe was switched on.
extern "C" int gs_gcond_probe(gs_gcond* e, int on) {
  if (!e) GS_FAIL(GS_ERR_ARG, "null engine");
  e->probing = on != 0;
  if (on) {
    e->probe_n = 0;
    e->probe2_n = 0;
    e->probe_stride = on > 1 ? on : 1;      // (on > 1: HIP-event brackets on every on-th launch only; the device-side spans on all)
    e->probe_seen[0] = e->probe_seen[1] = e->probe2_seen = 0;
    // (every event the probe may record exists before the first probed launch: creating them lazily put ~300 hipEventCreate calls
    // into the first probed steps — the timed ones of bench.py)
    for (int i = 0; i < 2 * 256; ++i) {
      if (!e->probe_ev[i]) GS_HIP(hipEventCreate(&e->probe_ev[i]));
      if (!e->probe2_ev[i]) GS_HIP(hipEventCreate(&e->probe2_ev[i]));
    }
    // every slot: start = max u64 (atomicMin target), end = 0 (atomicMax target)
    unsigned long long init[2 * 256];
    for (int i = 0; i < 256; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0ull; }
    GS_HIP(hipMemcpy(e->probe_span, init, sizeof(init), hipMemcpyHostToDevice));
  }
  return GS_OK;
}

// Device-side spans of the probed launches: (last workgroup's end tick - first workgroup's start tick) summed over the launches
// that recorded one, in ms (s_memrealtime counts at 100 MHz).  The HIP-event bracket of gs_gcond_probe_read also contains the
// time the launch waited for compute units held by the other stream's resident workgroups; this span does not, and is the
// figure a rocprofv3 kernel trace reports as the kernel's duration.
extern "C" int gs_gcond_probe_read_spans(gs_gcond* e, double* total_ms, long* launches) {
  if (!e || !total_ms || !launches) GS_FAIL(GS_ERR_ARG, "gs_gcond_probe_read_spans: null argument");
  unsigned long long h[2 * 256];
  GS_HIP(hipDeviceSynchronize());
  GS_HIP(hipMemcpy(h, e->probe_span, sizeof(h), hipMemcpyDeviceToHost));
  double t = 0;
  long k = 0;
  for (int i = 0; i < e->probe_n; ++i)
    if (h[2 * i + 1] > h[2 * i] && h[2 * i] != ~0ull) { t += (double)(h[2 * i + 1] - h[2 * i]) * 1e-5; ++k; }
  *total_ms = t;
  *launches = k;
  return GS_OK;
}

// The same two figures for ONE pass of the probed product (pass 0: training forward, 1: inference forward): HIP-event bracket [ms],
// launches, device-side span [ms] and the launches that recorded one.  bench.py reports the two kernels separately.
extern "C" int gs_gcond_probe_read_pass(gs_gcond* e, int pass, double* bracket_ms, long* launches, double* span_ms, long* span_launches) {
  if (!e || !bracket_ms || !launches || !span_ms || !span_launches) GS_FAIL(GS_ERR_ARG, "gs_gcond_probe_read_pass: null argument");
  unsigned long long h[2 * 256];
  GS_HIP(hipDeviceSynchronize());
  GS_HIP(hipMemcpy(h, e->probe_span, sizeof(h), hipMemcpyDeviceToHost));
  double tb = 0, ts = 0;
  long kb = 0, ks = 0;
  for (int i = 0; i < e->probe_n; ++i) {
    if ((int)e->probe_pass[i] != (pass != 0)) continue;
    if (e->probe_has_ev[i]) {
      float ms = 0.f;
      GS_HIP(hipEventSynchronize(e->probe_ev[2 * i + 1]));
      GS_HIP(hipEventElapsedTime(&ms, e->probe_ev[2 * i], e->probe_ev[2 * i + 1]));
      tb += ms; ++kb;
    }
    if (h[2 * i + 1] > h[2 * i] && h[2 * i] != ~0ull) { ts += (double)(h[2 * i + 1] - h[2 * i]) * 1e-5; ++ks; }
  }
  *bracket_ms = tb; *launches = kb; *span_ms = ts; *span_launches = ks;
  return GS_OK;
}

extern "C" int gs_gcond_probe_read(gs_gcond* e, double* total_ms, long* launches) {
  if (!e || !total_ms || !launches) GS_FAIL(GS_ERR_ARG, "gs_gcond_probe_read: null argument");
  double t = 0;
  long k = 0;
  for (int i = 0; i < e->probe_n; ++i) {
    if (!e->probe_has_ev[i]) continue;
    float ms = 0.f;
    GS_HIP(hipEventSynchronize(e->probe_ev[2 * i + 1]));
    GS_HIP(hipEventElapsedTime(&ms, e->probe_ev[2 * i], e->probe_ev[2 * i + 1]));
    t += ms; ++k;
  }
  *total_ms = t;
  *launches = k;
  return GS_OK;
}

// the same bracket around the hop-2 gather SpMM of the real side (the HBM / Infinity-Cache bound kernel of the iteration), recorded
// on the stream it is launched on (the engine's second stream when the branches overlap)
extern "C" int gs_gcond_probe_read_spmm(gs_gcond* e, double* total_ms, long* launches) {
  if (!e || !total_ms || !launches) GS_FAIL(GS_ERR_ARG, "gs_gcond_probe_read_spmm: null argument");
  double t = 0;
  for (int i = 0; i < e->probe2_n; ++i) {
    float ms = 0.f;
    GS_HIP(hipEventSynchronize(e->probe2_ev[2 * i + 1]));
    GS_HIP(hipEventElapsedTime(&ms, e->probe2_ev[2 * i], e->probe2_ev[2 * i + 1]));
    t += ms;
  }
  *total_ms = t;
  *launches = e->probe2_n;
  return GS_OK;
}

extern "C" int gs_gcond_read_loss(gs_gcond* e, double* loss_sum, long* steps, gs_stream_t stream) {
  if (!e || !loss_sum || !steps) GS_FAIL(GS_ERR_ARG, "gs_gcond_read_loss: null argument");
  hipStream_t s = (hipStream_t)stream;
  GS_HIP(hipMemcpyAsync(loss_sum, e->loss_sum, sizeof(double), hipMemcpyDeviceToHost, s));
  GS_HIP(hipMemcpyAsync(steps, e->steps, sizeof(long), hipMemcpyDeviceToHost, s));
  GS_HIP(hipMemsetAsync(e->loss_sum, 0, sizeof(double), s));
  GS_HIP(hipMemsetAsync(e->steps, 0, sizeof(long), s));
  unsigned ip_err = 0;
  if (e->ip_ws) GS_HIP(hipMemcpyAsync(&ip_err, e->ipw.sync + 16, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  GS_HIP(hipStreamSynchronize(s));
  if (ip_err) {      // a grid barrier of the persistent inner-step launch gave up (gs_inner.hip): theta of that iteration is garbage
    (void)hipMemsetAsync(e->ipw.sync, 0, 256, s);
    (void)hipStreamSynchronize(s);
    e->ip_counter = 0;
    GS_FAIL(GS_ERR_STATE, "gs_gcond: grid barrier %u of the persistent inner-step launch timed out (its workgroups were not co-resident "
                          "for seconds): the model parameters of that iteration are invalid — run with the launch chain "
                          "(gs_gcond_set_knob(e, GS_KNOB_INNER_PERSIST, 0)) when the device is shared", ip_err);
  }
  return GS_OK;
}

extern "C" int gs_gcond_buffer(gs_gcond* e, const char* name, void** ptr, long* count) {
  if (!e || !name || !ptr || !count) GS_FAIL(GS_ERR_ARG, "gs_gcond_buffer: null argument");
  GS_TRY(quiesce_hoist(e));
  const long n = e->n, C = e->C, Ph = e->Ph;
  struct { const char* k; void* p; long c; } tab[] = {
      {"gw_real", e->gw_real, C * e->P},   {"gw_syn", e->gw_syn, C * e->P},     {"loss", e->loss_last, 1},
      {"loss_c", e->loss_c, C},            {"d_feat", e->d_feat, n * e->d},     {"d_adj_norm", e->dAn, n * n},
      {"d_adj", e->d_adj, n * n},          {"d_pge", e->d_pge, e->Ppge},        {"adj_norm", e->adj_norm, n * n},
      {"adj_raw", e->adj_raw, n * n},      {"adj_norm_inner", e->adj_norm_in, n * n}, {"D", e->D, C * e->P},
      {"g_model", e->g_model, e->P},       {"m_model", e->m_model, e->P},       {"v_model", e->v_model, e->P},       {"Sr", e->Sr, C * e->bcap * e->Fp},   {"rsr", e->rsr, C * e->bcap},
      {"Q", e->Q, n * n},                  {"Ss", e->Ss, n * e->F},             {"Ps", e->Ps, n * C},
      {"dQ", e->dQ, n * n},                {"e3", e->e3, n * n},                {"H1full", e->H1full, e->H1full ? (long)e->N * e->h : 0},
      // all-reduce(sum) buffers of the sharded engine.  ar_bn2 / ar_l3 are fp64 (count in doubles), the rest fp32.
      {"ar_bn2", e->sums, 2 * Ph},         {"ar_l3", e->sums, 3 * Ph},          {"ar_syn", e->ar_syn, n * e->d + n * n + C},
      {"ar_marg", e->ar_marg, 2 * n * Ph}, {"ar_w1", e->d_pge + e->pW1, Ph * Ph + Ph},
      {"ar_gw_real", e->gw_real, C * e->P},  {"H1bits", e->H1bits, e->H1bits ? (long)e->N * (e->h / 32) : 0},
      // PGE layer-2 pre-activations of this rank's pair rows and the BatchNorm-2 batch statistics (kernel A/B tests)
      {"E2", e->E2, (long)(e->r_hi - e->r_lo) * n * Ph}, {"mu2", e->mu2, Ph}, {"is2", e->is2, Ph},
      {"edge_bits", e->edge_bits, e->edge_bits ? (long)C * e->cap1 * e->f2 * (e->h / 32) : 0},
      {"edge_plan", e->edge_plan, e->edge_plan ? 2L * C + 1 : 0},      // balanced slabs of the last real-side stage (ints)
      // distinct sources of the current blocks' outermost hop (ints): counts, sources, CSC
      {"dd_n0", e->dd[e->blk_cur].n0, e->cap0 ? C : 0}, {"dd_src0", e->dd[e->blk_cur].src0, e->cap0 ? C * (long)e->cap0 : 0},
      {"dd_sptr0", e->dd[e->blk_cur].sptr0, e->cap0 ? C * ((long)e->cap0 + 1) : 0},
      {"dd_sedge0", e->dd[e->blk_cur].sedge0, e->cap0 ? C * (long)e->cap0 : 0},
      {"dd_nent", e->dd[e->blk_cur].nent, e->cap0 ? C : 0}, {"dd_ent_src", e->dd[e->blk_cur].ent_src, e->cap0 ? C * (long)e->cap0 : 0},
      {"dd_ent_row", e->dd[e->blk_cur].ent_row, e->cap0 ? 2 * C * (long)e->cap0 : 0}, {"dd_ent_w", e->dd[e->blk_cur].ent_w, e->cap0 ? 2 * C * (long)e->cap0 : 0},
      // hidden activations of the synthetic rows and their gradient as the last inner step left them (two-transform models)
      {"H1s", e->H1s, e->H1s ? n * e->h : 0}, {"dH1m", e->dH1m, e->dH1m ? n * e->h : 0},
      // PGE backward intermediates where they are materialised (diagnosis: tools/diag_panel_stages.py)
      {"dE2", e->dE2, n * n * Ph}, {"A1", e->A1, n * n * Ph}, {"dY1", e->dY1, n * n * Ph}, {"W1T", e->W1T, Ph * Ph}, {"de3", e->de3, n * n},
  };
  for (auto& t : tab)
    if (!strcmp(t.k, name)) {
      *ptr = t.p;
      *count = t.c;
      return GS_OK;
    }
  GS_FAIL(GS_ERR_ARG, "gs_gcond_buffer: unknown buffer '%s'", name);
}
