// Row-panel product for the three big dense contractions of the GCond iteration (VERDICT r01 "next" 2-4):
//
//     C[M x 256] = A[M x K] * B^T  (+ bias[n]) (relu) (* [mask > 0])        A: [M][lda] fp32, k contiguous
//                                                                             B: [256][ldb] fp32, k contiguous
//   PGE layer 2      E2  = A1  W1^T + b1     M = n'^2 = 206 116, K = 256     (parametrized_adj.py:57-64)  + BN-2 column sums
//   PGE layer 2 bwd  dY1 = (dE2 W1) * [A1>0]                                  (autograd of the same layer)
//   SGC transform    H1  = relu(X W1 + b1)   M = N = 169 343,    K = 128     (models/sgc.py:38-43)
//
// One workgroup = one 128-row panel over ALL 256 output columns, so A is read from HBM exactly once (the 128x128 tile read
// it once per N tile); wave w owns rows [32w, 32w + 32) x 256 columns = 8 MFMA tiles of 32x32 (128 accumulator registers).
//   * A never touches LDS: each lane loads 16 consecutive k of ITS row (4 x 16 B; the two half-waves share a 128-B line)
//     straight into the MFMA fragment registers, one K block ahead.
//   * B (<= 256 KB, L2-resident) is streamed through LDS in K blocks of 32 by direct-to-LDS loads (global_load_lds_dwordx4,
//     no staging registers), double buffered, one barrier per K block.  The LDS image is lane-linear [256 rows][8 chunks of
//     16 B]; bank conflicts are removed by XOR-ing the chunk index with (row >> 1) & 7 on the SOURCE address and on the read
//     (the 16 lanes of a ds_read_b128 group then hit 16 distinct 16-B slots).
// Operand precision (gs_gemm_desc.prec):
//   0  exact fp32          v_mfma_f32_32x32x2_f32 (k-ordered fmaf chain, the reference's arithmetic)
//   1  split bf16 x3       A = Ah + Al, B = Bh + Bl in bf16 (2^-16 relative representation), C += Ah Bh + Ah Bl + Al Bh on
//                          v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 3 MFMAs at 16x the fp32 rate -> HBM-bound
//   2  bf16                C += Ah Bh only (BASELINE.json configs[1] "bf16": operands rounded to 8 significant bits)
// B is pre-split once per call by a tiny kernel into [256][K/32][hi 32 | lo 32] bf16 (same 128 B per row and K block as the
// fp32 image, so the staging code is shared).
#include "gs_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* gs_lds_ptr;
typedef __attribute__((address_space(1))) const void* gs_gbl_ptr;

#define PN_BM 128

struct gs_panel_args {
  const float* A; long lda;
  const void* B; long ldb_bytes;     // fp32 [256][ldb] or packed bf16 [256][K/32][64]
  float* C; long ldc;
  int M, K;
  const float* bias;
  int relu;
  const float* mask; long ldmask;
  float* colstat;                    // [tiles_m][2][256] or null
  int n_full;                        // workgroups [0, n_full) run 128-row tiles, the rest 32-row tiles (the last partial round)
  // fused-marginals form (gs_panel_marg_kernel): rows are the pairs (r, c) of an [nr x n] grid, a tile = 16 r x 8 c
  int mg_n, mg_nr, mg_ncb;
  float* partU;                      // [ceil(nr/16)][n][N]: sum over the tile's 16 r
  float* partV;                      // [ceil(n/8)][nr][N]:  sum over the tile's 8 c
  // generated A operand (AGEN): A holds E2; the operand is dE2 = BatchNorm-2 + layer-3 backward of it, formed in registers
  const float* gen_c;                // [K][8]: per column k  {is2, -mu2 is2, g2, be2, g2 is2 w2, g2 is2 dbe2 / N, g2 is2 dg2 / N, 0}
  const float* gen_d;                // [rows]: de3 of each pair row
  // generated layer-1 activations: A1[(r, c)][k] = relu(g1 ((U[c][k] + V[r][k] + b0 - mu1) is1) + be1), the arithmetic of
  // gs_pge_make_a1_v5_kernel.  FGEN: the A operand of the forward product is formed from U, V (A is not read).  MGEN: the
  // [A1 > 0] mask of the fused-marginals epilogue is recomputed from them (mask is not read).
  const float* a1_U; const float* a1_V; long a1_ld;     // U [n][ld], V already offset to this rank's first pair row r
  const float* a1_c;                 // [Ph][8]: per column {b0, mu1, is1, g1, be1, 0, 0, 0}
  int a1_fold;                       // MGEN: a1_U / a1_V hold the folded terms Us, Vs (gs_pge_scale_uv_kernel)
  // row-list form (gs_panel_rows_kernel): row i of the product reads A[rowidx[i]] and writes C[rowidx[i]]; the row count is read
  // on the device (*m_dev, <= the M the grid was sized for) — the rows of H1 = relu(X W1 + b1) a rank's sampled blocks gather
  const int* rowidx; const int* m_dev;
  unsigned* bits; long ldbits;       // optional sign bits of the written C: bit (n & 31) of bits[row * ldbits + (n >> 5)] = [C(row, n) > 0]
  unsigned long long* span;          // optional [2]: device-side launch span — min over workgroups of the start tick, max of the
                                     // end tick (s_memrealtime, 100 MHz); see gs_gcond_probe
  int krep, triage;                  // measurement hooks (gs_panel_debug): K loop repeated krep times; bit 0: no stores;
  unsigned long long* trace;         // bit 1: per-workgroup [hw_id, xcc_id, t_start, t_loop_end, t_end] (s_memrealtime, 10 ns)
};

// Epilogue of one wave: its 32 rows x 32 NTW columns from the accumulators.  C/D layout of a 32x32 tile: col = lane & 31,
// row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
// No load may sit between the stores of two tiles: vmcnt retires in order, so waiting for such a load waits for every store
// issued before it — a full write round trip per tile (17-37 us of the 70 us the first build of this kernel spent outside its
// K loop, tools/trace_panel.py).  So the bias row is parked in LDS before the K loop (LDS reads count on lgkmcnt, not vmcnt) and
// the code is branch-free per (HAS_MASK, FULL) instantiation.
template <int NTW, bool HAS_MASK, bool FULL, int BS = 1, bool ROWS = false, bool BITS = false>
__device__ __forceinline__ void panel_epilogue(const gs_panel_args& p, const f32x16 (&acc)[NTW], const float* sbias, int m0,
                                               int row_in_tile, int col0, float (&cs1)[NTW], float (&cs2)[NTW]) {
  const float floor_ = p.relu ? 0.f : -__builtin_inff();
  // Addresses = a tile-uniform base (scalar registers) + a 32-bit per-lane offset + an immediate per N tile: one offset
  // register per accumulator row instead of a 64-bit address pair per store (the pair form needed > 256 registers and
  // spilled; a kernel that needs scratch memory also makes the runtime manage scratch per queue on the launch path).
  float* __restrict__ Cb = ROWS ? p.C : p.C + (long)m0 * p.ldc;
  const float* __restrict__ Mb = HAS_MASK ? p.mask + (long)m0 * p.ldmask : nullptr;
  unsigned coff[16], moff[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const unsigned rt = (unsigned)(row_in_tile + (r & 3) + 8 * (r >> 2));      // row inside the tile
    if constexpr (ROWS) coff[r] = (unsigned)p.rowidx[min(m0 + (int)rt, p.M - 1)] * (unsigned)p.ldc + (unsigned)col0;
    else coff[r] = rt * (unsigned)p.ldc + (unsigned)col0;
    if (HAS_MASK) moff[r] = (FULL ? rt : min(rt, (unsigned)(p.M - 1 - m0))) * (unsigned)p.ldmask + (unsigned)col0;
  }
  // (mask variant: the 16 mask values of a tile are requested right before use — a two-tile ping-pong needs 32 more
  // registers next to the 128 accumulators; its in-order wait on the previous tile's stores is the known cost of this
  // variant)
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    float mk[16];
    if (HAS_MASK) {
#pragma unroll
      for (int r = 0; r < 16; ++r) mk[r] = Mb[moff[r] + 32u * j];
    }
    float v[16];
    const float bvj = sbias[(col0 + 32 * j) * BS];      // (BS = 8: the bias sits in slot 7 of the mode-3 constants table)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      v[r] = fmaxf(acc[j][r] + bvj, floor_);
      if (HAS_MASK) v[r] = (mk[r] > 0.f) ? v[r] : 0.f;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (FULL || m0 + row_in_tile + (r & 3) + 8 * (r >> 2) < p.M) {
        Cb[coff[r] + 32u * j] = v[r];
        s1 += v[r];
        s2 = fmaf(v[r], v[r], s2);
      }
    }
    // (BITS is a compile-time form of its own: as a run-time test this block put a branch between the stores of two tiles and the
    // plain products' epilogues got 10 us slower and 9 % dirtier in WRITE_SIZE / FETCH_SIZE — round-3 PMC diff)
    if constexpr (BITS) {      // sign bits: one ballot per accumulator register = 32 columns of two rows (the two half-waves)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned long long m = __ballot(v[r] > 0.f);
        const int rt = m0 + row_in_tile + (r & 3) + 8 * (r >> 2);
        if ((threadIdx.x & 31) == 0 && (FULL || rt < p.M)) {
          const long grow = ROWS ? (long)p.rowidx[rt] : (long)rt;
          p.bits[grow * p.ldbits + ((col0 + 32 * j) >> 5)] = (threadIdx.x & 32) ? (unsigned)(m >> 32) : (unsigned)m;
        }
      }
    }
    cs1[j] = s1;
    cs2[j] = s2;
  }
}

// Fused-marginals epilogue (the PGE backward needs dY1 = (dE2 W1) * [A1 > 0] ONLY through its two marginals
// colsum_r[r] = sum_c dY1[(r, c)] and rowsum_c[c] = sum_r dY1[(r, c)]): the tile's 128 rows are the pairs of a 16 r x 8 c patch
// (row i of the tile = pair (16 rb + (i >> 3), 8 cb + (i & 7))), so that in the C/D layout (row = (reg & 3) + 8 (reg >> 2) +
// 4 (lane >> 5) inside the wave's 32 rows) reg >> 2 is the r of the wave's 4 and (reg & 3) + 4 (lane >> 5) is the c of the 8:
//   sum over c  = 4 registers + the other half-wave  -> partV[cb][r][col]   (written by the wave itself)
//   sum over r  = 4 registers, then the 4 waves through LDS -> partU[rb][c][col]
// dY1 (211 MB at ogbn-arxiv) is neither written nor read back; gs_reduce_slabs_f32 adds the partials in a fixed order.
template <int NT, bool MGEN>
__device__ __forceinline__ void panel_epilogue_marg(const gs_panel_args& p, const f32x16 (&acc)[NT], char* smem, int rb, int cb,
                                                    int wave, int li, int h) {
  constexpr int N = 32 * NT;
  const int n = p.mg_n, nr = p.mg_nr;
  unsigned moff[MGEN ? 1 : 16], uoff[4], voff[4];
  bool ok[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rl = 16 * rb + 4 * wave + (r >> 2), c = 8 * cb + (r & 3) + 4 * h;
    ok[r] = rl < nr && c < n;
    if constexpr (!MGEN) moff[r] = (unsigned)(min(rl, nr - 1) * n + min(c, n - 1)) * (unsigned)p.ldmask + (unsigned)li;
  }
  if constexpr (MGEN) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uoff[q] = (unsigned)min(8 * cb + q + 4 * h, n - 1) * (unsigned)p.a1_ld + (unsigned)li;
      voff[q] = (unsigned)min(16 * rb + 4 * wave + q, nr - 1) * (unsigned)p.a1_ld + (unsigned)li;
    }
  }
  __syncthreads();                                   // every wave is done with the B stages: they become the [4][8][N] scratch
  float* red = reinterpret_cast<float*>(smem);
  // MGEN: the layer-1 constants of the N output columns go behind the scratch ([N][8] floats); the U / V values of the next
  // N tile are requested BEFORE this tile's partial sums are stored (a load issued after a store waits for it: vmcnt retires
  // in order)
  const f32x4* __restrict__ a1c = reinterpret_cast<const f32x4*>(smem + 32 * N * 4);
  float un[4], vn[4];
  if constexpr (MGEN) {
    f32x4* w = reinterpret_cast<f32x4*>(smem + 32 * N * 4);
    for (int i = threadIdx.x; i < 2 * N; i += 256) w[i] = reinterpret_cast<const f32x4*>(p.a1_c)[i];
#pragma unroll
    for (int q = 0; q < 4; ++q) { un[q] = p.a1_U[uoff[q]]; vn[q] = p.a1_V[voff[q]]; }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    float v[16];
    if constexpr (MGEN) {
      float uc[4], vc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { uc[q] = un[q]; vc[q] = vn[q]; }
      if (j + 1 < NT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { un[q] = p.a1_U[uoff[q] + 32u * (j + 1)]; vn[q] = p.a1_V[voff[q] + 32u * (j + 1)]; }
      }
      const f32x4 c0 = a1c[2 * (32 * j + li)], c1 = a1c[2 * (32 * j + li) + 1];
#pragma unroll
      for (int r = 0; r < 16; ++r) {       // register r: r index (r >> 2) of the wave's 4, c index (r & 3) of the half-wave's 4
        const float x = ((uc[r & 3] + vc[r >> 2]) + c0[0]) - c0[1];
        // (a1_fold: a1_U / a1_V are the folded terms Us, Vs and the activation is max(Us[c] + Vs[r], 0) — the forward's form)
        const bool pos = p.a1_fold ? uc[r & 3] + vc[r >> 2] > 0.f : fmaf(c0[3], x * c0[2], c1[0]) > 0.f;
        v[r] = (ok[r] && pos) ? acc[j][r] : 0.f;
      }
    } else {
      float mk[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) mk[r] = p.mask[moff[r] + 32u * j];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = (ok[r] && mk[r] > 0.f) ? acc[j][r] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float sv = (v[4 * q] + v[4 * q + 1]) + (v[4 * q + 2] + v[4 * q + 3]);          // over this half-wave's 4 c
      sv += __shfl_xor(sv, 32, 64);                                                   // + the other 4
      const int rl = 16 * rb + 4 * wave + q;
      if (h == 0 && rl < nr) p.partV[((long)cb * nr + rl) * N + 32 * j + li] = sv;
      const float su = (v[q] + v[4 + q]) + (v[8 + q] + v[12 + q]);                    // over the wave's 4 r, c = 8 cb + q + 4 h
      red[(wave * 8 + q + 4 * h) * N + 32 * j + li] = su;
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 8 * N; idx += 256) {
    const int ci = idx / N, col = idx - ci * N, c = 8 * cb + ci;
    const float t = ((red[(0 * 8 + ci) * N + col] + red[(1 * 8 + ci) * N + col]) + red[(2 * 8 + ci) * N + col]) + red[(3 * 8 + ci) * N + col];
    if (c < n) p.partU[((long)rb * n + c) * N + col] = t;
  }
}

// In-place transform of a lane's 16 A elements (k = 32 kb + 16 h + t) for the generated-operand forms.
//   AGEN: a holds E2[row][k] -> dE2 = g2 is2 (dy2 - dbe2 / N - xh2 dg2 / N), xh2 = (E2 - mu2) is2, dy2 = de3 w2 [g2 xh2 + be2 > 0]
//         (gs_pge_bn2_bwd_v4_kernel, same arithmetic, constants folded into the table)
//   FGEN: a holds U[c][k]   -> A1[(r, c)][k] = relu(g1 ((U[c][k] + V[r][k] + b0 - mu1) is1) + be1)   (gs_pge_make_a1_v5_kernel,
//         same arithmetic; the tile's two V rows in table slots 5 / 6 (mode 3) or in LDS at vs)
// The per-column constants sit in LDS as a STRUCTURE OF ARRAYS tab[slot][GS_TAB_K] (slot = the position inside the [K][8] rows the
// producers write): one 16-byte LDS read delivers a constant for 4 consecutive k, so the 16 k of a lane need 4 rounds of 6-7 reads
// issued together.  The first form read one [8]-row per k — 32 dependent LDS round trips per 32-k block, each waited for with
// lgkmcnt(0), in front of the block's MFMAs (round-3 ISA reading; the product ran at 3x its MFMA floor).  Same values: same
// operations in the same order per element.
#define GS_TAB_K 256
template <bool AGEN, bool FGEN, bool VTAB = false, bool FOLD = false>
__device__ __forceinline__ void panel_transform_a(const gs_panel_args& p, f32x4 (&a)[4], const float* __restrict__ tab,
                                                  const float* __restrict__ vs, float gen_d, int rsel, int kb, int h) {
  const int k0 = 32 * kb + 16 * h;
  if constexpr (AGEN) {
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const float* q = tab + k0 + 4 * t4;
      const f32x4 is2 = *reinterpret_cast<const f32x4*>(q), nm = *reinterpret_cast<const f32x4*>(q + GS_TAB_K),
                  g2 = *reinterpret_cast<const f32x4*>(q + 2 * GS_TAB_K), be2 = *reinterpret_cast<const f32x4*>(q + 3 * GS_TAB_K),
                  c4 = *reinterpret_cast<const f32x4*>(q + 4 * GS_TAB_K), c5 = *reinterpret_cast<const f32x4*>(q + 5 * GS_TAB_K),
                  c6 = *reinterpret_cast<const f32x4*>(q + 6 * GS_TAB_K);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float xh = fmaf(a[t4][t], is2[t], nm[t]);
        const float y = fmaf(g2[t], xh, be2[t]);
        const float sel = y > 0.f ? gen_d * c4[t] : 0.f;
        a[t4][t] = fmaf(-xh, c6[t], sel - c5[t]);
      }
    }
  }
  if constexpr (FGEN) {
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const float* q = tab + k0 + 4 * t4;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(q), mu1 = *reinterpret_cast<const f32x4*>(q + GS_TAB_K),
                  is1 = *reinterpret_cast<const f32x4*>(q + 2 * GS_TAB_K), g1 = *reinterpret_cast<const f32x4*>(q + 3 * GS_TAB_K),
                  be1 = *reinterpret_cast<const f32x4*>(q + 4 * GS_TAB_K);
      f32x4 vv;
      if constexpr (VTAB) vv = *reinterpret_cast<const f32x4*>(q + (5 + rsel) * GS_TAB_K);      // the tile's two V rows: slots 5 / 6
      else vv = *reinterpret_cast<const f32x4*>(vs + rsel * p.K + k0 + 4 * t4);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if constexpr (FOLD) {      // a holds Us[c][k], vv = Vs[r][k] (gs_pge_scale_uv_kernel): the folded form every exact-split kernel uses
          a[t4][t] = fmaxf(a[t4][t] + vv[t], 0.f);
        } else {
          const float x = ((a[t4][t] + vv[t]) + b0[t]) - mu1[t];
          a[t4][t] = fmaxf(fmaf(g1[t], x * is1[t], be1[t]), 0.f);
        }
      }
    }
  }
}

// One tile.  WN = 1: 128 rows, wave w owns rows [32 w, 32 w + 32) x all N columns (NT MFMA tiles, the bulk of the launch).
//            WN = 4: 32 rows, wave w owns all 32 rows x the N / 4 columns [32 NT w / 4, ...) — the LAST partial round of a
// launch runs in these: 206 116 rows are 1 611 tiles of 128 rows on 512 workgroup slots = 3.15 rounds, and the 75 tiles of the
// fourth round kept 75 CUs busy for a whole solo tile (43 us of 258) while 181 CUs idled; as 300 tiles of a quarter of the
// work they take one short round on every CU.  Same arithmetic per output element either way (same k order).
template <int MODE, int NT, int WN, bool MARG = false, bool AGEN = false, bool MGEN = false, bool FGEN = false, bool RIDX = false,
          bool BITS = false, bool FOLD = false>
__device__ __forceinline__ void panel_tile(const gs_panel_args& p, char* smem, int m0, int stat_row) {
  constexpr int PN_N = 32 * NT, PN_STAGE = PN_N * 128, NTW = NT / WN, ROWS = 128 / WN;
  static_assert(!MARG || WN == 1, "the fused-marginals form uses 128-row tiles");
  static_assert(!AGEN || MARG, "the generated dE2 operand is used by the fused-marginals form only");
  static_assert(!MGEN || MARG, "the regenerated mask belongs to the fused-marginals epilogue");
  static_assert(!FGEN || !MARG, "the generated A1 operand is used by the forward product only");
  // LDS: modes 0-2  [2 B stages of N x 128 B][bias N floats][constants table K x 8 floats (AGEN / FGEN)][2 V rows (FGEN)]
  //      mode 3     [3 B sub-stages of N x 96 B][ONE table of 256 x 8 floats: generated-operand constants in slots 0-4 (0-6),
  //                  the tile's two V rows in slots 5 / 6 (FGEN), the bias in slot 7] — 72 + 8 = 80 KB at N = 256: exactly two
  //                  workgroups per CU with a prefetch distance of two sub-stages
  constexpr int S3 = PN_N * 96, OFF_X = MODE == 3 ? 3 * S3 : 2 * PN_STAGE, BS = 1;
  float* sbias = reinterpret_cast<float*>(smem + OFF_X) + (MODE == 3 ? 7 * GS_TAB_K : 0);      // (mode 3: slot 7 of the table)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  int row = m0 + (WN == 1 ? wave * 32 : 0) + li;
  if (MARG) {       // m0 = tile id: tile (rb, cb) holds the pairs (r, c), r in [16 rb, 16 rb + 16), c in [8 cb, 8 cb + 8)
    const int rb = m0 / p.mg_ncb, cb = m0 - rb * p.mg_ncb, i = wave * 32 + li;
    row = min(16 * rb + (i >> 3), p.mg_nr - 1) * p.mg_n + min(8 * cb + (i & 7), p.mg_n - 1);
  }
  const int j0 = WN == 1 ? 0 : wave * NTW;            // first N tile of this wave
  const float* __restrict__ arow = p.A + (long)(RIDX ? p.rowidx[min(row, p.M - 1)] : min(row, p.M - 1)) * p.lda + 16 * h;
  int rsel = 0;                                        // FGEN: which of the tile's two V rows this lane's pair row uses
  if constexpr (FGEN) {     // pair row (r, c) = (row / n, row % n): the lane streams U[c]; V[r] (<= 2 distinct r per tile) sits in LDS
    const int rr = min(row, p.M - 1), r = rr / p.mg_n;
    arow = p.a1_U + (long)(rr - r * p.mg_n) * p.a1_ld + 16 * h;
    rsel = r - m0 / p.mg_n;
  }
  const char* __restrict__ Bb = (const char*)p.B;
  const int nkb = p.K >> 5;
  const int sw = (li >> 1) & 7;        // read-side swizzle of this lane's B rows (rows 32 j + li: (row >> 1) & 7 == (li >> 1) & 7)

  auto stage_b = [&](int kb, int st) {
#pragma unroll
    for (int q = 0; q < NT; ++q) {       // N * 128 B = 4 NT pieces of 1 KiB (8 rows each): NT per wave
      const int lrow = (wave * NT + q) * 8 + (lane >> 3);
      const int logical = (lane & 7) ^ ((lrow >> 1) & 7);
      const char* src = Bb + (long)lrow * p.ldb_bytes + kb * 128 + logical * 16;
      __builtin_amdgcn_global_load_lds((gs_gbl_ptr)src, (gs_lds_ptr)(smem + st * PN_STAGE + (wave * NT + q) * 1024), 16, 0, 0);
    }
  };
  auto load_a = [&](int kb, f32x4 (&dst)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4u t = *reinterpret_cast<const f32x4u*>(arow + kb * 32 + 4 * j);
      dst[j] = f32x4{t.x, t.y, t.z, t.w};
    }
  };

  unsigned long long t_start = 0, t_loop = 0;
  if (p.triage & 2) t_start = __builtin_amdgcn_s_memrealtime();
  if (p.span && tid == 0) atomicMin(p.span, (unsigned long long)__builtin_amdgcn_s_memrealtime());

  f32x16 acc[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // the bias row goes to LDS now and is read back in the epilogue: an LDS read counts on lgkmcnt, so it can sit between the
  // stores of two tiles without waiting for them (a global load there would: vmcnt retires in order), and no register is held
  // across the K loop
  if constexpr (MODE != 3) {
    if (tid < PN_N) sbias[tid] = p.bias ? p.bias[tid] : 0.f;
  }
  // AGEN: the per-column constants of the operand transform sit behind the bias row ([K <= 256][8] floats), this lane's
  // pair row has ONE upstream scalar (de3 of the pair)
  const float* __restrict__ gc = reinterpret_cast<const float*>(smem + OFF_X + (MODE == 3 ? 0 : PN_N * 4));      // tab[8][GS_TAB_K]
  float gen_d = 0.f;
  if constexpr (AGEN) gen_d = p.gen_d[min(row, p.M - 1)];
  // The [K][8] constants rows of the producers are transposed into the structure-of-arrays image tab[slot][GS_TAB_K] (see
  // panel_transform_a): thread i owns half-row i of the source (row k = i >> 1, slots 4 (i & 1) .. + 3): every float of the image
  // is written by exactly one thread.  Mode 3 also parks the tile's two V rows (slots 5, 6; FGEN) and the bias (slot 7) there.
  {
    float* tw = const_cast<float*>(gc);
    const float* __restrict__ src = AGEN ? p.gen_c : (FGEN ? p.a1_c : nullptr);
    const int r0 = FGEN ? m0 / p.mg_n : 0;
    if constexpr (AGEN || FGEN || MODE == 3) {
      for (int i = tid; i < 2 * GS_TAB_K; i += 256) {
        const int k = i >> 1, hf = i & 1;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if constexpr (AGEN || FGEN) {
          if (k < p.K) t = reinterpret_cast<const f32x4*>(src)[i];
        }
        if constexpr (MODE == 3) {
          if (hf) {
            if constexpr (FGEN) {
              if (k < p.K) {
                t[1] = p.a1_V[(long)min(r0, p.mg_nr - 1) * p.a1_ld + k];
                t[2] = p.a1_V[(long)min(r0 + 1, p.mg_nr - 1) * p.a1_ld + k];
              }
            }
            if constexpr (!MARG) t[3] = (p.bias && k < PN_N) ? p.bias[k] : 0.f;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) tw[(4 * hf + q) * GS_TAB_K + k] = t[q];
      }
    }
  }

  // FGEN (modes 0-2): the V rows of the tile's (at most two) r behind the table
  const float* __restrict__ vs = reinterpret_cast<const float*>(smem + 2 * PN_STAGE + PN_N * 4 + 256 * 32);
  if constexpr (FGEN && MODE != 3) {
    f32x4* vw = reinterpret_cast<f32x4*>(smem + 2 * PN_STAGE + PN_N * 4 + 256 * 32);
    const int kq = p.K >> 2, r0 = m0 / p.mg_n;
    for (int i = tid; i < 2 * kq; i += 256) {
      const int q = i >= kq;
      vw[i] = *reinterpret_cast<const f32x4*>(p.a1_V + (long)min(r0 + q, p.mg_nr - 1) * p.a1_ld + 4 * (i - q * kq));
    }
  }

  f32x4 a_cur[4], a_nxt[4];

  if constexpr (MODE == 3) {
    // ---- exact three-way bf16 split, six products (see the header comment of this file) -------------------------------
    // One loop step = a 16-k SUB-block: B sub-stage = N rows x 96 B = [h | m | l planes] x [k half 0 | k half 1] x 8 bf16; a
    // 32-k block of A (16 k per lane, as in the other modes) spans two sub-blocks u = 0, 1 whose MFMAs take the lane's
    // elements [8 u, 8 u + 8).  Slot of piece p in LDS row r: (p + 3 ((r >> 3) & 1)) % 6 — conflict-free 16-B reads.
    auto stage_b3 = [&](int sb, int st) {
      constexpr int PW3 = 3 * NT / 4;      // 1-KiB pieces per wave: 6 N / 64 lanes / 4 waves
#pragma unroll
      for (int q = 0; q < PW3; ++q) {
        const int g = (wave * PW3 + q) * 64 + lane;
        const int lrow = g / 6, slot = g - lrow * 6;
        const int piece = (slot + 3 * ((lrow >> 3) & 1)) % 6;
        const char* src = Bb + (long)lrow * p.ldb_bytes + sb * 96 + piece * 16;
        __builtin_amdgcn_global_load_lds((gs_gbl_ptr)src, (gs_lds_ptr)(smem + st * S3 + (wave * PW3 + q) * 1024), 16, 0, 0);
      }
    };
    const int sh3 = 3 * ((li >> 3) & 1);
    int boff[3];                           // byte offset of this lane's (plane P, k half h) piece inside its LDS row
#pragma unroll
    for (int P = 0; P < 3; ++P) boff[P] = ((2 * P + h + sh3) % 6) * 16;
    // Ring of three sub-stages, prefetch distance two: a 16-k step is ~0.7 us of MFMA time, less than the L2 -> LDS latency
    // (the one-step-ahead form ran this loop at 136 us per pass against an MFMA floor of 64).  vmcnt retires in order; the
    // issue order inside a step is [A fragments of the next 32-k block (u == 0)] then [sub-stage ss + 2], so
    //   top of step ss:   sub-stage ss has landed when at most (PW3 + (u == 1 ? 4 : 0)) newer loads are outstanding
    //   end of a u == 1 step: the A fragments (issued one step ago) have landed when at most 2 PW3 newer loads are outstanding
    // (0 once nothing newer has been issued: the last step).
    constexpr int PW3c = 3 * NT / 4;
    bf16x8 ah, am, al;
    const int nsb = 2 * nkb, nsb_run = nsb * p.krep;
    load_a(0, a_cur);
    stage_b3(0, 0);
    if (nsb_run > 1) stage_b3(1 % nsb, 1);
    int slot = 0;
    for (int ss = 0; ss < nsb_run; ++ss) {
      const int sb = ss % nsb, u = sb & 1, kb = sb >> 1;
      if (p.triage & 4) {      // measurement only (tools/triage_panel3.py): no waits, no barrier — results are garbage
      } else
      if (ss + 1 >= nsb_run) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (u == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW3c + 4) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW3c) : "memory");
      // (a bare barrier: __syncthreads() carries a workgroup-scope release fence, which the compiler lowers to vmcnt(0) and
      // would drain the two prefetched sub-stages at every step; LDS visibility of the direct-to-LDS loads is what the vmcnt
      // above waits for, this wave's own LDS reads of the stage about to be overwritten are done once lgkmcnt is 0)
      if (!(p.triage & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (u == 0 && !(p.triage & 8)) {
        if (ss + 2 < nsb_run) load_a(((sb + 2) % nsb) >> 1, a_nxt);
      }
      if (ss + 2 < nsb_run && !(p.triage & 32)) stage_b3((sb + 2) % nsb, slot == 0 ? 2 : slot - 1);      // slot of ss + 2 = (slot + 2) % 3
      if (u == 0) panel_transform_a<AGEN, FGEN, true, FOLD>(p, a_cur, gc, vs, gen_d, rsel, kb, h);
      if (!(p.triage & 16) || ss == 0)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = u == 0 ? a_cur[e >> 2][e & 3] : a_cur[2 + (e >> 2)][e & 3];
        const __bf16 hb = (__bf16)x;
        const float r1 = x - (float)hb;
        const __bf16 mb = (__bf16)r1;
        ah[e] = hb; am[e] = mb; al[e] = (__bf16)(r1 - (float)mb);
      }
      const char* __restrict__ bs3 = smem + slot * S3 + (32 * j0 + li) * 96;
      // B fragments one tile ahead of the MFMAs that consume them (the six MFMAs of a tile are 192 cycles: too short to hide an
      // LDS read issued right in front of them)
      bf16x8 bq[2][3];
#pragma unroll
      for (int P = 0; P < 3; ++P) bq[0][P] = *reinterpret_cast<const bf16x8*>(bs3 + boff[P]);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        if (j + 1 < NTW) {
#pragma unroll
          for (int P = 0; P < 3; ++P) bq[(j + 1) & 1][P] = *reinterpret_cast<const bf16x8*>(bs3 + (j + 1) * (32 * 96) + boff[P]);
        }
        const bf16x8 bh = bq[j & 1][0], bm = bq[j & 1][1], bl = bq[j & 1][2];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);     // small terms first
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
      }
      // Issue order for the scheduler, tile after tile: [2 MFMAs of tile j][3 LDS reads of tile j + 1][4 MFMAs of tile j] — the
      // reads of the NEXT tile get 128 cycles of this tile's MFMAs before the lgkmcnt(0) in front of the next tile's first MFMA
      // (the waitcnt pass always waits for ALL outstanding LDS reads here, so the reads must not be the last thing issued before
      // it).  Left to itself the scheduler grouped the tiles in pairs and put every read right in front of its first use: three
      // exposed LDS latencies per 12 MFMAs (round-3 ISA reading).
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        if (j + 1 < NTW) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
      if (u == 1) {
        if (ss + 1 < nsb_run) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW3c) : "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) a_cur[j] = a_nxt[j];
      }
      slot = slot == 2 ? 0 : slot + 1;
    }
  } else {
  stage_b(0, 0);
  load_a(0, a_cur);
  const int nkb_run = nkb * p.krep;      // krep = 1 in the product; > 1 only in the loop-rate measurement
  for (int kk = 0; kk < nkb_run; ++kk) {
    const int kb = kk % nkb;
    // B(kb) (direct-to-LDS, issued one block ago) has landed for every wave, and every wave is done reading the other stage
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kk + 1 < nkb_run) {
      stage_b((kb + 1) % nkb, (kk + 1) & 1);
      load_a((kb + 1) % nkb, a_nxt);
    }
    const char* __restrict__ bs = smem + (kk & 1) * PN_STAGE + li * 128 + j0 * 4096;
    panel_transform_a<AGEN, FGEN>(p, a_cur, gc, vs, gen_d, rsel, kb, h);
    if constexpr (MODE == 0) {
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        f32x4 b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = *reinterpret_cast<const f32x4*>(bs + j * 4096 + (((4 * h + q) ^ sw) << 4));
#pragma unroll
        for (int t = 0; t < 16; ++t)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t >> 2][t & 3], b[t >> 2][t & 3], acc[j], 0, 0, 0);
      }
    } else {
      bf16x8 ah[2], al[2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = a_cur[2 * u + (e >> 2)][e & 3];
          const __bf16 hb = (__bf16)x;
          ah[u][e] = hb;
          if constexpr (MODE == 1) al[u][e] = (__bf16)(x - (float)hb);
        }
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bs + j * 4096 + (((2 * h + u) ^ sw) << 4));
          if constexpr (MODE == 1) {
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(bs + j * 4096 + (((4 + 2 * h + u) ^ sw) << 4));
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[u], bh, acc[j], 0, 0, 0);     // small terms first
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], bl, acc[j], 0, 0, 0);
          }
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], bh, acc[j], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) a_cur[j] = a_nxt[j];
  }
  }

  if (p.triage & 2) t_loop = __builtin_amdgcn_s_memrealtime();
  if constexpr (MARG) {
    panel_epilogue_marg<NT, MGEN>(p, acc, smem, m0 / p.mg_ncb, m0 % p.mg_ncb, wave, li, h);
    return;
  }
  // epilogue: straight-line code per (mask?, interior tile?) combination — see panel_epilogue
  const bool has_stat = p.colstat != nullptr;
  const int rit = (WN == 1 ? wave * 32 : 0) + 4 * h;      // first accumulator row of this lane inside the tile
  const int col0 = 32 * j0 + li;
  const bool full = m0 + ROWS <= p.M;
  float cs1[NTW], cs2[NTW];
  if constexpr (RIDX) {
    if (full) panel_epilogue<NTW, false, true, BS, true, BITS>(p, acc, sbias, m0, rit, col0, cs1, cs2);
    else panel_epilogue<NTW, false, false, BS, true, BITS>(p, acc, sbias, m0, rit, col0, cs1, cs2);
  } else if (!BITS && p.mask) {
    if (full) panel_epilogue<NTW, true, true, BS>(p, acc, sbias, m0, rit, col0, cs1, cs2);
    else panel_epilogue<NTW, true, false, BS>(p, acc, sbias, m0, rit, col0, cs1, cs2);
  } else {
    if (full) panel_epilogue<NTW, false, true, BS, false, BITS>(p, acc, sbias, m0, rit, col0, cs1, cs2);
    else panel_epilogue<NTW, false, false, BS, false, BITS>(p, acc, sbias, m0, rit, col0, cs1, cs2);
  }
  if (has_stat) {
    // per-tile column sums / sums of squares of the written rows: colstat[(stat_row * 2 + q) * N + col]
    __syncthreads();                               // the B stages are dead: reuse them as scratch
    float* red = reinterpret_cast<float*>(smem);   // [2][WN == 1 ? 4 waves : 1][N]
    constexpr int SL = WN == 1 ? 4 : 1;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const float a = cs1[j] + __shfl_xor(cs1[j], 32, 64), b = cs2[j] + __shfl_xor(cs2[j], 32, 64);
      if (h == 0) {
        red[(0 * SL + (WN == 1 ? wave : 0)) * PN_N + col0 + 32 * j] = a;
        red[(1 * SL + (WN == 1 ? wave : 0)) * PN_N + col0 + 32 * j] = b;
      }
    }
    __syncthreads();
    for (int c = tid; c < 2 * PN_N; c += 256) {
      const int q = c / PN_N, cc = c % PN_N;
      float t = red[(q * SL + 0) * PN_N + cc];
      if (WN == 1) t = ((t + red[(q * SL + 1) * PN_N + cc]) + red[(q * SL + 2) * PN_N + cc]) + red[(q * SL + 3) * PN_N + cc];
      p.colstat[((long)stat_row * 2 + q) * PN_N + cc] = t;
    }
  }
  if (p.span && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    atomicMax(p.span + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
  }
  if ((p.triage & 2) && p.trace && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* o = p.trace + (long)blockIdx.x * 5;
    o[0] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, all 32 bits
    o[1] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
    o[2] = t_start; o[3] = t_loop; o[4] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int MODE, int NT, bool BITS = false>     // NT = output columns / 32: 8 (N = 256) or 4 (N = 128, the Planetoid PGE width)
__global__ __launch_bounds__(256, 2) void gs_panel_gemm_kernel(gs_panel_args p) {
  constexpr int PN_N = 32 * NT, PN_STAGE = PN_N * 128;   // bytes of one K block of B in LDS: N rows x 128 B
  // (ONE __shared__ object: a second one beside a direct-to-LDS staging array can make hipcc drain vmcnt before every
  // ds_read, cdna_hip_programming.md section 5 "three .s-level traps" (a)) — the bias row lives behind the two B stages
  __shared__ __attribute__((aligned(16))) char smem[MODE == 3 ? 3 * PN_N * 96 + 8192 : 2 * PN_STAGE + PN_N * 4];
  const int b = blockIdx.x;
  if (b < p.n_full) panel_tile<MODE, NT, 1, false, false, false, false, false, BITS>(p, smem, b * PN_BM, b);
  else panel_tile<MODE, NT, 4, false, false, false, false, false, BITS>(p, smem, p.n_full * PN_BM + (b - p.n_full) * 32, b);
}

// row-list form: C[rowidx[i]] = A[rowidx[i]] B^T (+ bias)(relu) for i < *m_dev; the grid is sized for the largest possible count and
// the tiles beyond the device-side count leave at once (128-row tiles throughout: the count is not known when the grid is sized)
template <int MODE, int NT, bool BITS = false>
__global__ __launch_bounds__(256, 2) void gs_panel_rows_kernel(gs_panel_args pin) {
  constexpr int PN_N = 32 * NT, PN_STAGE = PN_N * 128;
  __shared__ __attribute__((aligned(16))) char smem[MODE == 3 ? 3 * PN_N * 96 + 8192 : 2 * PN_STAGE + PN_N * 4];
  gs_panel_args p = pin;
  p.M = min(*pin.m_dev, pin.M);
  const int m0 = blockIdx.x * PN_BM;
  if (m0 >= p.M) return;
  panel_tile<MODE, NT, 1, false, false, false, false, true, BITS>(p, smem, m0, blockIdx.x);
}

template <int MODE, int NT, bool AGEN, bool MGEN>
__global__ __launch_bounds__(256, 2) void gs_panel_marg_kernel(gs_panel_args p) {
  constexpr int PN_N = 32 * NT, PN_STAGE = PN_N * 128;
  __shared__ __attribute__((aligned(16))) char smem[MODE == 3 ? 3 * PN_N * 96 + 8192 : 2 * PN_STAGE + PN_N * 4 + (AGEN ? 256 * 32 : 0)];
  panel_tile<MODE, NT, 1, true, AGEN, MGEN>(p, smem, blockIdx.x, 0);
}

// forward product with the generated A1 operand: E2 = A1(U, V) W1^T + b1 (+ BN-2 column sums); same tiling as gs_panel_gemm_kernel
template <int MODE, int NT, bool FOLD = false>
__global__ __launch_bounds__(256, 2) void gs_panel_fwdgen_kernel(gs_panel_args p) {
  constexpr int PN_N = 32 * NT, PN_STAGE = PN_N * 128;
  __shared__ __attribute__((aligned(16))) char smem[MODE == 3 ? 3 * PN_N * 96 + 8192 : 2 * PN_STAGE + PN_N * 4 + 256 * 32 + 2 * 256 * 4];
  const int b = blockIdx.x;
  if (b < p.n_full) panel_tile<MODE, NT, 1, false, false, false, true, false, false, FOLD>(p, smem, b * PN_BM, b);
  else panel_tile<MODE, NT, 4, false, false, false, true, false, false, FOLD>(p, smem, p.n_full * PN_BM + (b - p.n_full) * 32, b);
}

// gen_c[k][8] from the BatchNorm-2 / layer-3 quantities (one thread per column)
__global__ void gs_panel_gen_consts_kernel(const float* __restrict__ mu2, const float* __restrict__ is2, const float* __restrict__ g2,
                                           const float* __restrict__ be2, const float* __restrict__ w2, const float* __restrict__ dbe2,
                                           const float* __restrict__ dg2, float invN, int K, float* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  gs_panel_gen_row(mu2[k], is2[k], g2[k], be2[k], w2[k], dbe2[k], dg2[k], invN, out + 8 * k);
}

// W [N][ldb] fp32 -> packed [N][K/32][hi 32 | lo 32] bf16 (lo = bf16(w - hi); all zeros are kept exact)
__global__ void gs_panel_split_b_kernel(const float* __restrict__ W, long ldb, int N, int K, __bf16* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (row, k)
  if (idx >= (long)N * K) return;
  const int n = (int)(idx / K), k = (int)(idx % K);
  const float w = W[(long)n * ldb + k];
  const __bf16 hb = (__bf16)w;
  const __bf16 lb = (__bf16)(w - (float)hb);
  __bf16* o = out + ((long)n * (K >> 5) + (k >> 5)) * 64;
  o[k & 31] = hb;
  o[32 + (k & 31)] = lb;
}

// W [N][ldb] fp32 -> the image of the exact three-way split: [N][K/16 sub-blocks][plane h, m, l][k half 0, 1][8] bf16, where the
// sub-block of k is 2 (k >> 5) + ((k >> 3) & 1), its half (k >> 4) & 1, its element k & 7 (the order panel_tile<3, ...> consumes).
// h = bf16(w), m = bf16(w - h), l = bf16(w - h - m): 8 + 8 + 8 significant bits = the 24 of an fp32, so h + m + l == w exactly.
__global__ void gs_panel_split_b3_kernel(const float* __restrict__ W, long ldb, int N, int K, __bf16* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (row, k)
  if (idx >= (long)N * K) return;
  const int n = (int)(idx / K), k = (int)(idx % K);
  gs_split3_store(W[(long)n * ldb + k], n, k, K, out);
}

// pre-split image of B for the operand mode (no-op for exact fp32); returns the row pitch in bytes
// (b_ready: the caller has already written the three-way split image of W into `scratch` on this stream — gs_pge_bn1_fold_kernel,
// gs_pge_bwd_prep_kernel — so the stand-alone split launch is skipped; mode 3 only)
static long panel_prepare_b(const float* W, long ldw, int N, int K, int prec, void* scratch, hipStream_t s, const void** Bout,
                            bool b_ready = false) {
  if (prec == 0) { *Bout = W; return ldw * 4; }
  if (prec == 3 && b_ready) { *Bout = scratch; return (long)K * 6; }
  const dim3 grid(gs_cdiv((long)N * K, 256)), block(256);
  *Bout = scratch;
  if (prec == 3) {
    hipLaunchKernelGGL(gs_panel_split_b3_kernel, grid, block, 0, s, W, ldw, N, K, (__bf16*)scratch);
    return (long)K * 6;
  }
  hipLaunchKernelGGL(gs_panel_split_b_kernel, grid, block, 0, s, W, ldw, N, K, (__bf16*)scratch);
  return (long)K * 4;
}

// the tall (256-row, persistent) form of the exact-split products at N = 256: gs_panel2.hip
int gs_tall_plain(const float* A, long lda, const void* Bimg, long ldb_bytes, const float* bias, int relu, float* C, long ldc, long c_rows,
                  int M, int K, const int* rowidx, const int* m_dev, long a_rows, hipStream_t s);
int gs_tall_forward_a1(const float* Us, const float* Vs, long ld, int n, int nr, const void* Bimg, long ldb_bytes,
                       const float* bias, float* C, long ldc, int K, float* colstat, int* stat_slabs, unsigned long long* span,
                       int alone, hipStream_t s);
int gs_tall_marginals(const float* E2, const void* Bimg, long ldb_bytes, const float* gen_table, const float* de3, const float* Us,
                      const float* Vs, long ld, int n, int nr, int K, float* partU, float* partV, hipStream_t s);

static int g_panel_krep = 1, g_panel_triage = 0;
extern "C" int gs_panel_debug(int krep, int triage) {      // measurement hook (tools/ab_panel.py), not part of the product ABI
  g_panel_krep = krep < 1 ? 1 : krep;
  g_panel_triage = triage;
  return GS_OK;
}
static int g_panel_enabled = 1;

extern "C" int gs_gemm_set_panel(int enable) {
  g_panel_enabled = enable != 0;
  return GS_OK;
}

// does gs_gemm_f32 hand this product to the panel kernel?
bool gs_panel_eligible(const gs_gemm_desc& p) {
  if (!g_panel_enabled) return false;
  if ((p.N != 256 && p.N != 128) || p.K < 32 || (p.K & 31)) return false;
  if (p.M < (p.prec != 0 ? 1 : 2048)) return false;   // exact fp32: a routing choice for big M; a split mode asked for: always honoured
  if (p.nb1 > 1 || p.nb2 > 1 || p.kchunk > 0 || p.offA || p.offB || p.offC || p.Mb || p.Kb) return false;
  if (p.gatherA_m || p.gatherA_k || p.gatherB_k || p.bgen_mask) return false;
  if (p.sAk != 1 || p.sBk != 1 || p.alpha != 1.f || p.beta != 0.f) return false;
  if ((p.sAm & 3) || (p.sBn & 3) || (((uintptr_t)p.A | (uintptr_t)p.B) & 15)) return false;
  if (p.prec < 0 || p.prec > 3) return false;
  if (p.prec != 0 && !p.scratch) return false;
  return true;
}

static int g_panel_fwd_pad = 0;     // gs_panel_set_fwd_lds_pad
extern "C" int gs_panel_set_fwd_lds_pad(int bytes) {
  g_panel_fwd_pad = bytes < 0 ? 0 : bytes;
  return GS_OK;
}
static int g_panel_slots = 0;       // resident workgroups per launch: 2 per CU (registers: 2 waves / SIMD; LDS: 2 x 65 KB)
static int g_panel_narrow = 1;      // finish the last partial round in 32-row tiles (0: 128-row tiles throughout)

extern "C" int gs_gemm_set_panel_narrow_tail(int on) {
  g_panel_narrow = on != 0;
  return GS_OK;
}

// 128-row tiles for every FULL round of the chip, 32-row tiles for what is left
static void panel_split(int M, int* n_full, int* n_narrow) {
  if (g_panel_slots == 0) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    g_panel_slots = 2 * (cus > 0 ? cus : 256);
  }
  const int tiles = gs_cdiv(M, PN_BM);
  const int nf = g_panel_narrow ? (tiles / g_panel_slots) * g_panel_slots : tiles;   // (a 128-row tile may be the ragged last one)
  const long left = (long)M - (long)nf * PN_BM;
  *n_full = nf;
  *n_narrow = left > 0 ? gs_cdiv(left, 32) : 0;
}

int gs_panel_tiles(int M) {
  int nf, nn;
  panel_split(M, &nf, &nn);
  return nf + nn;
}

template <int NT>
static void panel_dispatch(const gs_gemm_desc& d, gs_panel_args& a, hipStream_t s) {
  int nf, nn;
  panel_split(d.M, &nf, &nn);
  a.n_full = nf;
  const dim3 grid(nf + nn), block(256);
  a.ldb_bytes = panel_prepare_b(d.B, d.sBn, d.N, d.K, d.prec, d.scratch, s, &a.B);
  if constexpr (NT == 8) {
    if (d.prec == 3 && !a.bits && !a.mask && !a.colstat && g_panel_krep == 1 && !g_panel_triage &&
        gs_tall_plain(a.A, a.lda, a.B, a.ldb_bytes, a.bias, a.relu, a.C, a.ldc, a.M, a.M, a.K, nullptr, nullptr, a.M, s) == GS_OK)
      return;
    if (a.bits) {      // the sign-bit epilogue is a form of its own (N = 256, no mask: gs_panel_launch checks)
      if (d.prec == 0) hipLaunchKernelGGL((gs_panel_gemm_kernel<0, NT, true>), grid, block, 0, s, a);
      else if (d.prec == 1) hipLaunchKernelGGL((gs_panel_gemm_kernel<1, NT, true>), grid, block, 0, s, a);
      else if (d.prec == 2) hipLaunchKernelGGL((gs_panel_gemm_kernel<2, NT, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((gs_panel_gemm_kernel<3, NT, true>), grid, block, 0, s, a);
      return;
    }
  }
  if (d.prec == 0) hipLaunchKernelGGL((gs_panel_gemm_kernel<0, NT>), grid, block, 0, s, a);
  else if (d.prec == 1) hipLaunchKernelGGL((gs_panel_gemm_kernel<1, NT>), grid, block, 0, s, a);
  else if (d.prec == 2) hipLaunchKernelGGL((gs_panel_gemm_kernel<2, NT>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((gs_panel_gemm_kernel<3, NT>), grid, block, 0, s, a);
}

int gs_panel_launch(const gs_gemm_desc& d, hipStream_t s) {
  gs_panel_args a;
  memset(&a, 0, sizeof(a));
  a.A = d.A; a.lda = d.sAm;
  a.C = d.C; a.ldc = d.ldc;
  a.M = d.M; a.K = d.K;
  a.bias = d.bias; a.relu = d.relu; a.mask = d.mask; a.ldmask = d.ldmask; a.colstat = d.colstat;
  if (d.signbits && (d.N != 256 || d.mask)) GS_FAIL(GS_ERR_UNSUPPORTED, "gs_gemm_f32: signbits needs N = 256 and no mask");
  a.bits = d.signbits; a.ldbits = d.ld_signbits;
  a.krep = g_panel_krep; a.triage = g_panel_triage;
  a.trace = (g_panel_triage & 2) ? (unsigned long long*)d.scratch : nullptr;   // (trace runs use prec = 0: scratch is free)
  if (d.N == 256) panel_dispatch<8>(d, a, s);
  else panel_dispatch<4>(d, a, s);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// engine-internal: C[rowidx[i]][N] = relu?(A[rowidx[i]][K] W^T + bias) for i < *m_dev (device count, at most m_max): the row-list
// form of the plain product.  GS_ERR_UNSUPPORTED when the shape is not covered (the caller then computes every row).
int gs_panel_rows(const float* A, long lda, const float* W, long ldw, const float* bias, int relu, float* Cout, long ldc, int N, int K,
                  const int* rowidx, const int* m_dev, int m_max, long rows_total, int prec, void* scratch, unsigned* bits, long ldbits,
                  hipStream_t s) {
  if (!g_panel_enabled || !rowidx || !m_dev || m_max < 1) return GS_ERR_UNSUPPORTED;
  if ((N != 256 && N != 128) || K < 32 || (K & 31) || (lda & 3) || (ldw & 3) || prec < 0 || prec > 3 || (prec != 0 && !scratch)) return GS_ERR_UNSUPPORTED;
  if ((double)rows_total * (double)ldc >= 4294967296.0 || (((uintptr_t)A | (uintptr_t)W) & 15)) return GS_ERR_UNSUPPORTED;
  gs_panel_args a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.C = Cout; a.ldc = ldc; a.M = m_max; a.K = K; a.bias = bias; a.relu = relu; a.krep = 1;
  a.rowidx = rowidx; a.m_dev = m_dev;
  a.bits = bits; a.ldbits = ldbits;
  a.n_full = gs_cdiv(m_max, PN_BM);
  const dim3 grid(a.n_full), block(256);
  a.ldb_bytes = panel_prepare_b(W, ldw, N, K, prec, scratch, s, &a.B);
  if (bits && N != 256) return GS_ERR_UNSUPPORTED;
  if (prec == 3 && N == 256 && !bits &&
      gs_tall_plain(A, lda, a.B, a.ldb_bytes, bias, relu, Cout, ldc, rows_total, m_max, K, rowidx, m_dev, rows_total, s) == GS_OK)
    return GS_OK;
#define GS_ROWS_LAUNCH(MODE_)                                                                                  \
  do {                                                                                                         \
    if (N == 256 && bits) hipLaunchKernelGGL((gs_panel_rows_kernel<MODE_, 8, true>), grid, block, 0, s, a);    \
    else if (N == 256) hipLaunchKernelGGL((gs_panel_rows_kernel<MODE_, 8>), grid, block, 0, s, a);             \
    else hipLaunchKernelGGL((gs_panel_rows_kernel<MODE_, 4>), grid, block, 0, s, a);                           \
  } while (0)
  if (prec == 0) GS_ROWS_LAUNCH(0);
  else if (prec == 1) GS_ROWS_LAUNCH(1);
  else if (prec == 2) GS_ROWS_LAUNCH(2);
  else GS_ROWS_LAUNCH(3);
#undef GS_ROWS_LAUNCH
  GS_LAUNCH_CHECK();
  return GS_OK;
}

static int g_panel_marg = 2;
// measurement knob: 0 = materialise dY1 and run the marginals kernel, 1 = fused marginals on a materialised dE2,
// 2 (default) = fused marginals with dE2 generated from E2 in the operand path where it applies
extern "C" int gs_gemm_set_panel_marg(int level) {
  g_panel_marg = level < 0 ? 0 : (level > 2 ? 2 : level);
  return GS_OK;
}

// engine-internal: partU / partV of (dE2 [nr * n][K] x W^T [N][K]) * [mask > 0] without materialising the product.
// Returns GS_ERR_UNSUPPORTED when the shape is not covered (the caller then takes the two-kernel route).
// With `gen` (7 device arrays mu2, is2, g2, be2, w2, dbe2, dg2 of length K, de3 per pair row, total pair rows N_total and an
// [K][8] float scratch table) the first operand is E2 and dE2 is formed inside the kernel (no dE2 pass, no dE2 buffer).
// (table_ready: the caller has filled `table` already — the engine's gs_pge_bwd_prep_kernel does, saving a launch)
struct gs_panel_gen { const float *mu2, *is2, *g2, *be2, *w2, *dbe2, *dg2, *de3; long total_rows; float* table; int table_ready; };
// Layer-1 activations as a function of U [n][ld], V [n][ld] (V + r_lo * ld is passed), the layer-0 bias and the BatchNorm-1
// quantities (5 arrays of the hidden width) and the [width][8] float constants table {b0, mu1, is1, g1, be1, 0, 0, 0} per column,
// which the CALLER fills (the engine's gs_pge_bn1_stats_kernel writes it next to mu1 / is1): with `a1` the marginals' [A1 > 0]
// mask is recomputed from them (`mask` may be null), and gs_panel_forward_a1 forms the A operand of E2 = A1 W1^T + b1 from them.
struct gs_panel_a1 { const float *U, *V; long ld; const float *b0, *mu1, *is1, *g1, *be1; float* table; const float *Us, *Vs; int b_ready; };      // Us, Vs: the folded
      // terms (gs_pge_scale_uv_kernel) or null; b_ready: `scratch` already holds the three-way split image of the product's W

static int g_panel_a1gen = 1;
// measurement knob: 0 = always materialise A1 (round-2 first form), 1 (default) = generate it in the operand / mask path
extern "C" int gs_gemm_set_panel_a1gen(int on) {
  g_panel_a1gen = on != 0;
  return GS_OK;
}

// engine-internal: E2 [nr * n][N] = A1 W^T + bias with A1 generated from `a1` (K = its width), column statistics per tile as in
// gs_gemm_desc.colstat ([gs_panel_tiles(nr * n)][2][N]).  GS_ERR_UNSUPPORTED when the shape is not covered (the caller then
// materialises A1 and takes the plain product).
int gs_panel_forward_a1(const gs_panel_a1* a1, int n, int nr, const float* W, long ldw, const float* bias, float* C, long ldc, int N,
                        int K, float* colstat, int* stat_slabs, int prec, void* scratch, unsigned long long* span, int alone,
                        hipStream_t s) {
  if (!g_panel_a1gen || !g_panel_enabled || !a1 || !a1->table || n < 128 || nr < 1) return GS_ERR_UNSUPPORTED;   // (<= 2 r per tile)
  if ((N != 256 && N != 128) || K < 32 || K > 256 || (K & 31) || (ldw & 3) || (a1->ld & 3)) return GS_ERR_UNSUPPORTED;
  if ((double)nr * n * (double)ldc >= 4294967296.0 || (prec != 0 && !scratch) || prec < 0 || prec > 3) return GS_ERR_UNSUPPORTED;
  if ((((uintptr_t)a1->U | (uintptr_t)a1->V | (uintptr_t)W) & 15)) return GS_ERR_UNSUPPORTED;
  gs_panel_args a;
  memset(&a, 0, sizeof(a));
  a.M = nr * n; a.K = K; a.C = C; a.ldc = ldc; a.bias = bias; a.colstat = colstat; a.krep = g_panel_krep; a.triage = g_panel_triage & ~2;
  a.mg_n = n; a.mg_nr = nr;
  a.a1_U = a1->U; a.a1_V = a1->V; a.a1_ld = a1->ld; a.a1_c = a1->table;
  a.A = a1->U; a.lda = a1->ld;
  a.span = span;
  int nf, nn;
  panel_split(a.M, &nf, &nn);
  a.n_full = nf;
  const dim3 grid(nf + nn), block(256);
  a.ldb_bytes = panel_prepare_b(W, ldw, N, K, prec, scratch, s, &a.B, a1->b_ready != 0);
  if (stat_slabs) *stat_slabs = nf + nn;
  if (prec == 3 && N == 256 && g_panel_krep == 1 && !a.triage &&
      gs_tall_forward_a1(a1->Us, a1->Vs, a1->ld, n, nr, a.B, a.ldb_bytes, bias, C, ldc, K, colstat, stat_slabs, span, alone, s) == GS_OK)
    return GS_OK;
  if (prec == 3 && N == 256 && a1->Us && a1->Vs) {      // the folded form (what the 256-row kernels and gs_dw1_generated use): same bits everywhere
    a.a1_U = a1->Us; a.a1_V = a1->Vs; a.A = a1->Us;
    // (measurement knob gs_panel_set_fwd_lds_pad: dynamic LDS added to the TRAINING pass's launch — 80 KB static + pad > 80 KB leaves ONE
    // workgroup per CU, i.e. half of the register file and LDS free for the real-side stream's kernels running beside it)
    hipLaunchKernelGGL((gs_panel_fwdgen_kernel<3, 8, true>), grid, block, alone ? 0 : g_panel_fwd_pad, s, a);
    GS_LAUNCH_CHECK();
    return GS_OK;
  }
#define GS_FWD_LAUNCH(MODE_)                                                                    \
  do {                                                                                          \
    if (N == 256) hipLaunchKernelGGL((gs_panel_fwdgen_kernel<MODE_, 8>), grid, block, 0, s, a); \
    else hipLaunchKernelGGL((gs_panel_fwdgen_kernel<MODE_, 4>), grid, block, 0, s, a);          \
  } while (0)
  if (prec == 0) GS_FWD_LAUNCH(0);
  else if (prec == 1) GS_FWD_LAUNCH(1);
  else if (prec == 2) GS_FWD_LAUNCH(2);
  else GS_FWD_LAUNCH(3);
#undef GS_FWD_LAUNCH
  GS_LAUNCH_CHECK();
  return GS_OK;
}

int gs_panel_marginals(const float* dE2, long lda, const float* W, long ldw, const float* mask, long ldmask, int n, int nr, int N,
                       int K, float* partU, float* partV, int prec, void* scratch, const gs_panel_gen* gen, const gs_panel_a1* a1,
                       hipStream_t s) {
  const bool b_ready = a1 && a1->b_ready;
  if (a1 && (!g_panel_a1gen || !a1->table || (a1->ld & 3))) a1 = nullptr;
  if (!a1 && !mask) return GS_ERR_UNSUPPORTED;
  // (generated operand: modes 0 and 3 at N = 256 — the up-front bf16 conversions of modes 1 and 2 next to it need more than 256
  // registers there — any mode at N = 128)
  if (gen && (g_panel_marg < 2 || K > 256 || !gen->table || (N == 256 && (prec == 1 || prec == 2)))) return GS_ERR_UNSUPPORTED;
  if (!g_panel_marg || !g_panel_enabled || (N != 256 && N != 128) || K < 32 || (K & 31) || (lda & 3) || (ldw & 3)) return GS_ERR_UNSUPPORTED;
  if ((double)nr * n * (double)ldmask >= 4294967296.0 || (prec != 0 && !scratch) || nr < 1) return GS_ERR_UNSUPPORTED;
  if ((((uintptr_t)dE2 | (uintptr_t)W) & 15)) return GS_ERR_UNSUPPORTED;
  gs_panel_args a;
  memset(&a, 0, sizeof(a));
  a.A = dE2; a.lda = lda; a.M = nr * n; a.K = K;
  a.mask = mask; a.ldmask = ldmask;
  a.krep = 1;
  a.mg_n = n; a.mg_nr = nr; a.mg_ncb = gs_cdiv(n, 8);
  a.partU = partU; a.partV = partV;
  if (gen) {
    if (!gen->table_ready)
      hipLaunchKernelGGL(gs_panel_gen_consts_kernel, dim3(gs_cdiv(K, 64)), dim3(64), 0, s, gen->mu2, gen->is2, gen->g2, gen->be2, gen->w2,
                         gen->dbe2, gen->dg2, 1.f / (float)gen->total_rows, K, gen->table);
    a.gen_c = gen->table; a.gen_d = gen->de3;
  }
  if (a1) {
    a.a1_U = a1->U; a.a1_V = a1->V; a.a1_ld = a1->ld; a.a1_c = a1->table;
    if (prec == 3 && N == 256 && a1->Us && a1->Vs) { a.a1_U = a1->Us; a.a1_V = a1->Vs; a.a1_fold = 1; }
  }
  const dim3 grid((unsigned)(gs_cdiv(nr, 16) * a.mg_ncb)), block(256);
  a.ldb_bytes = panel_prepare_b(W, ldw, N, K, prec, scratch, s, &a.B, b_ready);
  if (prec == 3 && N == 256 && gen && a1 && lda == K &&
      gs_tall_marginals(dE2, a.B, a.ldb_bytes, a.gen_c, a.gen_d, a1->Us, a1->Vs, a1->ld, n, nr, K, partU, partV, s) == GS_OK)
    return GS_OK;
#define GS_MARG_LAUNCH(MODE_)                                                                                        \
  do {                                                                                                               \
    if (N == 256 && gen && a1) hipLaunchKernelGGL((gs_panel_marg_kernel<MODE_, 8, (MODE_ == 0 || MODE_ == 3), true>), grid, block, 0, s, a); \
    else if (N == 256 && gen) hipLaunchKernelGGL((gs_panel_marg_kernel<MODE_, 8, (MODE_ == 0 || MODE_ == 3), false>), grid, block, 0, s, a); \
    else if (N == 256 && a1) hipLaunchKernelGGL((gs_panel_marg_kernel<MODE_, 8, false, true>), grid, block, 0, s, a);        \
    else if (N == 256) hipLaunchKernelGGL((gs_panel_marg_kernel<MODE_, 8, false, false>), grid, block, 0, s, a);             \
    else if (gen && a1) hipLaunchKernelGGL((gs_panel_marg_kernel<MODE_, 4, true, true>), grid, block, 0, s, a);              \
    else if (gen) hipLaunchKernelGGL((gs_panel_marg_kernel<MODE_, 4, true, false>), grid, block, 0, s, a);                   \
    else if (a1) hipLaunchKernelGGL((gs_panel_marg_kernel<MODE_, 4, false, true>), grid, block, 0, s, a);                    \
    else hipLaunchKernelGGL((gs_panel_marg_kernel<MODE_, 4, false, false>), grid, block, 0, s, a);                           \
  } while (0)
  if (prec == 0) GS_MARG_LAUNCH(0);
  else if (prec == 1) GS_MARG_LAUNCH(1);
  else if (prec == 2) GS_MARG_LAUNCH(2);
  else GS_MARG_LAUNCH(3);
#undef GS_MARG_LAUNCH
  GS_LAUNCH_CHECK();
  return GS_OK;
}
