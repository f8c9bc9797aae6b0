// "Tall" row-panel product: the exact three-way bf16 split (operand mode 3) of the big dense contractions at N = 256,
// rebuilt around what round 3's triage of gs_panel.hip isolated (profiles/r03/triage_panel3.txt; VERDICT r03 "next" 1):
//
//     PLAIN  C[M x 256]  = relu?(A[M x K] W^T + bias)                 H1 = relu(X W1 + b1)          (models/sgc.py:37-43)
//     FGEN   E2          = A1(Us, Vs) W1^T + b1 (+ BN-2 column sums)  PGE layer 2, A1 generated      (parametrized_adj.py:57-64)
//     MARG   partU/partV = marginals of (dE2(E2) W1) * [A1(U, V) > 0] its backward, dY1 never stored (autograd of the same layer)
//
// Same arithmetic per output element as panel_tile<3, 8, ...> of gs_panel.hip (same 16-k sub-blocks in the same order, same six
// products small-terms-first): the products are BITWISE those of gs_panel.hip (tests/test_gpu_panel.py).  What is different:
//   * ONE persistent workgroup of 8 waves per compute unit walks a contiguous share of the rows; a tile is 256 rows x 256
//     columns (wave w: rows [32 w, 32 w + 32) x all columns, 128 accumulator registers), so the 393-KB split image of B is
//     staged through LDS once per 256 rows instead of once per 128 and there is no partial last round: a share is cut into
//     256-row tiles plus 128-row (4 x 2 waves) / 64-row (2 x 4 waves) tiles.
//   * A arrives in FULL 128-byte lines: every wave stages its own 32 rows x 32 k block by four direct-to-LDS loads (8 rows x
//     128 B each, XOR-swizzled on the source address), wave-private, and reads its MFMA fragments from LDS.  The old form loaded
//     fragment-shaped (each lane 64 B of its own row: 32 lines per wave instruction, a quarter of each used).
//   * Operands one sub-step ahead, the two waves of a SIMD in opposite order.  All 8 waves share one barrier per 16-k
//     sub-step, so the two waves of a SIMD (w and w + 4) would otherwise prepare operands (LDS reads, the generated-operand
//     transform, the three-way split: ~100 VALU) at the same time and then compete for the matrix pipe at the same time.  Each
//     step therefore has two independent parts — the 48 MFMAs on the operands split during the PREVIOUS step, and [issue the
//     loads, read / transform / split the NEXT step's operands] — which waves 0-3 run in the order [prepare, MFMA] and waves
//     4-7 in the order [MFMA, prepare].
//   * The load stream never stops at a tile boundary: the ring of B sub-stages runs on through the epilogue and the first two
//     A blocks (+ V rows / per-row scalars) of the NEXT tile are requested during the current tile's last sub-steps.  vmcnt
//     retires in order and every wait is an exact count: at the top of step s everything issued up to step s - 2 has landed
//     when at most (operations issued during step s - 1 [+ the stores of the epilogue in between]) are outstanding.  Stores
//     of rows beyond M go through buffer stores with an out-of-range offset (always issued, dropped by the bounds check) so
//     that the counts hold.
//   * The epilogue turns the accumulators (column on the lane, rows in registers) into row segments through a wave-private LDS
//     image and stores 16 B per lane (8 x 128-byte full lines per instruction); BN-2 column sums are accumulated per workgroup
//     ([grid][2][256] partials instead of one per tile).  Its LDS scratch is the B sub-stage consumed last (free until the next
//     tile's first step refills it).
//   * Workgroups start their shares with 0, 64, 128 or 192 rows' worth of short tiles (by workgroup id), so the tile boundaries —
//     the store bursts — of the chip are spread over four phases instead of coinciding.
// LDS (one object, all 160 KB): [3 B sub-stages x 24 KB][A: 2 buffers x 8 waves x 4 KB][12 table rows x 1 KB][per-row scalars 4 KB]
// [8 KB epilogue scratch of waves 6, 7 (MARG)].
#include "gs_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* gs_lds_ptr;
typedef __attribute__((address_space(1))) const void* gs_gbl_ptr;

#define TP_SLOT 24576                      // one 16-k sub-stage of B's split image: 256 rows x 96 B
#define TP_OFF_A (3 * TP_SLOT)             // A staging: [2 buffers][8 waves][32 rows x 128 B]
#define TP_OFF_T (TP_OFF_A + 65536)        // 12 table rows of 256 floats
#define TP_OFF_G (TP_OFF_T + 12 * 1024)    // MARG: upstream scalar of each lane's pair row, [2 parities][8 waves][64 lanes]
#define TP_OFF_X (TP_OFF_G + 4096)         // spare 8 KB
#define TP_LDS (TP_OFF_X + 8192)           // = 163 840
#define TP_TAB(s) (TP_OFF_T + (s) * 1024)

enum { TP_PLAIN = 0, TP_FGEN = 1, TP_MARG = 2 };

struct gs_tall_args {
  const float* A; long lda;              // PLAIN: rows of A; MARG: E2 (this rank's pair rows), lda = K
  const void* B; long ldb_bytes;         // split image [256][K/16][h, m, l][k half][8] bf16 (gs_panel_split_b3_kernel)
  float* C; long ldc; long c_rows;       // output rows addressable through C (bounds of the buffer stores)
  int M, K;
  const float* bias; int relu;
  const int* rowidx; const int* m_dev;   // PLAIN row-list form: row i reads A[rowidx[i]], writes C[rowidx[i]]; count on the device
  float* colstat;                        // FGEN: [gridDim.x][2][256] column sums / sums of squares of this workgroup's rows
  const float* a1_U; const float* a1_V; long a1_ld; int n, nr;      // FGEN / MARG: the FOLDED layer-1 terms Us [n][ld], Vs (offset to this rank's first r):
                                                                    // A1[(r, c)] = max(Us[c] + Vs[r], 0) (gs_pge_scale_uv_kernel)
  const float* gen_c; const float* gen_d; float* partU; float* partV; int H, ncb;      // MARG: H = ceil(nr / 16) half units per c block
  unsigned long long* span;              // optional device-side launch span (gs_gcond_probe)
  int krep;                              // measurement hook: K loop of every tile repeated krep times (results scale by krep)
  int triage;                            // measurement hook (results are garbage): 1 no epilogue, 2 no A staging, 4 no B staging,
                                         // 8 no waits / barriers in the K loop, 16 no MFMAs, 32 no operand preparation
  unsigned* trace;                       // TRI == 2: [workgroup < 16][2816] stamps
  int phases;                            // 1: shares start with 0 / 64 / 128 / 192 rows of short tiles by workgroup id (default)
};

struct tp_tile { int variant, m0, cb, r0; };      // variant 0: 256 rows (8 x 1 waves), 1: 128 (4 x 2), 2: 64 (2 x 4), -1: none
struct tp_iter { int u, u_hi, pre; };               // units left [u, u_hi), short tiles still to be put in front (bit 1: 2 units, bit 0: 1)

// ---- work decomposition (wave-uniform) ---------------------------------------------------------------------------------------
// row forms: units of 64 rows (a full tile = 4); MARG: units of 16 r x 8 c (a full tile = 32 r x 8 c = two units of one c block)
template <int FORM>
__device__ __forceinline__ tp_tile tp_next(const gs_tall_args& p, tp_iter& it) {
  tp_tile t;
  t.variant = -1; t.m0 = 0; t.cb = 0; t.r0 = 0;
  if (it.u >= it.u_hi) return t;
  const int rem = it.u_hi - it.u;
  if constexpr (FORM != TP_MARG) {
    t.m0 = it.u * 64;
    if ((it.pre & 2) && rem >= 2) { it.pre &= ~2; t.variant = 1; it.u += 2; }
    else if ((it.pre & 1) && rem >= 1) { it.pre = 0; t.variant = 2; it.u += 1; }
    else if (rem >= 4) { it.pre = 0; t.variant = 0; it.u += 4; }
    else if (rem >= 2) { it.pre = 0; t.variant = 1; it.u += 2; }
    else { it.pre = 0; t.variant = 2; it.u += 1; }
  } else {
    const int cb = it.u / p.H, hb = it.u - cb * p.H;
    t.cb = cb; t.r0 = 16 * hb;
    const bool single = it.pre != 0;
    it.pre = 0;
    if (!single && hb + 1 < p.H && rem >= 2) { t.variant = 0; it.u += 2; }
    else if (p.nr - 16 * hb <= 8) { t.variant = 2; it.u += 1; }
    else { t.variant = 1; it.u += 1; }
  }
  return t;
}

// per-lane source offsets (bytes from the A base) of the four 1-KiB pieces a wave stages per 32-k block: piece q = rows
// 8 q + (lane >> 3) of the wave's 32, LDS chunk (lane & 7) holds logical chunk (lane & 7) ^ ((row >> 1) & 7)
template <int FORM, bool RIDX>
__device__ __forceinline__ void tp_aoff(const gs_tall_args& p, const tp_tile& t, int wave, int lane, int M, unsigned (&aoff)[4],
                                        unsigned (&ridx)[4]) {
  const int wm = wave >> t.variant;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rt = 8 * q + (lane >> 3);
    const unsigned sw = (unsigned)(((lane & 7) ^ ((rt >> 1) & 7)) << 4);
    unsigned base;
    if constexpr (FORM == TP_MARG) {
      // pair row (r, c): r = r0 + 4 wm + q is wave-uniform — stage_a adds its term; the per-lane part is the same for the four
      // pieces up to bit 6 of the swizzle ((8 q + (lane >> 3)) >> 1 & 7 = (lane >> 4) | 4 (q & 1)), so only aoff[0] is kept
      const int c = min(8 * t.cb + (lane >> 3), p.n - 1);
      base = (unsigned)c * (unsigned)(p.K * 4);
      ridx[q] = 0;
    } else {
      const int g = min(t.m0 + 32 * wm + rt, M - 1);
      if constexpr (FORM == TP_FGEN) {
        base = (unsigned)(g % p.n) * (unsigned)(p.a1_ld * 4);
        ridx[q] = (unsigned)g;
      } else {
        const unsigned gr = RIDX ? (unsigned)p.rowidx[g] : (unsigned)g;
        base = gr * (unsigned)(p.lda * 4);
        ridx[q] = gr;
      }
    }
    aoff[q] = base + sw;
  }
}

struct tp_state {
  int gstep, slot;        // sub-block (position in B's image) of the step about to run, its LDS slot
  int abuf0;              // A buffer holding block 0 of the current tile (block kb sits in (abuf0 + kb) & 1)
  int vpar;               // which set of per-tile table rows (FGEN: V rows 5..7 or 8..10; MARG: upstream scalars) the current tile uses
  int s_prev;             // stores this wave issued in the epilogue right before the current tile (0: none)
  int tstep;              // trace instantiation: stamps written so far
};

template <int N> __device__ __forceinline__ void tp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most (BASE + s_prev) newer operations are outstanding; s_prev is one of {0, S0, S1, S2}
template <int BASE, int S0, int S1, int S2>
__device__ __forceinline__ void tp_wait_after_epilogue(int s_prev) {
  if (s_prev == 0) tp_wait_vm<BASE>();
  else if (s_prev == S0) tp_wait_vm<BASE + S0>();
  else if (s_prev == S1) tp_wait_vm<BASE + S1>();
  else tp_wait_vm<BASE + S2>();
}
// trace instantiation: low word of s_memtime -> LDS table rows 0..10 (free in the plain form), [wave >> 2][1408]
__device__ __forceinline__ void tp_stamp(char* smem, int wave, int& idx) {
  const unsigned t = (unsigned)__builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0 && idx < 1408) *reinterpret_cast<unsigned*>(smem + TP_OFF_T + ((wave >> 2) * 1408 + idx) * 4) = t;
  ++idx;
}
__device__ __forceinline__ void tp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The operands of one sub-step: this lane's 8 A elements k = 32 kb + 16 h + 8 u + e of row li, read from the wave's LDS image of the
// block, transformed (generated-operand forms: same operations in the same order as panel_transform_a of gs_panel.hip) and split into
// their three bf16 parts.
template <int FORM>
__device__ __forceinline__ void tp_prep(const char* smem, int wave, int li, int h, int buf, int kb, int u, int rsel, float gen_d, int vpar,
                                        bf16x8& ah, bf16x8& am, bf16x8& al) {
  const float* __restrict__ tab = reinterpret_cast<const float*>(smem + TP_OFF_T);
  const char* ab = smem + TP_OFF_A + (buf * 8 + wave) * 4096 + li * 128;
  const int swa = (li >> 1) & 7;
  f32x4 x[2];
  x[0] = *reinterpret_cast<const f32x4*>(ab + (((4 * h + 2 * u) ^ swa) << 4));
  x[1] = *reinterpret_cast<const f32x4*>(ab + (((4 * h + 2 * u + 1) ^ swa) << 4));
  const int k0 = 32 * kb + 16 * h + 8 * u;
  if constexpr (FORM == TP_MARG) {        // dE2 from E2 (panel_transform_a<AGEN> of gs_panel.hip: same operations, same order)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2) {      // two k at a time: the seven constant vectors of four k at once do not fit the register budget
      const float* q = tab + k0 + 2 * t2;
      const f32x2 is2 = *reinterpret_cast<const f32x2*>(q), nm = *reinterpret_cast<const f32x2*>(q + 256),
                  g2 = *reinterpret_cast<const f32x2*>(q + 512), be2 = *reinterpret_cast<const f32x2*>(q + 768),
                  c4 = *reinterpret_cast<const f32x2*>(q + 1024), c5 = *reinterpret_cast<const f32x2*>(q + 1280),
                  c6 = *reinterpret_cast<const f32x2*>(q + 1536);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int e = 2 * t2 + t;
        const float xh = fmaf(x[e >> 2][e & 3], is2[t], nm[t]);
        const float y = fmaf(g2[t], xh, be2[t]);
        const float sel = y > 0.f ? gen_d * c4[t] : 0.f;
        x[e >> 2][e & 3] = fmaf(-xh, c6[t], sel - c5[t]);
      }
    }
  }
  if constexpr (FORM == TP_FGEN) {        // A1[(r, c)] = max(Us[c] + Vs[r], 0): the folded form of gs_pge_scale_uv_kernel (gs_edge.hip)
#pragma unroll
    for (int t4 = 0; t4 < 2; ++t4) {
      const f32x4 vv = *reinterpret_cast<const f32x4*>(tab + (3 * vpar + rsel) * 256 + k0 + 4 * t4);
#pragma unroll
      for (int t = 0; t < 4; ++t) x[t4][t] = fmaxf(x[t4][t] + vv[t], 0.f);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = x[e >> 2][e & 3];
    const __bf16 hb = (__bf16)v;
    const float r1 = v - (float)hb;
    const __bf16 mb = (__bf16)r1;
    ah[e] = hb; am[e] = mb; al[e] = (__bf16)(r1 - (float)mb);
  }
}

// own-row data of a tile for this lane (row li of its wave's 32): FGEN which of the tile's V rows, MARG the upstream scalar (from LDS)
template <int FORM>
__device__ __forceinline__ void tp_own(const gs_tall_args& p, const char* smem, const tp_tile& t, int wave, int lane, int M, int par,
                                       int& rsel, float& gen_d) {
  if constexpr (FORM == TP_FGEN) {
    const int g = min(t.m0 + 32 * (wave >> t.variant) + (lane & 31), M - 1);
    rsel = g / p.n - t.m0 / p.n;
  }
  if constexpr (FORM == TP_MARG) gen_d = *reinterpret_cast<const float*>(smem + TP_OFF_G + (par * 8 + wave) * 256 + lane * 4);
}

// One tile: main loop + epilogue.  On entry (ah, am, al) are the operands of its first sub-step and (rsel, gen_d) its own-row data;
// on exit they are those of the next tile `nxt` (a harmless repeat of the current tile when there is none).
template <int FORM, int WM, int WN, bool RIDX, int TRI, int EARLY>      // EARLY: 0 [MFMA, prepare], 1 [prepare, MFMA], 2 prepare interleaved with the MFMAs
__device__ __forceinline__ void tall_tile(const gs_tall_args& p, char* smem, tp_state& S, const tp_tile& cur, const tp_tile& nxt,
                                          const unsigned (&aoff)[4], const unsigned (&aoff_n)[4], const unsigned (&ridx)[4], int M,
                                          float& wgstat, int& rsel, float& gen_d, bf16x8& ah, bf16x8& am, bf16x8& al) {
  constexpr int NTW = 8 / WN;                        // 32-column MFMA tiles per wave
  // stores a wave issues per epilogue, by variant (full, half, quarter): the exact vmcnt counts of the next tile's first waits
  constexpr int ST0 = FORM == TP_MARG ? 18 : 32, ST1 = FORM == TP_MARG ? 9 : 16, ST2 = FORM == TP_MARG ? 5 : 8;
  constexpr int ST_ME = WN == 1 ? ST0 : (WN == 2 ? ST1 : ST2);
  constexpr int XTRA = FORM == TP_PLAIN ? 0 : 1;     // per-tile extra load (V rows / upstream scalars) issued with the next tile's block 0
  // (the thread id is made opaque per tile: what is derived from it would otherwise be hoisted out of the workgroup's tile loop and
  // kept live — or spilled — across every tile; recomputing it costs a few dozen VALU instructions per tile)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN, j0 = wn * NTW;
  const int li = lane & 31, h = lane >> 5;
  const int nsb = p.K >> 4, nkb = p.K >> 5;
  const int triage = TRI == 1 ? p.triage : 0, krep = TRI == 1 ? p.krep : 1;      // (measurement instantiation only)
  constexpr bool TRACE = TRI == 2;                 // clean product + three time stamps per step of waves 0 and 4 (tools/trace_tall.py)
  const char* __restrict__ Bb = (const char*)p.B;
  const char* __restrict__ Ab = FORM == TP_FGEN ? (const char*)p.a1_U : (const char*)p.A;
  const float* __restrict__ tab = reinterpret_cast<const float*>(smem + TP_OFF_T);

  // B staging: 24 pieces of 1 KiB per sub-stage, 3 per wave; piece order inside an LDS row (slot + 3 ((row >> 3) & 1)) % 6
  unsigned bsrc[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int g = (wave * 3 + q) * 64 + lane;
    const int lrow = g / 6, slot = g - lrow * 6;
    const int piece = (slot + 3 * ((lrow >> 3) & 1)) % 6;
    bsrc[q] = (unsigned)lrow * (unsigned)p.ldb_bytes + (unsigned)(piece * 16);
  }
  auto stage_b = [&](int sb, int slot) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
      __builtin_amdgcn_global_load_lds((gs_gbl_ptr)(Bb + bsrc[q] + (unsigned)(sb * 96)), (gs_lds_ptr)(smem + slot * TP_SLOT + (wave * 3 + q) * 1024), 16, 0, 0);
  };
  auto stage_a = [&](const unsigned (&off)[4], const tp_tile& t, int kb, int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned o = off[q];
      if constexpr (FORM == TP_MARG)
        o = (off[0] ^ (unsigned)((q & 1) << 6)) + (unsigned)(min(t.r0 + 4 * (wave >> t.variant) + q, p.nr - 1) * p.n) * (unsigned)(p.K * 4);
      __builtin_amdgcn_global_load_lds((gs_gbl_ptr)(Ab + o + (unsigned)(kb * 128)), (gs_lds_ptr)(smem + TP_OFF_A + (buf * 8 + wave) * 4096 + q * 1024), 16, 0, 0);
    }
  };
  // the per-tile extra: FGEN the (at most three) V rows of the tile's pair rows -> table rows 5 + 3 par + {0, 1, 2} (every wave issues
  // ONE piece, waves 3..7 repeat rows 0..2 — same bytes — so that the operation counts are the same in every wave); MARG the upstream
  // scalar de3 of each lane's pair row -> [par][wave][lane]
  auto stage_x = [&](const tp_tile& t, int par) {
    if constexpr (FORM == TP_FGEN) {
      const int r = min(t.m0 / p.n + wave % 3, p.nr - 1);
      __builtin_amdgcn_global_load_lds((gs_gbl_ptr)((const char*)p.a1_V + (long)r * p.a1_ld * 4 + lane * 16), (gs_lds_ptr)(smem + TP_TAB(3 * par + wave % 3)), 16, 0, 0);
    }
    if constexpr (FORM == TP_MARG) {
      const int r = min(t.r0 + 4 * (wave >> t.variant) + (li >> 3), p.nr - 1), c = min(8 * t.cb + (li & 7), p.n - 1);
      __builtin_amdgcn_global_load_lds((gs_gbl_ptr)(p.gen_d + (long)r * p.n + c), (gs_lds_ptr)(smem + TP_OFF_G + (par * 8 + wave) * 256), 4, 0, 0);
    }
  };

  f32x16 acc[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int sh3 = 3 * ((li >> 3) & 1);
  int boff[3];
#pragma unroll
  for (int P = 0; P < 3; ++P) boff[P] = ((2 * P + h + sh3) % 6) * 16;

  for (int rep = 0; rep < krep; ++rep) {
    // (krep > 1, measurement only: every pass but the last treats the tile itself as its successor)
    const bool fin = rep + 1 == krep;
    for (int kb = 0; kb < nkb; ++kb) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        // ---- everything issued up to two steps ago has landed: B of this step, the A block / table rows the NEXT step's operands
        // are read from ----------------------------------------------------------------------------------------------------------
        if (triage & 14) {
          if (!(triage & 8)) asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
          if (u == 0) {
            if (kb == 0) tp_wait_after_epilogue<7, ST0, ST1, ST2>(rep == 0 ? S.s_prev : 0);
            else if (XTRA && kb == nkb - 1) tp_wait_vm<7 + XTRA>();
            else tp_wait_vm<7>();
          } else {
            if (kb == 0) tp_wait_after_epilogue<3, ST0, ST1, ST2>(rep == 0 ? S.s_prev : 0);
            else tp_wait_vm<3>();
          }
          if (TRACE && (wave & 3) == 0) tp_stamp(smem, wave, S.tstep);      // stamp 0: arrived at the barrier (waits done)
          tp_barrier();
          if (TRACE && (wave & 3) == 0) tp_stamp(smem, wave, S.tstep);      // stamp 1: released
        }
        auto issue = [&]() {
          if (u == 1 && !(triage & 2)) {
            const int buf = (S.abuf0 + kb) & 1;        // block kb was last read during step (kb, 0): its buffer takes block kb + 2
            if (kb + 2 < nkb) stage_a(aoff, cur, kb + 2, buf);
            else if (fin) stage_a(aoff_n, nxt, kb + 2 - nkb, buf);
            else stage_a(aoff, cur, kb + 2 - nkb, buf);
            if (XTRA && kb == nkb - 2) stage_x(fin ? nxt : cur, S.vpar ^ 1);
          }
          if (!(triage & 4)) stage_b(S.gstep + 2 >= nsb ? S.gstep + 2 - nsb : S.gstep + 2, S.slot == 0 ? 2 : S.slot - 1);
        };
        bf16x8 nh, nm, nl;
        auto prep = [&]() {
          if (triage & 32) { nh = ah; nm = am; nl = al; return; }
          if (u == 0) tp_prep<FORM>(smem, wave, li, h, (S.abuf0 + kb) & 1, kb, 1, rsel, gen_d, S.vpar, nh, nm, nl);
          else if (kb + 1 < nkb) tp_prep<FORM>(smem, wave, li, h, (S.abuf0 + kb + 1) & 1, kb + 1, 0, rsel, gen_d, S.vpar, nh, nm, nl);
          else {
            // first step of the successor: its own-row data replace this tile's (nothing of this tile reads them any more)
            tp_own<FORM>(p, smem, fin ? nxt : cur, wave, lane, M, S.vpar ^ 1, rsel, gen_d);
            tp_prep<FORM>(smem, wave, li, h, (S.abuf0 + nkb) & 1, 0, 0, rsel, gen_d, S.vpar ^ 1, nh, nm, nl);
          }
        };
        auto mfma = [&]() {
          const char* __restrict__ bs3 = smem + S.slot * TP_SLOT + (32 * j0 + li) * 96;
          bf16x8 bq[2][3];
#pragma unroll
          for (int P = 0; P < 3; ++P) bq[0][P] = *reinterpret_cast<const bf16x8*>(bs3 + boff[P]);
          if (!(triage & 16) || kb == 0)
#pragma unroll
          for (int j = 0; j < NTW; ++j) {
            if (j + 1 < NTW) {
#pragma unroll
              for (int P = 0; P < 3; ++P) bq[(j + 1) & 1][P] = *reinterpret_cast<const bf16x8*>(bs3 + (j + 1) * (32 * 96) + boff[P]);
            }
            const bf16x8 bh = bq[j & 1][0], bm = bq[j & 1][1], bl = bq[j & 1][2];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);     // small terms first (as gs_panel.hip)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
          }
          // issue order per N tile: [2 MFMAs][the 3 fragment reads of the next tile][4 MFMAs] (see gs_panel.hip)
          if constexpr (EARLY == 2) {
            // interleaved form (measurement): the LDS reads of the next sub-step's operands first, then two of its VALU operations in the
            // gap behind every MFMA (an MFMA holds the SIMD's vector issue for 8 of its 32 cycles: MI355X_MICROARCH.md)
            __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
#pragma unroll
              for (int q = 0; q < 6; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                if (q == 1 && j + 1 < NTW) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
              }
            }
          } else {
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
          for (int j = 0; j < NTW; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if (j + 1 < NTW) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          }
          }
        };
        if constexpr (EARLY == 2) {
          issue();
          __builtin_amdgcn_sched_barrier(0);
          prep();
          mfma();
        } else if constexpr (EARLY == 1) {
          issue();
          if constexpr (TRACE) {      // stamp 2: the loads are issued
            __builtin_amdgcn_sched_barrier(0);
            if ((wave & 3) == 0) tp_stamp(smem, wave, S.tstep);
            __builtin_amdgcn_sched_barrier(0);
          }
          prep();
          __builtin_amdgcn_sched_barrier(0);
          if (TRACE && (wave & 3) == 0) { asm volatile("" :: "v"(nh), "v"(nm), "v"(nl)); tp_stamp(smem, wave, S.tstep); }      // stamp 3: between the parts
          mfma();
        } else {
          mfma();
          __builtin_amdgcn_sched_barrier(0);
          if (TRACE && (wave & 3) == 0) { asm volatile("" :: "v"(acc[NTW - 1])); tp_stamp(smem, wave, S.tstep); }      // stamp 2: between the parts
          issue();
          if constexpr (TRACE) {      // stamp 3: the loads are issued
            __builtin_amdgcn_sched_barrier(0);
            if ((wave & 3) == 0) tp_stamp(smem, wave, S.tstep);
            __builtin_amdgcn_sched_barrier(0);
          }
          prep();
        }
        __builtin_amdgcn_sched_barrier(0);
        ah = nh; am = nm; al = nl;
        S.gstep = S.gstep + 1 == nsb ? 0 : S.gstep + 1;
        S.slot = S.slot == 2 ? 0 : S.slot + 1;
      }
    }
    S.abuf0 = (S.abuf0 + nkb) & 1;
    S.vpar ^= 1;
  }
  S.s_prev = ST_ME;
  // the sub-stage consumed last is free until the next tile's first step refills it: the epilogue's LDS scratch (24 KB), once every
  // wave is through its last MFMAs' fragment reads
  const int fs = (S.slot == 0 ? 2 : S.slot - 1) * TP_SLOT;

  // (Lane-dependent addresses of the epilogue are loop-invariant across tiles; left visible, the compiler hoists ~35 of them out of
  // the workgroup's tile loop and keeps them live through every main loop: over the 256-register budget.  An empty asm makes the
  // thread id opaque here, so they are recomputed per tile — a few dozen VALU instructions.)
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  const int lane_e = tid_e & 63, li_e = tid_e & 31, h_e = (tid_e >> 5) & 1;
  if (triage & 1) {
    wgstat += acc[0][0];
    return;
  }
  tp_barrier();
  if constexpr (FORM != TP_MARG) {
    // ---- epilogue: bias / relu, column sums, C/D layout -> row segments through LDS, 16-byte buffer stores -------------------
    const float floor_ = p.relu ? 0.f : -__builtin_inff();
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(p.c_rows * p.ldc * 4), 0x00020000);
    unsigned crow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int g = cur.m0 + 32 * wm + 8 * i + (lane_e >> 3);
      const unsigned gr = (FORM == TP_PLAIN && RIDX) ? ridx[i] : (unsigned)g;
      crow[i] = g < M ? gr * (unsigned)(p.ldc * 4) + (unsigned)((lane_e & 7) * 16) : 0x80000000u;
    }
    bool okr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) okr[r] = cur.m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * h_e < M;
    // a wave's 32 x 32 image (4 KB): waves 0..5 in the free sub-stage, waves 6, 7 in the spare 8 KB.  LDS executes a wave's operations
    // in order, so the image of N tile j + 1 is written right behind the reads of tile j (no wait in between): only the stores wait for
    // their read data, one N tile behind the writes.
    float* T = reinterpret_cast<float*>(smem + (wave < 6 ? fs + wave * 4096 : TP_OFF_X + (wave - 6) * 4096));
    float cs1[NTW], cs2[NTW];
    f32x4 o[2][4];
    auto put = [&](int j) {
      const float bv = tab[11 * 256 + 32 * (j0 + j) + li_e];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = fmaxf(acc[j][r] + bv, floor_);
        if (FORM == TP_FGEN && okr[r]) { s1 += v; s2 = fmaf(v, v, s2); }
        T[((r & 3) + 8 * (r >> 2) + 4 * h_e) * 32 + li_e] = v;
      }
      cs1[j] = s1; cs2[j] = s2;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[j & 1][i] = *reinterpret_cast<const f32x4*>(T + (8 * i + (lane_e >> 3)) * 32 + 4 * (lane_e & 7));
    };
    put(0);
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      if (j + 1 < NTW) put(j + 1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, o[j & 1][i]), rc, crow[i] + (unsigned)(128 * (j0 + j)), 0, 0);
    }
    if constexpr (FORM == TP_FGEN) {
      // column sums of this tile: per wave -> scratch [2][32 NTW], then thread (q, col) adds the WM waves in a fixed order
      float* sw_ = T;
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const float a = cs1[j] + __shfl_xor(cs1[j], 32, 64), b = cs2[j] + __shfl_xor(cs2[j], 32, 64);
        if (h_e == 0) { sw_[32 * j + li_e] = a; sw_[32 * NTW + 32 * j + li_e] = b; }
      }
      tp_barrier();
      const int q = tid_e >> 8, col = tid_e & 255, wnc = col / (32 * NTW), lc = col - wnc * (32 * NTW);
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        const int ww = w * WN + wnc;
        t += reinterpret_cast<const float*>(smem + (ww < 6 ? fs + ww * 4096 : TP_OFF_X + (ww - 6) * 4096))[q * 32 * NTW + lc];
      }
      wgstat += t;
    }
  } else {
    // ---- fused-marginals epilogue (panel_epilogue_marg<8, MGEN> of gs_panel.hip, 8 waves): the tile's rows are the pairs
    // (r0 + (i >> 3), 8 cb + (i & 7)); in the C/D layout reg >> 2 is the r of the wave's 4, (reg & 3) + 4 h the c of the 8 -----
    const int n = p.n, nr = p.nr;
    const int rw = cur.r0 + 4 * wm;                       // first r of this wave
    bool okr[4], okc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { okr[q] = rw + q < nr; okc[q] = 8 * cur.cb + q + 4 * h_e < n; }
    unsigned uoff[4], voff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uoff[q] = (unsigned)min(8 * cur.cb + q + 4 * h_e, n - 1) * (unsigned)p.a1_ld + (unsigned)li_e;
      voff[q] = (unsigned)min(rw + q, nr - 1) * (unsigned)p.a1_ld + (unsigned)li_e;
    }
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(p.partV, 0, (int)((long)p.ncb * nr * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(p.partU, 0, (int)((long)p.H * n * 1024), 0x00020000);
    float un[4], vn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { un[q] = p.a1_U[uoff[q] + 32u * j0]; vn[q] = p.a1_V[voff[q] + 32u * j0]; }
    constexpr int RND = WN == 1 ? 2 : 1, JR = NTW / RND;      // rounds of the cross-wave sum, N tiles per round
    constexpr int CW = 32 * JR;                                // columns a wave contributes per round
    // a wave's [8 c][CW] partial sums: waves 0..5 in the free sub-stage, waves 6, 7 in the spare 8 KB
    auto red_of = [&](int w) { return reinterpret_cast<float*>(smem + (w < 6 ? fs + w * 4096 : TP_OFF_X + (w - 6) * 4096)); };
    float* red = red_of(wave);
#pragma unroll
    for (int R = 0; R < RND; ++R) {
      if (R > 0) tp_barrier();      // round 0's sums have been read
#pragma unroll
      for (int jj = 0; jj < JR; ++jj) {
        const int j = R * JR + jj;
        float uc[4], vc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { uc[q] = un[q]; vc[q] = vn[q]; }
        if (j + 1 < NTW) {
#pragma unroll
          for (int q = 0; q < 4; ++q) { un[q] = p.a1_U[uoff[q] + 32u * (j0 + j + 1)]; vn[q] = p.a1_V[voff[q] + 32u * (j0 + j + 1)]; }
        }
        const int col = 32 * (j0 + j) + li_e;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = (okr[r >> 2] && okc[r & 3] && uc[r & 3] + vc[r >> 2] > 0.f) ? acc[j][r] : 0.f;
        float sv0, sv1, sv2, sv3;
#define TP_SUMC(q) ((v[4 * (q)] + v[4 * (q) + 1]) + (v[4 * (q) + 2] + v[4 * (q) + 3]))      /* over this half-wave's 4 c */
        sv0 = TP_SUMC(0); sv1 = TP_SUMC(1); sv2 = TP_SUMC(2); sv3 = TP_SUMC(3);
#undef TP_SUMC
        sv0 += __shfl_xor(sv0, 32, 64); sv1 += __shfl_xor(sv1, 32, 64);                       // + the other 4
        sv2 += __shfl_xor(sv2, 32, 64); sv3 += __shfl_xor(sv3, 32, 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) red[(q + 4 * h_e) * CW + 32 * jj + li_e] = (v[q] + v[4 + q]) + (v[8 + q] + v[12 + q]);      // over the wave's 4 r
        // partV[cb][r][col]: half-wave 0 stores r = rw + t, half-wave 1 r = rw + 2 + t (two 128-byte lines per instruction)
        {
          const float sa = h_e ? sv2 : sv0, sb_ = h_e ? sv3 : sv1;
          const int rl = rw + 2 * h_e;
          const unsigned o0 = rl < nr ? ((unsigned)(cur.cb * nr + rl) * 256u + (unsigned)col) * 4u : 0x80000000u;
          const unsigned o1 = rl + 1 < nr ? ((unsigned)(cur.cb * nr + rl + 1) * 256u + (unsigned)col) * 4u : 0x80000000u;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, sa), rv, o0, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, sb_), rv, o1, 0, 0);
        }
      }
      tp_barrier();
      // sum over the waves of each 16-r slab in the order ((w0 + w1) + w2) + w3: thread -> (slab, c, 4 columns)
      constexpr int NSL = WM == 8 ? 2 : 1, WPS = WM >= 4 ? 4 : WM;      // slabs per tile, waves per slab
      constexpr int CT = 256 / RND;                                       // columns handled per round
      const int f = tid_e;                                                  // 512 float4 per round: NSL x 8 c x CT / 4
      const int sl = NSL == 2 ? f >> 8 : 0, ci = (f >> (NSL == 2 ? 5 : 6)) & 7, c4 = f & (CT / 4 - 1);
      const int colr = 4 * c4, wnc = colr / CW, lc = colr - wnc * CW;
      f32x4 t = *reinterpret_cast<const f32x4*>(red_of((4 * sl + 0) * WN + wnc) + ci * CW + lc);
#pragma unroll
      for (int w = 1; w < WPS; ++w) t += *reinterpret_cast<const f32x4*>(red_of((4 * sl + w) * WN + wnc) + ci * CW + lc);
      const int c = 8 * cur.cb + ci, hb = cur.r0 / 16 + sl;
      const unsigned off = (c < n && 16 * hb < nr) ? ((unsigned)(hb * n + c) * 256u + (unsigned)(R * CT + colr)) * 4u : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, t), ru, off, 0, 0);
    }
  }
}

// a workgroup's share of the units, one half of its waves (EARLY: waves 0-3)
template <int FORM, bool RIDX, int TRI, int EARLY>
__device__ __forceinline__ void tall_share(const gs_tall_args& p, char* smem, tp_iter& it, int M, int wave, int lane, float& wgstat) {
    tp_state S;
    S.gstep = 0; S.slot = 0; S.abuf0 = 0; S.vpar = 0; S.s_prev = 0; S.tstep = 0;
    tp_tile cur = tp_next<FORM>(p, it);
    unsigned aoff[4], ridx[4], aoff_n[4], ridx_n[4];
    tp_aoff<FORM, RIDX>(p, cur, wave, lane, M, aoff, ridx);
    // prologue: [A block 0, per-tile extra, B sub-stage 0] [A block 1, B sub-stage 1] — what the last two steps of a predecessor
    // tile would have issued; then the operands of the first step
    {
      const char* Ab = FORM == TP_FGEN ? (const char*)p.a1_U : (const char*)p.A;
      const char* Bb = (const char*)p.B;
      const int li = lane & 31;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned o = aoff[q];
          if constexpr (FORM == TP_MARG)
            o = (aoff[0] ^ (unsigned)((q & 1) << 6)) + (unsigned)(min(cur.r0 + 4 * (wave >> cur.variant) + q, p.nr - 1) * p.n) * (unsigned)(p.K * 4);
          __builtin_amdgcn_global_load_lds((gs_gbl_ptr)(Ab + o + st * 128), (gs_lds_ptr)(smem + TP_OFF_A + (st * 8 + wave) * 4096 + q * 1024), 16, 0, 0);
        }
        if (st == 0) {
          if constexpr (FORM == TP_FGEN) {
            const int r = min(cur.m0 / p.n + wave % 3, p.nr - 1);
            __builtin_amdgcn_global_load_lds((gs_gbl_ptr)((const char*)p.a1_V + (long)r * p.a1_ld * 4 + lane * 16), (gs_lds_ptr)(smem + TP_TAB(wave % 3)), 16, 0, 0);
          }
          if constexpr (FORM == TP_MARG) {
            const int r = min(cur.r0 + 4 * (wave >> cur.variant) + (li >> 3), p.nr - 1), c = min(8 * cur.cb + (li & 7), p.n - 1);
            __builtin_amdgcn_global_load_lds((gs_gbl_ptr)(p.gen_d + (long)r * p.n + c), (gs_lds_ptr)(smem + TP_OFF_G + wave * 256), 4, 0, 0);
          }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int g = (wave * 3 + q) * 64 + lane;
          const int lrow = g / 6, slot = g - lrow * 6;
          const int piece = (slot + 3 * ((lrow >> 3) & 1)) % 6;
          __builtin_amdgcn_global_load_lds((gs_gbl_ptr)(Bb + (long)lrow * p.ldb_bytes + st * 96 + piece * 16), (gs_lds_ptr)(smem + st * TP_SLOT + (wave * 3 + q) * 1024), 16, 0, 0);
        }
      }
    }
    tp_wait_vm<7>();      // the first group (and every plain load before it) has landed
    tp_barrier();
    int rsel = 0;
    float gen_d = 0.f;
    bf16x8 ah, am, al;
    tp_own<FORM>(p, smem, cur, wave, lane, M, 0, rsel, gen_d);
    tp_prep<FORM>(smem, wave, lane & 31, lane >> 5, 0, 0, 0, rsel, gen_d, 0, ah, am, al);
    while (true) {
      tp_tile nxt = tp_next<FORM>(p, it);
      const bool more = nxt.variant >= 0;
      if (!more) nxt = cur;
      tp_aoff<FORM, RIDX>(p, nxt, wave, lane, M, aoff_n, ridx_n);
      if (cur.variant == 0) tall_tile<FORM, 8, 1, RIDX, TRI, EARLY>(p, smem, S, cur, nxt, aoff, aoff_n, ridx, M, wgstat, rsel, gen_d, ah, am, al);
      else if (cur.variant == 1) tall_tile<FORM, 4, 2, RIDX, TRI, EARLY>(p, smem, S, cur, nxt, aoff, aoff_n, ridx, M, wgstat, rsel, gen_d, ah, am, al);
      else tall_tile<FORM, 2, 4, RIDX, TRI, EARLY>(p, smem, S, cur, nxt, aoff, aoff_n, ridx, M, wgstat, rsel, gen_d, ah, am, al);
      if (!more) break;
      cur = nxt;
      aoff[0] = aoff_n[0]; aoff[1] = aoff_n[1]; aoff[2] = aoff_n[2]; aoff[3] = aoff_n[3];
      ridx[0] = ridx_n[0]; ridx[1] = ridx_n[1]; ridx[2] = ridx_n[2]; ridx[3] = ridx_n[3];
    }
  }

template <int FORM, bool RIDX, int TRI, int SPLIT>      // SPLIT: 0 every wave [MFMA, prepare], 1 the two halves in opposite order, 2 every wave interleaved
__global__ __launch_bounds__(512, 2) void gs_tall_kernel(gs_tall_args p) {
  __shared__ __attribute__((aligned(16))) char smem[TP_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (p.span && tid == 0) atomicMin(p.span, (unsigned long long)__builtin_amdgcn_s_memrealtime());
  int M = p.M;
  if constexpr (RIDX) M = min(*p.m_dev, p.M);
  // this workgroup's share of the units
  long units;
  if constexpr (FORM == TP_MARG) units = (long)p.ncb * p.H;
  else units = (M + 63) / 64;
  tp_iter it;
  it.u = (int)(units * blockIdx.x / gridDim.x);
  it.u_hi = (int)(units * (blockIdx.x + 1) / gridDim.x);
  it.pre = (p.phases & 1) ? (int)((blockIdx.x >> 3) & (FORM == TP_MARG ? 1 : 3)) : 0;
  // measurement: static issue priority for one half of the workgroup (the two waves of a SIMD arbitrate VALU issue by priority, then age)
  if ((p.phases & 2) && wave >= 4) __builtin_amdgcn_s_setprio(1);
  if ((p.phases & 4) && wave < 4) __builtin_amdgcn_s_setprio(1);

  // table rows of 256 floats.  MARG: rows 0..6 = the [K][8] constants rows of the dE2 transform as a structure of arrays (see
  // gs_panel.hip); FGEN: rows 0..5 = the tiles' Vs rows (two sets); row 11 = the bias (PLAIN, FGEN)
  {
    float* tw = reinterpret_cast<float*>(smem + TP_OFF_T);
    if constexpr (FORM == TP_MARG) {
      for (int i = tid; i < 2 * 256; i += 512) {
        const int k = i >> 1, hf = i & 1;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (k < p.K) t = reinterpret_cast<const f32x4*>(p.gen_c)[i];
        if (hf == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) tw[q * 256 + k] = t[q];
        } else {
#pragma unroll
          for (int q = 0; q < 3; ++q) tw[(4 + q) * 256 + k] = t[q];
        }
      }
    } else {
      if (tid < 256) tw[11 * 256 + tid] = p.bias ? p.bias[tid] : 0.f;
    }
  }
  float wgstat = 0.f;
  if (it.u < it.u_hi) {
    // waves w and w + 4 share a SIMD: the two halves of the workgroup run the two parts of every step in opposite order (two copies
    // of the whole share loop: a branch inside the step would make the compiler merge the 128 accumulators of both orders)
    if constexpr (SPLIT == 2) tall_share<FORM, RIDX, TRI, 2>(p, smem, it, M, wave, lane, wgstat);
    else if (SPLIT == 1 && wave < 4) tall_share<FORM, RIDX, TRI, 1>(p, smem, it, M, wave, lane, wgstat);
    else tall_share<FORM, RIDX, TRI, 0>(p, smem, it, M, wave, lane, wgstat);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing may still be writing this workgroup's LDS
  if constexpr (TRI == 2) {
    __syncthreads();
    if (p.trace && blockIdx.x < 16)
      for (int i = tid; i < 2816; i += 512) p.trace[blockIdx.x * 2816 + i] = *reinterpret_cast<const unsigned*>(smem + TP_OFF_T + i * 4);
  }
  if constexpr (FORM == TP_FGEN) {
    if (p.colstat) p.colstat[(long)blockIdx.x * 512 + tid] = wgstat;      // [grid][q = tid >> 8][col = tid & 255]
  }
  if (p.span && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    atomicMax(p.span + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
// which products take the 256-row kernels, a bit mask: 1 plain products (H1 = relu(X W1 + b1)), 2 PGE forward beside the real-side
// stream (training pass), 4 PGE forward alone on the chip (inference pass), 8 fused marginals.  0: gs_panel.hip's 128-row kernels
// everywhere (measurement / bitwise A-B knob)
#define TP_SITES_DEFAULT 13      /* (the training pass runs beside the real-side stream: there the persistent grid loses, profiles/r04) */
static int g_tall = TP_SITES_DEFAULT;
static int g_tall_grid = 0;        // workgroups per launch; 0 = one per compute unit
static int g_tall_krep = 1, g_tall_triage = 0, g_tall_phases = 1;
static unsigned* g_tall_trace = nullptr;
static int g_tall_split = 1;      // 1: waves 0-3 [prepare, MFMA], waves 4-7 [MFMA, prepare]; 0: every wave [MFMA, prepare]

static int g_tall_hoist_share = 13;     // compute units (in sixteenths) the hoisted pass's persistent grid takes (gs_tall_forward_a1, alone = 2)
static int g_tall_plain_share = 14;     // compute units (in sixteenths) of the persistent grid of H1 = relu(X W1 + b1) (gs_tall_plain): it is HBM-bound (the same 75 us on 12/16 of the chip) and a full grid holds every register file, so the main stream's first launches of the iteration waited for it (profiles/r06/stage_times_plain_share.txt: fused iteration 1 471 -> 1 448 us at 14/16)
extern "C" int gs_panel_set_plain_share(int sixteenths) {
  g_tall_plain_share = sixteenths < 1 ? 1 : (sixteenths > 16 ? 16 : sixteenths);
  return GS_OK;
}
extern "C" int gs_panel_set_hoist_share(int sixteenths) {
  g_tall_hoist_share = sixteenths < 1 ? 1 : (sixteenths > 16 ? 16 : sixteenths);
  return GS_OK;
}
extern "C" int gs_gemm_set_panel_tall(int on) {
  g_tall = on == 1 ? TP_SITES_DEFAULT : (on & 15);      // 0 off, 1 default sites, else an explicit mask of sites (16 + mask allows mask 1)
  if (on >= 16) g_tall = on & 15;
  return GS_OK;
}
// measurement hooks (tools/ab_panel.py): persistent grid size (0 = every compute unit), K loop repeated krep times
extern "C" int gs_panel_tall_debug(int grid, int krep, int triage) {
  g_tall_grid = grid < 0 ? 0 : grid;
  g_tall_krep = krep < 1 ? 1 : krep;
  g_tall_triage = triage & 63;
  g_tall_phases = ((triage & 64) ? 0 : 1) | ((triage & 256) ? 2 : 0) | ((triage & 512) ? 4 : 0);      // bit 6: every share starts with a full tile (tile
                                    // boundaries coincide chip-wide); bit 8 / 9: static s_setprio 1 for waves 4-7 / waves 0-3
  g_tall_split = (triage & 1024) ? 2 : ((triage & 128) ? 0 : 1);      // bit 7: every wave runs a step in the order [MFMA, prepare]; bit 10: every wave
                                    // interleaves the next sub-step's operand preparation with its MFMAs (PLAIN / FGEN product instantiations only)
  return GS_OK;
}

// the next plain-product launches record time stamps of their first 16 workgroups into buf ([16][2816] words); null switches it off
extern "C" int gs_panel_tall_trace(void* buf) {
  g_tall_trace = (unsigned*)buf;
  return GS_OK;
}

static int tall_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, c = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
    cus = c > 0 ? c : 256;
  }
  return cus;
}
// one persistent workgroup per compute unit, fewer when the problem has less than two full tiles per workgroup
static int tall_grid(long units, int units_per_full_tile) {
  const int cap = g_tall_grid > 0 ? g_tall_grid : tall_cus();
  (void)units_per_full_tile;
  long g = units;                                  // small problems: more workgroups with shorter tiles
  if (g < 1) g = 1;
  return (int)(g < cap ? g : cap);
}

// C[M x 256] = relu?(A W^T + bias), W given as its three-way split image; optional row list (see gs_panel_rows).  Returns
// GS_ERR_UNSUPPORTED when the shape is not covered (the caller then launches gs_panel.hip's kernel).
int gs_tall_plain(const float* A, long lda, const void* Bimg, long ldb_bytes, const float* bias, int relu, float* C, long ldc, long c_rows,
                  int M, int K, const int* rowidx, const int* m_dev, long a_rows, hipStream_t s) {
  if (!(g_tall & 1) || M < 1 || K < 64 || (K & 31) || (lda & 3) || (ldc & 3) || ldc < 256) return GS_ERR_UNSUPPORTED;
  if ((double)a_rows * lda * 4.0 >= 4294967296.0 || (double)c_rows * ldc * 4.0 >= 2147483648.0) return GS_ERR_UNSUPPORTED;
  if ((((uintptr_t)A | (uintptr_t)C | (uintptr_t)Bimg) & 15) || (rowidx && !m_dev)) return GS_ERR_UNSUPPORTED;
  gs_tall_args a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.B = Bimg; a.ldb_bytes = ldb_bytes; a.C = C; a.ldc = ldc; a.c_rows = c_rows; a.M = M; a.K = K;
  a.bias = bias; a.relu = relu; a.rowidx = rowidx; a.m_dev = m_dev; a.krep = g_tall_krep; a.triage = g_tall_triage; a.phases = g_tall_phases;
  dim3 grid(tall_grid(gs_cdiv(M, 64), 4));
  const dim3 block(512);
  if (g_tall_plain_share < 16 && g_tall_grid == 0) { const int cap = tall_cus() * g_tall_plain_share / 16; if ((int)grid.x > cap) grid.x = cap > 0 ? cap : 1; }
#define TP_LAUNCH(FORM_, RIDX_, TRI_)                                                                               \
  do {                                                                                                             \
    if (g_tall_split) hipLaunchKernelGGL((gs_tall_kernel<FORM_, RIDX_, TRI_, 1>), grid, block, 0, s, a);           \
    else hipLaunchKernelGGL((gs_tall_kernel<FORM_, RIDX_, TRI_, 0>), grid, block, 0, s, a);                        \
  } while (0)
  const bool tri = g_tall_triage || g_tall_krep != 1;
  if (rowidx) { if (tri) TP_LAUNCH(TP_PLAIN, true, 1); else TP_LAUNCH(TP_PLAIN, true, 0); }
  else if (g_tall_trace) { a.trace = g_tall_trace; TP_LAUNCH(TP_PLAIN, false, 2); }
  else if (tri) TP_LAUNCH(TP_PLAIN, false, 1);
  else if (g_tall_split == 2) hipLaunchKernelGGL((gs_tall_kernel<TP_PLAIN, false, 0, 2>), grid, block, 0, s, a);      // (measurement: interleaved form)
  else TP_LAUNCH(TP_PLAIN, false, 0);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// E2 [nr * n][256] = A1(U, V) W^T + bias, column sums / sums of squares per WORKGROUP in colstat [*stat_slabs][2][256]
int gs_tall_forward_a1(const float* Us, const float* Vs, long ld, int n, int nr, const void* Bimg, long ldb_bytes,
                       const float* bias, float* C, long ldc, int K, float* colstat, int* stat_slabs, unsigned long long* span,
                       int alone, hipStream_t s) {
  if (!(g_tall & (alone ? 4 : 2)) || K != 256 || n < 128 || nr < 1 || (ld & 3) || (ldc & 3) || ldc < 256) return GS_ERR_UNSUPPORTED;
  if ((double)nr * n * ldc * 4.0 >= 2147483648.0 || (double)n * ld * 4.0 >= 4294967296.0) return GS_ERR_UNSUPPORTED;
  if (!Us || !Vs || (((uintptr_t)Us | (uintptr_t)Vs | (uintptr_t)C | (uintptr_t)Bimg) & 15)) return GS_ERR_UNSUPPORTED;
  gs_tall_args a;
  memset(&a, 0, sizeof(a));
  a.B = Bimg; a.ldb_bytes = ldb_bytes; a.C = C; a.ldc = ldc; a.c_rows = (long)nr * n; a.M = nr * n; a.K = K;
  a.bias = bias; a.colstat = colstat; a.a1_U = Us; a.a1_V = Vs; a.a1_ld = ld; a.n = n; a.nr = nr;
  a.span = span; a.krep = g_tall_krep; a.triage = g_tall_triage; a.phases = g_tall_phases;
  dim3 grid(tall_grid(gs_cdiv(a.M, 64), 4)), block(512);
  // alone == 2: the pass runs beside a chain of latency-bound launches (the inner SGC steps while the next iteration's training forward
  // is hoisted, gs_engine.hip): the persistent grid leaves three compute units in sixteen to them (sweep: profiles/r05/stage_times_hoist_share.txt) — a full grid made every one of their
  // ~20 small launches wait for a workgroup of this kernel to retire (inner steps 140 -> 323 us, profiles/r05/timeline_hoisted_full_grid.txt)
  if (alone == 2 && g_tall_grid == 0) { const int cap = tall_cus() * g_tall_hoist_share / 16; if ((int)grid.x > cap) grid.x = cap > 0 ? cap : 1; }
  if (g_tall_split == 2) hipLaunchKernelGGL((gs_tall_kernel<TP_FGEN, false, 0, 2>), grid, block, 0, s, a);      // (measurement: interleaved form)
  else if (g_tall_triage || g_tall_krep != 1) TP_LAUNCH(TP_FGEN, false, 1);      // (measurement: phases switched off, results are garbage)
  else TP_LAUNCH(TP_FGEN, false, 0);
  GS_LAUNCH_CHECK();
  if (stat_slabs) *stat_slabs = (int)grid.x;
  return GS_OK;
}

// fused marginals with dE2 generated from E2 and the mask regenerated from U, V: partV [ceil(n / 8)][nr][256] (sum over the 8 c of a
// block), partU [ceil(nr / 16)][n][256] (sum over the 16 r of a slab) — the layouts of gs_panel_marginals
int gs_tall_marginals(const float* E2, const void* Bimg, long ldb_bytes, const float* gen_table, const float* de3, const float* Us,
                      const float* Vs, long ld, int n, int nr, int K, float* partU, float* partV, hipStream_t s) {
  // (n >= 128: the shapes whose forward takes gs_tall_forward_a1 — forward, mask and dW1 then all use the folded form of A1)
  if (!(g_tall & 8) || K != 256 || n < 128 || nr < 1 || (ld & 3) || !Us || !Vs) return GS_ERR_UNSUPPORTED;
  const int H = gs_cdiv(nr, 16), ncb = gs_cdiv(n, 8);
  if ((double)nr * n * K * 4.0 >= 4294967296.0 || (double)ncb * nr * 1024.0 >= 2147483648.0 || (double)H * n * 1024.0 >= 2147483648.0)
    return GS_ERR_UNSUPPORTED;
  if ((((uintptr_t)E2 | (uintptr_t)Bimg | (uintptr_t)gen_table | (uintptr_t)partU | (uintptr_t)partV) & 15))
    return GS_ERR_UNSUPPORTED;
  gs_tall_args a;
  memset(&a, 0, sizeof(a));
  a.A = E2; a.lda = K; a.B = Bimg; a.ldb_bytes = ldb_bytes; a.M = nr * n; a.K = K;
  a.a1_U = Us; a.a1_V = Vs; a.a1_ld = ld; a.n = n; a.nr = nr;
  a.gen_c = gen_table; a.gen_d = de3; a.partU = partU; a.partV = partV; a.H = H; a.ncb = ncb; a.krep = g_tall_krep; a.triage = g_tall_triage; a.phases = g_tall_phases;
  const dim3 grid(tall_grid((long)ncb * H, 2)), block(512);
  if (g_tall_triage || g_tall_krep != 1) hipLaunchKernelGGL((gs_tall_kernel<TP_MARG, false, 1, 0>), grid, block, 0, s, a);      // (measurement)
  else hipLaunchKernelGGL((gs_tall_kernel<TP_MARG, false, 0, 0>), grid, block, 0, s, a);      // (the [prepare, MFMA] order does not fit its register budget)
  GS_LAUNCH_CHECK();
  return GS_OK;
}
