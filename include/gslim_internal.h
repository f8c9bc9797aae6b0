/* Measurement knobs and debug hooks of libgslim_hip.so — NOT part of the product ABI (include/gslim.h).  They are process-global,
   not thread-safe, and exist for the A/B tools under tools/ and the bitwise form-against-form tests; a product build may drop them. */
#ifndef GSLIM_INTERNAL_H
#define GSLIM_INTERNAL_H
#include "gslim.h"
#ifdef __cplusplus
extern "C" {
#endif

/* exact-split products at N = 256 on the persistent 256-row kernel of csrc/gs_panel2.hip (default 1); 0 = the 128-row kernels of
   csrc/gs_panel.hip everywhere (bitwise the same products; BN-2 partial sums grouped per tile instead of per workgroup) */
int gs_gemm_set_panel_tall(int on);
/* Measurement knob: the share of the compute units (in sixteenths, default 13) the persistent grid of the HOISTED training forward
 * takes (gs_gcond_set_hoist_forward): the rest stays free for the inner steps' small launches running beside it. */
int gs_panel_set_hoist_share(int sixteenths);
/* csrc/gs_panel2.hip: workgroups per launch (0 = one per compute unit), K loop of every tile repeated krep times, triage bits
   (results are garbage): 1 no epilogue, 2 no A staging, 4 no B staging, 8 no waits / barriers, 16 no MFMAs, 32 no operand split;
   scheduling bits (results unchanged): 64 every share starts with a full tile, 128 every wave runs a step as [MFMA, prepare], 256 / 512
   static s_setprio 1 for waves 4-7 / 0-3, 1024 every wave interleaves the next sub-step's operand preparation with its MFMAs */
int gs_panel_tall_debug(int grid, int krep, int triage);

/* csrc/gs_panel2.hip: the next plain-product launches record s_memtime stamps (3 per 16-k step of waves 0 and 4) of their first 16
   workgroups into buf ([16][2816] 32-bit words, device memory); null switches it off (tools/trace_tall.py) */
int gs_panel_tall_trace(void* buf);

/* real side (T = 2): 1 = the hop-2 gather kernel also writes the sign bits of every gathered H1 row per edge slot and the
   edge product reads its ReLU mask from those 32 B (bitwise the same gradients); 0 (default: measured 20 us slower on, profiles/r04/edge_bits_ab.txt) = the edge product gathers the 1-KiB mask rows */
int gs_gcond_set_edge_bits(int on);


/* ---- moved here from gslim.h in round 4: process-global measurement knobs and debug hooks ---- */
/* K-tile depth (16 or 32) of the small tiles and of the 128x128 tile (0 = by operand mode); a measurement knob (defaults 32 / 0). */
int gs_gemm_set_tuning(int bk_small, int bk_big);

/* Routing of latency-bound products to the one-tile-per-workgroup kernel (32x32 tile, K split over the 4 waves, fragments
   fetched straight from global memory): taken when the 32x32 tile count is <= max_tiles and the product is <= max_gflop
   (defaults 8192 / 1.5); max_tiles = 0 switches it off.  A measurement knob. */
int gs_gemm_set_small(long max_tiles, double max_gflop);

/* route eligible products to the row-panel kernel (default 1); 0 sends them to the 128x128 tile (A/B measurements) */
int gs_gemm_set_panel(int enable);

/* route the real-side edge product (generated B, gathered K, N == 256) to its dedicated kernel (default 1: exact split done once at
   LDS-write time, hardware-transposed fragment reads); 2 = the same kernel with the split at fragment-read time (round 2's form,
   bitwise the same result); 0 = 128x128 tile */
int gs_gemm_set_edge_kernel(int enable);

/* the row-panel kernel finishes the last partial round of a launch in 32-row tiles (default 1); 0 = 128-row tiles throughout */
int gs_gemm_set_panel_narrow_tail(int on);

/* PGE backward marginals (engine-internal product dY1 = (dE2 W1) * [A1 > 0], needed only through its row / column sums):
   0 = materialise dY1 and reduce it, 1 = sums fused into the row-panel product's epilogue on a materialised dE2,
   2 (default) = as 1 with dE2 formed from E2 in the operand path where it applies (feature epochs).  A measurement knob. */
int gs_gemm_set_panel_marg(int level);

/* PGE layer-1 activations A1 = relu(BN1(U[c] + V[r] + b0)): 1 (default) = formed from U, V inside the layer-2 product's
   operand path and inside the marginals' mask (A1 is written only when dW1 needs it); 0 = always materialised first.
   Both forms give bitwise the same results (tests/test_gpu_panel.py).  A measurement knob. */
int gs_gemm_set_panel_a1gen(int on);

/* dynamic LDS (bytes) added to the training pass's 128-row forward launch (it runs beside the real-side stream): 80 KB static + a pad
   above 80 KB in total leaves one workgroup per CU instead of two.  0 (default) = none.  Same values either way. */
int gs_panel_set_fwd_lds_pad(int bytes);

/* row-panel kernel measurement hooks (tools/trace_panel.py): the K loop repeated krep times; triage bit 0 = no stores,
   bit 1 = per-workgroup placement / timing trace into desc.scratch.  krep = 1, triage = 0 restores the product path. */
int gs_panel_debug(int krep, int triage);

/* 1 (default): unsharded engines fold the last stage of the BatchNorm-2 / layer-3 statistics reductions into the consuming
   launch (two launches fewer per PGE pass); 0 = separate launches.  Bitwise the same results.  A measurement knob. */
int gs_gcond_set_folded_reductions(int on);

/* Measurement / test knob: 1 = the product H1 = relu(X W1 + b1) also writes its sign bits (gs_gemm_desc.signbits) and the
 * real-side edge product reads its ReLU mask from them (gs_gemm_desc.bgen_bits: 32 B instead of 1 KiB per sampled edge); 0 (default:
 * measured faster, the mask rows are L2-resident) = the mask comes from the fp32 rows.  Same predicate: bitwise the same gradients. */
int gs_gcond_set_h1_bits(int on);

/* Measurement / test knob: 1 (default) = the latency-bound chains run as fused launches (csrc/gs_chain.hip: the inner SGC steps of
 * gcond.py:63-72 with logits + softmax + NLL gradient in one launch, the two n' x n' x h products re-associated into n' x n' x C
 * ones, Adam in the epilogue of the gradient products; the synthetic match's logits + softmax in one launch); 0 = one launch per
 * product (first form).  Same values up to fp32 re-association. */
int gs_gcond_set_fused_chains(int on);
/* Measurement knob: the inner SGC steps of a two-transform SGC model (gcond.py:63-72) as ONE persistent launch of ~48 workgroups with
 * grid barriers between the four phases of a step (csrc/gs_inner.hip): 1 (default) = whenever no hoisted training forward runs beside
 * the stage (forward reuse — what GCond.reduce runs —, staged drivers, sharded ranks), 2 = always, 0 = never (the launch chain of
 * round 5: seven dependent launches per step).  Same values up to fp32 summation order.  Per engine: GS_KNOB_INNER_PERSIST.
 * gs_inner_persist_set_grid: workgroups of that launch (default 48 = what the hoisted forward's 13/16 grid leaves free). */
int gs_gcond_set_inner_persist(int on);
/* Measurement knob (default 0): 1 = the real-side edge product gW1_c = sum_e x_src(e) (x) (w_e [H1[src(e)] > 0] Y[row e]) runs over
 * ENTRIES = (distinct source of the outermost block, up to two of its in-edges) instead of edge slots: x and the mask row of a source are
 * fetched once per entry, its in-edges are summed in the operand path (gs_dedup_last_kernel builds the distinct sources, the block's CSC
 * over them and the entries behind every sampling / gs_gcond_set_blocks).  The same sum in another association.  SGC, two transforms,
 * two hops, 256 hidden units, 'ours'.  Measured at ogbn-arxiv (profiles/r06/edge_src_ab.txt): K 8 700 -> ~6 100 per class, the real
 * side 452 -> 429 us alone, but the de-duplication kernel costs more than that beside the rest of the iteration: not the default.
 * Per engine: GS_KNOB_EDGE_SRC. */
int gs_gcond_set_edge_src(int on);
/* the de-duplication behind every sampling: 1 (default) = a chain of chip-wide launches, 0 = one workgroup per class (round 6's first form:
 * measurement / test hook); bitwise the same structure */
int gs_dedup_set_form(int form);
/* compute units (in sixteenths) the persistent grid of H1 = relu(X W1 + b1) takes (16 = all: default; measurement knob — the rows are tiled
 * the same way whatever the grid: bitwise the same H1) */
int gs_panel_set_plain_share(int sixteenths);
int gs_inner_persist_set_grid(int workgroups);
int gs_inner_persist_trace(void* device_u64_G_by_128);      /* measurement: s_memrealtime stamps around every grid barrier (NULL = off) */

/* Measurement knob: 1 (default) = a sharded rank computes only the rows of H1 = relu(X W1 + b1) (models/sgc.py:38-43) that the
 * sampled blocks of ITS classes gather (marked and compacted on the device); 0 = every rank computes all N rows.  Same values. */
int gs_gcond_set_h1_rows(int on);

/* Measurement / test knob: 1 (default) = PGE epochs compute dW1 | db1 of the structure generator's second layer from generated
 * operands (dE2 from E2, the layer-1 activations from U, V: nothing materialised); 0 = round 2's form (both operands written to
 * memory for a general tiled product).  Same values up to fp32 rounding of the folded BatchNorm-1 affine. */
int gs_gcond_set_dw1_generated(int on);

/* Measurement / test knob: 1 (default) = the real-side edge product (csrc/gs_edge.hip) runs on BALANCED K slabs: the chip's 512 resident
 * workgroup slots are cut over the classes in proportion to their live edge counts (a device-side plan, one launch in place of the
 * kcount launch), so every workgroup runs about the same number of 16-edge steps; 0 = a fixed slab length for every class (rounds
 * 2-4: 22 % of the slots stayed empty at ogbn-arxiv).  Same sums in a different (fixed) slab partition: equal up to fp32 re-association. */
int gs_gcond_set_edge_plan(int on);

/* Measurement knob: 1 (default) = the workgroups of the real-side ELL gathers (hop 2, hop 1) are mapped so that every row of a class
 * is processed on ONE XCD (workgroup id % 8 = class % 8): a class's repeated source rows and the next hop's sources meet in one L2;
 * 0 = rows in memory order over all XCDs.  Same values (every row is computed by the same code either way). */
int gs_gcond_set_ell_xcd(int on);

/* Measurement knob: 0 (default) = the real side's gathers whose output is only read through the NEXT gather's live indices (outermost
 * and middle hops) leave their padding rows unwritten (a third of the capacity rows); 1 = they write zeros there (rounds 1-4).  Same
 * values: nothing reads those rows.  (The transposed gathers of the backward skip their padding rows unconditionally.) */
int gs_gcond_set_pad_zero(int on);

/* Process-wide measurement override of gs_gcond_set_hoist_forward (gslim.h; per engine, library default off): -1 (default) = every
 * engine follows its own setting, 0 / 1 / 2 = all engines of the process (2: hoisted with the edge product's 7/8 slab plan). */
int gs_gcond_set_hoist_forward_all(int on);

/* Priority of the second (real-side) stream of engines created AFTER the call: -1 = highest (default), 0 = normal, 1 = lowest. */
int gs_gcond_set_aux_priority(int level);

/* Test hook of the sampler's seed selection (gs_sample_blocks): members of the crossing key bin that are ranked directly
 * (1 .. 2048, default 2048); a small value forces the second histogram level on ordinary pools.  Same seeds either way. */
int gs_sample_set_cand_cap(int cap);

/* Engine-level knobs per ENGINE (the int-only setters above change the process-wide default an engine follows while its own value is
   -1): knob ids below, value 0 / 1, -1 = follow the default again.  gs_gcond_get_knob returns the value in effect (-1: bad arguments). */
enum { GS_KNOB_FUSED_CHAINS = 0, GS_KNOB_H1_ROWS = 1, GS_KNOB_H1_BITS = 2, GS_KNOB_EDGE_BITS = 3, GS_KNOB_DW1_GENERATED = 4,
       GS_KNOB_FOLDED_REDUCTIONS = 5, GS_KNOB_CHAINED_PRODUCTS = 6, GS_KNOB_FORK_LATE = 7, GS_KNOB_EDGE_PLAN = 8, GS_KNOB_INNER_PERSIST = 9, GS_KNOB_EDGE_SRC = 10, GS_KNOB_COUNT = 11 };
int gs_gcond_set_knob(gs_gcond* e, int knob, int value);
int gs_gcond_get_knob(const gs_gcond* e, int knob);

#ifdef __cplusplus
}
#endif
#endif
