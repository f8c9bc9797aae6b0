/* Measurement knobs and debug hooks of libgslim_hip.so — NOT part of the product ABI (include/gslim.h).  They are process-global,
   not thread-safe, and exist for the A/B tools under tools/ and the bitwise form-against-form tests; a product build may drop them. */
#ifndef GSLIM_INTERNAL_H
#define GSLIM_INTERNAL_H
#ifdef __cplusplus
extern "C" {
#endif

/* exact-split products at N = 256 on the persistent 256-row kernel of csrc/gs_panel2.hip (default 1); 0 = the 128-row kernels of
   csrc/gs_panel.hip everywhere (bitwise the same products; BN-2 partial sums grouped per tile instead of per workgroup) */
int gs_gemm_set_panel_tall(int on);
/* csrc/gs_panel2.hip: workgroups per launch (0 = one per compute unit), K loop of every tile repeated krep times, triage bits
   (results are garbage): 1 no epilogue, 2 no A staging, 4 no B staging, 8 no waits / barriers, 16 no MFMAs, 32 no operand split */
int gs_panel_tall_debug(int grid, int krep, int triage);

/* csrc/gs_panel2.hip: the next plain-product launches record s_memtime stamps (3 per 16-k step of waves 0 and 4) of their first 16
   workgroups into buf ([16][2816] 32-bit words, device memory); null switches it off (tools/trace_tall.py) */
int gs_panel_tall_trace(void* buf);

/* real side (T = 2): 1 = the hop-2 gather kernel also writes the sign bits of every gathered H1 row per edge slot and the
   edge product reads its ReLU mask from those 32 B (bitwise the same gradients); 0 (default: measured 20 us slower on, profiles/r04/edge_bits_ab.txt) = the edge product gathers the 1-KiB mask rows */
int gs_gcond_set_edge_bits(int on);

#ifdef __cplusplus
}
#endif
#endif
