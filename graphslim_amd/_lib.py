"""ctypes binding of libgslim_hip.so (C ABI declared in include/gslim.h).

The product path fails loudly when the HIP library is missing: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgslim_hip.so")

GS_OK = 0

vp = C.c_void_p


class GemmDesc(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("sAm", C.c_long), ("sAk", C.c_long), ("sBk", C.c_long), ("sBn", C.c_long), ("ldc", C.c_long),
                ("nb1", C.c_int), ("nb2", C.c_int),
                ("sAb1", C.c_long), ("sAb2", C.c_long), ("sBb1", C.c_long), ("sBb2", C.c_long), ("sCb1", C.c_long),
                ("sCb2", C.c_long),
                ("offA", vp), ("offB", vp), ("offC", vp), ("Mb", vp), ("Kb", vp),
                ("gatherA_m", vp), ("gatherA_k", vp), ("gatherB_k", vp), ("sGb1", C.c_long),
                ("kchunk", C.c_int), ("alpha", C.c_float), ("beta", C.c_float),
                ("bias", vp), ("sbias_b1", C.c_long), ("relu", C.c_int), ("mask", vp), ("ldmask", C.c_long),
                ("colstat", vp), ("bgen_mask", vp), ("bgen_ldm", C.c_long), ("bgen_idx", vp), ("bgen_w", vp), ("bgen_div", C.c_int),
                ("prec", C.c_int), ("scratch", vp),
                ("signbits", vp), ("ld_signbits", C.c_long), ("bgen_bits", vp), ("bgen_ldbits", C.c_long), ("bgen_bits_edge", C.c_int)]


class Blocks(C.Structure):
    _fields_ = [("C", C.c_int), ("bcap", C.c_int), ("f1", C.c_int), ("f2", C.c_int), ("cap1", C.c_int),
                ("n_seed", vp), ("n1", vp), ("nid1", vp), ("col1", vp), ("val1", vp), ("tptr1", vp), ("tedge1", vp),
                ("col0", vp), ("val0", vp), ("nmid", C.c_int), ("fm", C.c_int * 3), ("capm", C.c_int * 3), ("nm", vp * 3), ("nidm", vp * 3),
                ("colm", vp * 3), ("valm", vp * 3), ("tptrm", vp * 3), ("tedgem", vp * 3), ("scrm", vp * 3)]


class GCondConfig(C.Structure):
    _fields_ = [("n_nodes", C.c_int), ("d", C.c_int), ("ldx", C.c_int), ("nclass", C.c_int), ("n_syn", C.c_int),
                ("ntrans", C.c_int), ("hidden", C.c_int), ("pge_hidden", C.c_int), ("bcap", C.c_int), ("f1", C.c_int),
                ("f2", C.c_int), ("dis_metric", C.c_int), ("inner_loop", C.c_int),
                ("lr_feat", C.c_float), ("lr_adj", C.c_float), ("lr_model", C.c_float), ("seed", C.c_uint64),
                ("external_blocks", C.c_int), ("condense_model", C.c_int), ("pge_chunks", C.c_int), ("with_bn", C.c_int),
                ("nlayers", C.c_int), ("fmid", C.c_int * 3), ("dropout", C.c_float)]


class GCondIO(C.Structure):
    _fields_ = [("X", vp), ("rowptr", vp), ("col", vp), ("val", vp), ("cls_ptr", vp), ("cls_nodes", vp),
                ("labels_syn", vp), ("syn_lo", vp), ("syn_hi", vp), ("feat_syn", vp), ("pge_params", vp),
                ("model_params", vp), ("adj_syn", vp), ("workspace", vp), ("workspace_bytes", C.c_size_t)]


# measurement knobs / debug hooks declared in include/gslim_internal.h (not product ABI)
INTERNAL_SYMBOLS = ["gs_panel_set_fwd_lds_pad", "gs_gemm_set_panel_tall", "gs_panel_set_hoist_share", "gs_panel_tall_debug", "gs_panel_tall_trace", "gs_gcond_set_edge_bits", "gs_gcond_set_edge_plan", "gs_gcond_set_ell_xcd", "gs_gcond_set_pad_zero", "gs_gcond_set_hoist_forward_all", "gs_gcond_set_knob",
                    "gs_gcond_get_knob", "gs_gemm_set_tuning", "gs_gemm_set_small", "gs_gemm_set_panel", "gs_gemm_set_edge_kernel", "gs_gemm_set_panel_narrow_tail", "gs_gemm_set_panel_marg", "gs_gemm_set_panel_a1gen", "gs_panel_debug", "gs_gcond_set_folded_reductions", "gs_gcond_set_h1_bits", "gs_gcond_set_fused_chains", "gs_gcond_set_inner_persist", "gs_gcond_set_edge_src", "gs_dedup_set_form", "gs_panel_set_plain_share", "gs_inner_persist_set_grid", "gs_inner_persist_trace", "gs_gcond_set_h1_rows", "gs_gcond_set_dw1_generated", "gs_gcond_set_aux_priority", "gs_sample_set_cand_cap"]

# engine-level knob ids of include/gslim_internal.h (gs_gcond_set_knob / gs_gcond_get_knob); tests/test_abi.py pins them to the enum
KNOBS = {"FUSED_CHAINS": 0, "H1_ROWS": 1, "H1_BITS": 2, "EDGE_BITS": 3, "DW1_GENERATED": 4, "FOLDED_REDUCTIONS": 5,
         "CHAINED_PRODUCTS": 6, "FORK_LATE": 7, "EDGE_PLAN": 8, "INNER_PERSIST": 9, "EDGE_SRC": 10}

_lib = None

# every symbol include/gslim.h declares (tests/test_abi.py checks the list against the header)
SYMBOLS = [
    "gs_version", "gs_last_error", "gs_gemm_f32", "gs_gemm_group_f32", "gs_gemm_tiles_m", "gs_gemm_route", "gs_reduce_slabs_f32",
    "gs_reduce_slabs_batched_f32", "gs_spmm_csr_f32", "gs_gather_rows_f32", "gs_sample_blocks", "gs_dense_gcn_norm_fwd", "gs_dense_gcn_norm_bwd",
    "gs_adam_step_f32", "gs_gcond_workspace_bytes", "gs_gcond_create", "gs_gcond_destroy", "gs_gcond_new_epoch", "gs_gcond_set_blocks",
    "gs_gcond_outer_step", "gs_gcond_outer_step_ex", "gs_gcond_set_identity_structure", "gs_gcond_read_loss", "gs_gcond_stage_sample",
    "gs_gcond_stage_real_grads", "gs_gcond_stage_real_async", "gs_gcond_stage_pge_forward", "gs_gcond_stage_syn_match",
    "gs_gcond_stage_pge_backward", "gs_gcond_stage_update", "gs_gcond_stage_inner", "gs_gcond_buffer", "gs_gcond_get_blocks", "gs_gcond_set_shard",
    "gs_gcond_set_overlap", "gs_gcond_set_operand_mode", "gs_gcond_set_full_neighbour", "gs_gcond_set_forward_reuse", "gs_gcond_set_dropout_iter",
    "gs_gcond_get_dropout_iter", "gs_gcond_probe",
    "gs_gcond_probe_read", "gs_gcond_probe_read_spans", "gs_gcond_probe_read_pass", "gs_gcond_probe_read_spmm", "gs_gcond_invalidate_forward", "gs_gcond_forward_cached", "gs_gcond_forward_hoisted", "gs_gcond_set_hoist_forward",
    "gs_gcond_adopt_cached_forward", "gs_gcond_stage_pge_forward_part", "gs_gcond_stage_pge_backward_part", "gs_gcond_stage_syn_finish",
    "gs_gcond_stage_inner_steps", "gs_bias_act_f32", "gs_log_softmax_f32", "gs_softmax_nll_grad_f32", "gs_colsum_f32", "gs_count_correct",
    "gs_edge_pair_sum_f32", "gs_bn_scratch_bytes", "gs_bn_relu_fwd_f32", "gs_bn_relu_bwd_f32", "gs_edge_score_f32", "gs_edge_scatter_sym_f32",
    "gs_edge_score_bwd_f32", "gs_segment_sum_rows_f32", "gs_colsum_tall_f32", "gs_mul_f32", "gs_mul_bwd_f32", "gs_mix_heads_f32",
    "gs_mix_heads_bwd_f32", "gs_sym_sigmoid_f32", "gs_sym_sigmoid_bwd_f32", "gs_bn_running_update_f32", "gs_bn_relu_eval_f32", "gs_ot_spectrum_f32",
    "gs_ot_value_f32", "gs_scale_cols_f32", "gs_ot_divdiff_f32", "gs_clamp_finite_f32", "gs_gcond_set_structure", "gs_gcond_set_block_structure",
    "gs_comm_unique_id", "gs_comm_probe", "gs_comm_create_rccl", "gs_comm_create_callback", "gs_comm_create_null", "gs_comm_ipc_arena", "gs_comm_create_ipc", "gs_comm_ipc_error", "gs_comm_check", "gs_comm_destroy", "gs_comm_rank", "gs_comm_world",
    "gs_comm_allreduce", "gs_comm_set_timing", "gs_comm_read_timing", "gs_comm_read_timing2", "gs_gcond_outer_step_sharded", "gs_gcond_outer_step_sharded_ex",
]

# gs_allreduce_fn of include/gslim.h: int fn(void* user, void* buf, long count, int dtype, gs_stream_t stream)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, vp, vp, C.c_long, C.c_int, vp)


def lib():
    """Load the HIP library once.  Raises ImportError (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C graphslim_amd/csrc`).  graphslim_amd has no CPU fallback.")
    import torch  # noqa: F401  (first: libgslim_hip.so must bind to the HIP runtime torch has already loaded)
    L = C.CDLL(LIB_PATH)
    L.gs_last_error.restype = C.c_char_p
    L.gs_gcond_destroy.restype = None
    L.gs_gcond_destroy.argtypes = [vp]
    i, l, f, u64, u32 = C.c_int, C.c_long, C.c_float, C.c_uint64, C.c_uint32
    L.gs_gemm_f32.argtypes = [C.POINTER(GemmDesc), vp]
    L.gs_gemm_group_f32.argtypes = [C.POINTER(GemmDesc), i, vp]
    L.gs_gemm_tiles_m.argtypes = [C.POINTER(GemmDesc)]
    L.gs_gemm_route.argtypes = [C.POINTER(GemmDesc)]
    L.gs_gemm_set_tuning.argtypes = [i, i]
    L.gs_gemm_set_small.argtypes = [l, C.c_double]
    L.gs_gemm_set_panel.argtypes = [i]
    L.gs_gemm_set_edge_kernel.argtypes = [i]
    L.gs_gemm_set_panel_narrow_tail.argtypes = [i]
    L.gs_gemm_set_panel_marg.argtypes = [i]
    L.gs_gemm_set_panel_a1gen.argtypes = [i]
    L.gs_panel_debug.argtypes = [i, i]
    L.gs_gemm_set_panel_tall.argtypes = [i]
    L.gs_panel_set_hoist_share.argtypes = [i]
    L.gs_panel_tall_debug.argtypes = [i, i, i]
    L.gs_panel_tall_trace.argtypes = [C.c_void_p]
    L.gs_gcond_set_edge_bits.argtypes = [i]
    L.gs_gcond_set_edge_plan.argtypes = [i]
    L.gs_gcond_set_ell_xcd.argtypes = [i]
    L.gs_gcond_set_pad_zero.argtypes = [i]
    L.gs_gcond_set_hoist_forward_all.argtypes = [i]
    L.gs_gcond_set_knob.argtypes = [vp, i, i]
    L.gs_gcond_get_knob.argtypes = [vp, i]
    L.gs_reduce_slabs_f32.argtypes = [vp, vp, l, i, l, f, vp]
    L.gs_reduce_slabs_batched_f32.argtypes = [vp, l, vp, l, i, l, i, f, vp]
    L.gs_spmm_csr_f32.argtypes = [vp, vp, vp, i, vp, l, i, vp, l, vp]
    L.gs_gather_rows_f32.argtypes = [vp, l, vp, i, i, vp, l, vp]
    L.gs_sample_blocks.argtypes = [vp, vp, vp, i, vp, vp, vp, C.POINTER(Blocks), u64, u32, vp]
    L.gs_dense_gcn_norm_fwd.argtypes = [vp, i, vp, vp, vp]
    L.gs_dense_gcn_norm_bwd.argtypes = [vp, vp, vp, i, vp, vp]
    L.gs_adam_step_f32.argtypes = [vp, vp, vp, vp, l, f, f, f, f, f, i, vp]
    L.gs_bias_act_f32.argtypes = [vp, l, vp, l, i, i, vp]
    L.gs_softmax_nll_grad_f32.argtypes = [vp, l, vp, l, i, vp, vp, vp]
    L.gs_colsum_f32.argtypes = [vp, l, l, i, vp, vp]
    L.gs_log_softmax_f32.argtypes = [vp, l, l, i, vp, l, vp]
    L.gs_count_correct.argtypes = [vp, l, i, vp, vp, l, vp, vp]
    L.gs_gcond_workspace_bytes.argtypes = [C.POINTER(GCondConfig), C.POINTER(C.c_size_t)]
    L.gs_gcond_create.argtypes = [C.POINTER(GCondConfig), C.POINTER(GCondIO), vp, C.POINTER(vp)]
    L.gs_gcond_new_epoch.argtypes = [vp, vp]
    L.gs_gcond_set_blocks.argtypes = [vp, C.POINTER(Blocks), vp]
    L.gs_gcond_outer_step.argtypes = [vp, i, vp]
    L.gs_gcond_outer_step_ex.argtypes = [vp, i, i, i, i, vp]
    L.gs_gcond_set_identity_structure.argtypes = [vp, i, vp]
    L.gs_gcond_read_loss.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long), vp]
    for name in ("gs_gcond_stage_sample", "gs_gcond_stage_real_grads", "gs_gcond_stage_real_async", "gs_gcond_stage_real_async", "gs_gcond_stage_syn_match", "gs_gcond_stage_inner"):
        getattr(L, name).argtypes = [vp, vp]
    L.gs_gcond_set_shard.argtypes = [vp, i, i, i, i]
    L.gs_gcond_set_overlap.argtypes = [vp, i]
    L.gs_gcond_set_operand_mode.argtypes = [vp, i]
    L.gs_gcond_set_full_neighbour.argtypes = [vp, vp, l, vp]
    L.gs_gcond_set_forward_reuse.argtypes = [vp, i]
    L.gs_gcond_forward_hoisted.argtypes = [vp]
    L.gs_gcond_set_hoist_forward.argtypes = [vp, i]
    L.gs_gcond_set_dropout_iter.argtypes = [vp, C.c_uint64]
    L.gs_gcond_get_dropout_iter.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.gs_gcond_probe.argtypes = [vp, i]
    L.gs_gcond_probe_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.gs_gcond_probe_read_spans.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.gs_gcond_probe_read_pass.argtypes = [vp, i, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.gs_gcond_probe_read_spmm.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.gs_gcond_invalidate_forward.argtypes = [vp]
    L.gs_gcond_forward_cached.argtypes = [vp]
    L.gs_gcond_adopt_cached_forward.argtypes = [vp, vp]
    L.gs_gcond_stage_pge_forward_part.argtypes = [vp, i, i, vp]
    L.gs_gcond_stage_pge_backward_part.argtypes = [vp, i, i, i, vp]
    L.gs_gcond_stage_syn_finish.argtypes = [vp, vp]
    L.gs_gcond_stage_inner_steps.argtypes = [vp, vp]
    L.gs_gcond_stage_pge_forward.argtypes = [vp, i, vp]
    L.gs_gcond_stage_pge_backward.argtypes = [vp, i, i, vp]
    L.gs_gcond_stage_update.argtypes = [vp, i, vp]
    L.gs_gcond_buffer.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_long)]
    L.gs_gcond_get_blocks.argtypes = [vp, C.POINTER(Blocks)]
    L.gs_gcond_set_structure.argtypes = [vp, vp, i, vp]
    L.gs_gcond_set_block_structure.argtypes = [vp, vp, i, i, vp]
    L.gs_edge_pair_sum_f32.argtypes = [vp, vp, l, vp, vp, vp, l, i, vp, l, vp]
    L.gs_bn_scratch_bytes.argtypes = [i]
    L.gs_bn_scratch_bytes.restype = C.c_size_t
    L.gs_bn_relu_fwd_f32.argtypes = [vp, l, l, i, vp, vp, f, vp, l, vp, vp, vp, vp]
    L.gs_bn_relu_bwd_f32.argtypes = [vp, l, vp, l, vp, l, l, i, vp, vp, vp, vp, l, vp, vp, vp, vp]
    L.gs_colsum_tall_f32.argtypes = [vp, l, l, i, vp, vp, vp]
    L.gs_mul_f32.argtypes = [vp, vp, l, vp, vp]
    L.gs_mul_bwd_f32.argtypes = [vp, vp, vp, l, vp, vp, i, vp]
    L.gs_mix_heads_f32.argtypes = [vp, vp, vp, f, vp, vp, vp, f, l, i, vp, vp]
    L.gs_mix_heads_bwd_f32.argtypes = [vp, vp, f, vp, f, l, i, vp, vp, vp, vp, vp]
    L.gs_sym_sigmoid_f32.argtypes = [vp, i, vp, vp]
    L.gs_sym_sigmoid_bwd_f32.argtypes = [vp, vp, i, vp, vp]
    L.gs_bn_running_update_f32.argtypes = [vp, vp, l, i, f, f, vp, vp, vp]
    L.gs_bn_relu_eval_f32.argtypes = [vp, l, l, i, vp, vp, f, vp, vp, vp, l, vp, vp, vp]
    L.gs_ot_spectrum_f32.argtypes = [vp, i, f, vp, vp]
    L.gs_ot_value_f32.argtypes = [vp, vp, i, f, vp, vp, vp, vp]
    L.gs_scale_cols_f32.argtypes = [vp, vp, i, i, vp, vp]
    L.gs_clamp_finite_f32.argtypes = [vp, l, f, vp]
    L.gs_ot_divdiff_f32.argtypes = [vp, vp, vp, vp, vp, vp, i, vp]
    L.gs_edge_score_f32.argtypes = [vp, l, vp, vp, l, i, vp, vp]
    L.gs_edge_scatter_sym_f32.argtypes = [vp, vp, vp, l, i, vp, vp, vp]
    L.gs_edge_score_bwd_f32.argtypes = [vp, vp, vp, vp, vp, l, i, i, vp, vp, l, vp]
    L.gs_segment_sum_rows_f32.argtypes = [vp, l, vp, vp, i, i, vp, l, vp]
    L.gs_comm_unique_id.argtypes = [vp]
    L.gs_comm_probe.argtypes = [vp]
    L.gs_comm_ipc_arena.argtypes = [C.c_size_t, C.POINTER(vp), vp]
    L.gs_comm_create_ipc.argtypes = [vp, vp, C.c_size_t, i, i, vp, C.POINTER(vp)]
    L.gs_comm_ipc_error.argtypes = [vp, C.POINTER(i)]
    L.gs_comm_check.argtypes = [vp]
    L.gs_comm_create_rccl.argtypes = [vp, i, i, C.POINTER(vp)]
    L.gs_comm_create_callback.argtypes = [ALLREDUCE_FN, vp, i, i, C.POINTER(vp)]
    L.gs_comm_create_null.argtypes = [i, i, C.POINTER(vp)]
    L.gs_comm_destroy.restype = None
    L.gs_comm_destroy.argtypes = [vp]
    L.gs_comm_rank.argtypes = [vp]
    L.gs_comm_world.argtypes = [vp]
    L.gs_comm_allreduce.argtypes = [vp, vp, l, i, i, vp]
    L.gs_comm_set_timing.argtypes = [vp, i]
    L.gs_comm_read_timing.argtypes = [vp, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_double)]
    L.gs_gcond_outer_step_sharded.argtypes = [vp, i, vp, vp]
    L.gs_gcond_outer_step_sharded_ex.argtypes = [vp, i, i, i, i, vp, vp]
    L.gs_sample_set_cand_cap.argtypes = [i]
    L.gs_gcond_set_h1_rows.argtypes = [i]
    L.gs_gcond_set_fused_chains.argtypes = [i]
    L.gs_gcond_set_inner_persist.argtypes = [i]
    L.gs_gcond_set_edge_src.argtypes = [i]
    L.gs_dedup_set_form.argtypes = [i]
    L.gs_panel_set_plain_share.argtypes = [i]
    L.gs_inner_persist_set_grid.argtypes = [i]
    L.gs_inner_persist_trace.argtypes = [vp]
    L.gs_gcond_set_h1_bits.argtypes = [i]
    L.gs_gcond_set_aux_priority.argtypes = [i]
    L.gs_gcond_set_dw1_generated.argtypes = [i]
    _lib = L
    return L


class GslimError(RuntimeError):
    pass


def check(rc: int):
    if rc != GS_OK:
        raise GslimError(f"libgslim_hip error {rc}: {lib().gs_last_error().decode()}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return vp(t.data_ptr())


def stream():
    import torch
    return vp(torch.cuda.current_stream().cuda_stream)
