"""ctypes binding of librechorus_hip.so (the C ABI declared in include/rechorus_hip.h).

There is deliberately no fallback: if the HIP library is missing, importing fails loudly
(`RechorusHipMissing`) instead of routing through torch or the CPU oracle.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# RC_LIB_PATH: an experimental build of the same library (tools/build_variant.py, A/B timing); default = in-tree
LIB_PATH = os.environ.get("RC_LIB_PATH") or os.path.join(_PKG, "librechorus_hip.so")

RC_OK = 0
RC_OPT_SGD, RC_OPT_ADAM, RC_OPT_ADAGRAD, RC_OPT_ADADELTA = 0, 1, 2, 3
RC_SEG_SKIP_SINGLETONS = 1
OPT_BY_NAME = {"SGD": RC_OPT_SGD, "Adam": RC_OPT_ADAM, "Adagrad": RC_OPT_ADAGRAD, "Adadelta": RC_OPT_ADADELTA}


class RechorusHipMissing(ImportError):
    pass


class RechorusHipError(RuntimeError):
    def __init__(self, fn, code, text):
        super().__init__(f"{fn} failed (rc_status {code}): {text}")
        self.code = code


class OptHyper(C.Structure):
    """struct rc_opt_hyper"""
    _fields_ = [
        ("opt", C.c_int),
        ("reserved", C.c_int),
        ("lr", C.c_double),
        ("l2", C.c_double),
        ("beta1", C.c_double),
        ("beta2", C.c_double),
        ("eps", C.c_double),
        ("step", C.c_int64),
    ]


class StepTicket(C.Structure):
    """struct rc_step_ticket: what rc_bprmf_train_step_ahead prepared for which batch (caller-owned, zero-initialised)"""
    _fields_ = [
        ("generation", C.c_uint64),
        ("ws", C.c_uint64),
        ("slot", C.c_int32),
        ("device", C.c_int32),
        ("B", C.c_int32),
        ("C", C.c_int32),
        ("d", C.c_int32),
        ("flavour", C.c_int32),
        ("n_users", C.c_int64),
        ("n_items", C.c_int64),
        ("reserved", C.c_uint64 * 2),
    ]


_p = C.c_void_p
_i = C.c_int
_u64 = C.c_uint64
_tp = C.POINTER(StepTicket)
_i64 = C.c_int64
_f = C.c_float
_sz = C.c_size_t
_hp = C.POINTER(OptHyper)

# name -> (restype, argtypes); must list every symbol include/rechorus_hip.h declares
SIGNATURES = {
    "rc_version": (_i, []),
    "rc_last_error_string": (C.c_char_p, []),
    "rc_device_count": (_i, []),
    "rc_gather_rows": (_i, [_p, _i, _p, _i64, _p, _p]),
    "rc_gather_rows_pair": (_i, [_p, _p, _i, _p, _i64, _p, _p]),
    "rc_gather_dot_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "rc_weighted_row_sum": (_i, [_p, _p, _p, _i, _i, _i, _p, _p]),
    "rc_bpr_loss_fwd_bwd": (_i, [_p, _i, _i, _f, _p, _p, _p]),
    "rc_list_bpr_fwd_bwd": (_i, [_p, _p, _i, _i, _i, _i, _f, _p, _p, _p]),
    "rc_softmax_ce_fwd_bwd": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "rc_list_loss_fwd_bwd": (_i, [_p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p]),
    "rc_fm_second_order_fwd": (_i, [_p, _i64, _i, _i, _p, _p]),
    "rc_fm_second_order_bwd": (_i, [_p, _p, _i64, _i, _i, _p, _p]),
    "rc_fm_second_order_bwd_add": (_i, [_p, _p, _i64, _i, _i, _p, _p, _p]),
    "rc_gather_fields": (_i, [_p, _p, _p, _p, _i, _i64, _i, _i, _p, _p, _p]),
    "rc_gather_fields_pair": (_i, [_p, _p, _p, _p, _p, _i, _i64, _i, _i, _p, _p, _p, _p]),
    "rc_gather_fields_pair_mark": (_i, [_p, _p, _p, _p, _p, _i, _i64, _i, _i, _p, _p, _p, _p, _p, _i, _p]),
    "rc_neumf_zhead_supported": (_i, [_i, _i, _i]),
    "rc_neumf_zhead_workspace_bytes": (_sz, [_i, _i]),
    "rc_neumf_zhead_fwd_bwd": (_i, [_p, _i64, _p, _p, _i64, _p, _i, _i, _i, _i, _f, _p, _p, _p, _i64, _p, _i64, _p, _p, _p, _sz, _p]),
    "rc_seq_offsets": (_i, [_p, _i64, _i, _p, _p]),
    "rc_seq_embed_fwd": (_i, [_p, _p, _p, _p, _i64, _i, _i, _p, _p]),
    "rc_seq_pick_last_fwd": (_i, [_p, _p, _i64, _i, _i, _p, _p]),
    "rc_seq_pick_last_bwd": (_i, [_p, _p, _i64, _i, _i, _p, _p]),
    "rc_seq_pos_grad": (_i, [_p, _p, _i64, _i, _i, _i, _p, _p]),
    "rc_seq_attention_supported": (_i, [_i, _i]),
    "rc_seq_attention_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i64, _i, _i, _i, _p, _p, _p]),
    "rc_seq_attention_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i64, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "rc_seq_add_layernorm_fwd": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _f, _p, C.c_uint32, _p, _p, _p, _p]),
    "rc_seq_add_layernorm_bwd_workspace_bytes": (_sz, [_i]),
    "rc_seq_add_layernorm_bwd": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _f, _p, C.c_uint32, _p, _p, _p, _p, _p, _sz, _p]),
    "rc_list_metrics_supported": (_i, [_i, _i, _i]),
    "rc_list_metrics": (_i, [_p, _p, _p, _i64, _i, _i, _p, _i, _p, _p, _p]),
    "rc_gather_fields_mixed": (_i, [_p, _p, _p, _p, _p, _i64, _p, _i, _i64, _i, _i, _p, _p, _p, _p, _p, _i, _p]),
    "rc_gather_fields_fused": (_i, [_p, _p, _p, _p, _p, _i64, _p, _i, _i64, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _sz, _p, _p]),
    "rc_numeric_field_grads_workspace_bytes": (_sz, [_i64, _i, _i]),
    "rc_numeric_field_grads": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i64, _i, _i, _p, _p, _p, _sz, _p]),
    "rc_bce_ranking_fwd_bwd": (_i, [_p, _i64, _i, _f, _p, _p, _p]),
    "rc_bce_prob_fwd_bwd": (_i, [_p, _p, _i64, _f, _p, _p, _p]),
    "rc_sample_negatives": (_i, [_p, _i64, _i, _i64, _p, _p, C.c_uint64, C.c_uint64, _p, _p]),
    "rc_assemble_candidates": (_i, [_p, _i64, _i, _p, _p, _p, _p, _p, _p]),
    "rc_gather_history": (_i, [_p, _i64, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "rc_target_rank": (_i, [_p, _i64, _i, _p, _p]),
    "rc_full_catalogue_rank_supported": (_i, [_i]),
    "rc_full_catalogue_rank": (_i, [_p, _p, _p, _p, _i64, _i64, _i, _p, _p, _p, _p, _p]),
    "rc_route_workspace_bytes": (_sz, [_i64, _i]),
    "rc_route_by_owner": (_i, [_p, _i64, _i, _i64, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "rc_owner_unpack": (_i, [_p, _i64, _p, _p, _p, _p]),
    "rc_owner_backward_workspace_bytes": (_sz, [_i64]),
    "rc_owner_backward": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, _p, _i64, _i64, _hp, _p, _p, _sz, _p]),
    "rc_reduce_sum": (_i, [_p, _i64, _f, _p, _p]),
    "rc_bprmf_fwd_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _p, _p, _p, _p, _p]),
    "rc_bprmf_fused_supported": (_i, [_i, _i]),
    "rc_bprmf_fwd_bwd_update": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _hp, _p, _p, _p, _p, _p]),
    "rc_segment_heads": (_i, [_p, _p, _i64, _i, _p, _p, _p, _p]),
    "rc_sort_workspace_bytes": (_sz, [_i64]),
    "rc_sort_ids": (_i, [_p, _i64, _i64, _p, _p, _p, _sz, _p]),
    "rc_segmented_workspace_bytes": (_sz, [_i64, _i]),
    "rc_segmented_update": (_i, [_p, _p, _p, _i, _p, _p, _i64, _p, _p, _p, _i, _hp, _p, _p, _p, _i, _p, _sz, _p]),
    "rc_dense_update": (_i, [_p, _p, _p, _p, _i64, _hp, _p]),
    "rc_dense_update_multi": (_i, [_p, _p, _p, _p, _p, _p, _i, _p]),
    "rc_dense_update_multi_dev": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _p]),
    "rc_step_increment": (_i, [_p, _p]),
    "rc_step_increment2": (_i, [_p, _p, _p]),
    "rc_dense_update_rows_dev": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "rc_stage_batch": (_i, [_p, _i64, _p, _i64, _p, _i64, _p, _p]),
    "rc_segmented_update2": (_i, [_p, _p, _p, _i, _p, _p, _i64, _p, _p, _p, _i, _p, _i64, _i64, _i64, _hp, _p, _p,
                                  _p, _i, _p, _sz, _p]),
    "rc_ctr_head_fwd_bwd": (_i, [_p, _p, _i, _p, _p, _p, _i64, _p, _p, _p, _p]),
    "rc_ctr_head_fwd_bwd_sums": (_i, [_p, _p, _i, _p, _p, _p, _i64, _p, _p, _p, _p, _p]),
    "rc_ctr_head_fwd_full": (_i, [_p, _p, _i, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p]),
    "rc_ctr_head_bwd": (_i, [_p, _p, _p, _i64, _i, _p, _p, _p, _p]),
    "rc_small_row_sums_supported": (_i, [_i64, _i64, _i]),
    "rc_small_row_sums_workspace_bytes": (_sz, [_i64]),
    "rc_small_row_sums": (_i, [_p, _i64, _i64, _p, _i, _p, _p, _sz, _p]),
    "rc_small_row_sums_again": (_i, [_i64, _i64, _p, _i, _p, _p, _sz, _p]),
    "rc_small_row_sums_pair": (_i, [_p, _i64, _i64, _p, _i, _p, _p, _p, _p, _sz, _p]),
    "rc_small_row_sums_pair_numeric": (_i, [_p, _i64, _i64, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _i, _p, _p, _p, _sz, _p]),
    "rc_small_row_sums_planned": (_i, [_i64, _i64, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "rc_segmented_rows_workspace_bytes": (_sz, [_i64, _i64, _i]),
    "rc_segmented_update_rows": (_i, [_p, _p, _p, _i, _i64, _p, _p, _i64, _p, _p, _p, _i, _p, _i64, _hp, _p, _p, _sz, _p]),
    "rc_segmented_update_rows_dev": (_i, [_p, _p, _p, _i, _i64, _p, _p, _i64, _p, _p, _p, _i, _p, _i64, _hp, _p, _p, _p, _sz, _p]),
    "rc_rows_plan_supported": (_i, [_i64, _i64, _i]),
    "rc_rows_plan_workspace_bytes": (_sz, [_i64, _i64, _i]),
    "rc_rows_plan_build": (_i, [_p, _i64, _p, _i64, _p, _i, _i64, _i, _p, _sz, _p]),
    "rc_rows_plan_views": (_i, [_p, _i64, _i64, _i, _p, _p, _p, _p, _p]),
    "rc_rows_plan_update": (_i, [_p, _p, _p, _i, _i64, _i64, _p, _p, _p, _i, _p, _i64, _hp, _p, _p, _p, _sz, _p]),
    "rc_segmented_update_pair": (_i, [_p] * 6 + [_i, _p, _p, _i64, _p, _p, _hp, _p, _p, _p, _p, _p, _sz, _p]),
    "rc_sort_ids2": (_i, [_p, _i64, _p, _i64, _i64, _i64, _p, _p, _p, _sz, _p]),
    "rc_sasrec_supported": (_i, [_i, _i, _i, _i]),
    "rc_sasrec_dense_param_count": (_i, [_i]),
    "rc_sasrec_workspace_bytes": (_sz, [_i, _i, _i]),
    "rc_sasrec_fwd": (_i, [_p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "rc_sasrec_bwd": (_i, [_p, _i, _i, _p, _i, _i, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "rc_sasrec_pos_grad_workspace_bytes": (_sz, [_i, _i, _i]),
    "rc_sasrec_pos_grad": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _sz, _p]),
    "rc_sasrec_batch_state_floats": (_sz, [_i, _i, _i, _i]),
    "rc_sasrec_batch_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "rc_sasrec_batch_fwd": (_i, [_p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "rc_sasrec_batch_bwd": (_i, [_p, _i, _i, _p, _i, _i, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "rc_sasrec_batch_fwd_dropout": (_i, [_p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _f, _p, _p, _p, _p, _sz, _p]),
    "rc_sasrec_batch_bwd_dropout": (_i, [_p, _i, _i, _p, _i, _i, _i, _f, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "rc_sasrec_batch_bwd_part": (_i, [_p, _i, _i, _p, _i, _i, _i, _f, _p, _p, _p, _p, _p, _p, _sz, _i, _p]),
    "rc_sasrec_batch_bwd_splits": (_i, [_i, _i, _i, _i, _i, _f]),
    "rc_neumf_supported": (_i, [_i, _i]),
    "rc_neumf_fwd": (_i, [_p] * 9 + [_i, _i, _i, _i, _p, _p]),
    "rc_neumf_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "rc_neumf_bwd": (_i, [_p] * 10 + [_i, _i, _i, _i] + [_p] * 7 + [_p, _sz, _p]),
    "rc_neumf_train_step_supported": (_i, [_i, _i, _i]),
    "rc_neumf_train_step_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "rc_neumf_train_step_marks_bytes": (_sz, [_i64]),
    "rc_neumf_train_step": (_i, [_p] * 13 + [_i, _i, _i, _i, _i64, _p, _hp, _f] + [_p] * 9 + [_p, _sz, _p]),
    "rc_neumf_mark_rows": (_i, [_p, _i64, _i64, _p, _p]),
    "rc_neumf_unmark_rows": (_i, [_p, _i64, _i64, _p, _p]),
    "rc_neumf_train_step_dropout": (_i, [_p] * 13 + [_i, _i, _i, _i, _i64, _p, _i, _hp, _f, _f, _p] + [_p] * 9 + [_p, _sz, _p]),
    "rc_neumf_train_step_marked": (_i, [_p] * 13 + [_i, _i, _i, _i, _i64, _p, _hp, _f] + [_p] * 9 + [_p, _sz, _p]),
    "rc_neumf_head_fwd_bwd": (_i, [_p, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p, _i64, _p, _p, _i64,
                                   _p, _p, _p, _p, _sz, _p]),
    "rc_bench_mix": (_i, [_p, _i, _p, _i64, _f, _i, _p, C.POINTER(C.c_float), _p]),
    "rc_bench_mfma": (_i, [_i, _p, C.POINTER(C.c_float), _p]),
    "rc_linear_fwd": (_i, [_p, _p, _p, _i64, _i, _i, _i, _f, _p, C.c_uint32, _p, _p]),
    "rc_linear_fwd_workspace_bytes": (_sz, [_i64, _i, _i]),
    "rc_linear_fwd_ws": (_i, [_p, _p, _p, _i64, _i, _i, _i, _f, _p, C.c_uint32, _p, _p, _sz, _p]),
    "rc_linear_bwd_workspace_bytes": (_sz, [_i64, _i, _i]),
    "rc_linear_bwd": (_i, [_p, _p, _p, _p, _i64, _i, _i, _f, _p, _p, _p, _p, _sz, _p]),
    "rc_linear_bwd_chain": (_i, [_p, _p, _p, _p, _i64, _i, _i, _f, _i, _f, _p, _p, _p, _p, _sz, _p]),
    "rc_tower_tail_supported": (_i, [_i64, _i, _i]),
    "rc_tower_tail_workspace_bytes": (_sz, [_i64, _i, _i]),
    "rc_tower_tail_fwd": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _f, _p, C.c_uint32, _p, _p, _p]),
    "rc_tower_tail_bwd": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _f, _i, _f, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "rc_neumf_fwd_dropout": (_i, [_p] * 9 + [_i, _i, _i, _i, _f, _p, _p, _p]),
    "rc_neumf_bwd_dropout": (_i, [_p] * 10 + [_i, _i, _i, _i, _f, _p] + [_p] * 7 + [_p, _sz, _p]),
    "rc_bucket_plan_supported": (_i, [_i64, _i64, _i64, _i64]),
    "rc_bucket_plan_workspace_bytes": (_sz, [_i64, _i64]),
    "rc_bucket_plan_flags_bytes": (_sz, [_i64]),
    "rc_bucket_plan_status_ptr": (_p, [_p, _i64, _i64]),
    "rc_bucket_plan": (_i, [_p, _i64, _i64, _p, _i64, _i64, _i, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "rc_plan_update_workspace_bytes": (_sz, [_i64, _i]),
    "rc_plan_update": (_i, [_p, _p, _p, _i, _p, _p, _p, _i64, _p, _p, _p, _i, _p, _i64, _hp, _p, _sz, _p]),
    "rc_plan_update_pair": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i64, _p, _p, _i64, _hp, _p, _sz, _p]),
    "rc_plan_update_pair_zeroed": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i64, _p, _p, _i64, _hp, _p, _p, _sz, _p]),
    "rc_plan_update_pair_block": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i64, _p, _i64, _i64, _hp, _p, _p, _sz, _p]),
    "rc_plan_row_sums": (_i, [_p, _i, _p, _p, _p, _i64, _p, _p, _p, _i, _p, _i64, _p, _sz, _p]),
    "rc_plan_distinct": (_i, [_p, _p, _p, _i64, _i64, _p, _p, _p]),
    "rc_bprmf_step_workspace_bytes": (_sz, [_i, _i, _i]),
    "rc_bprmf_step_pipeline": (_i, [_i]),
    "rc_bprmf_train_step": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i64, _i64, _hp, _f,
                                 _p, _p, _p, _sz, _p, C.POINTER(C.c_float)]),
    "rc_bprmf_step_ahead_reset": (_i, [_tp, _p]),
    "rc_bprmf_train_step_ahead": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _u64, _p, _p, _u64, _tp, _i, _i, _i, _i64, _i64, _hp, _f,
                                       _p, _p, _p, _sz, _p, C.POINTER(C.c_float)]),
    "rc_bucket_bitmap_bytes": (_sz, [_i64]),
    "rc_bucket_multi_bitmap": (_i, [_p, _i64, _i64, _p, _p, _sz, _p]),
    "rc_bprmf_fwd_bwd_update_bitmap": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _hp, _p, _p, _p, _p, _p]),
}

_lib = None


def load():
    """dlopen the library once and attach the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RechorusHipMissing(
            f"{LIB_PATH} not found: build it with `python -m rechorus_amd.csrc.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU/torch fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RechorusHipMissing(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(fn_name, code):
    if code != RC_OK:
        text = load().rc_last_error_string()
        raise RechorusHipError(fn_name, code, text.decode() if text else "")


def call(fn_name, *args):
    """Call an int-returning entry point and raise on a non-zero status."""
    lib = load()
    code = getattr(lib, fn_name)(*args)
    check(fn_name, code)
