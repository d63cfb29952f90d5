"""ctypes binding of libhiprec.so (declared in include/hiprec.h).

The HIP library is THE product path: there is no CPU fallback.  If the shared object is missing or
a symbol is absent, loading fails loudly with a RuntimeError naming the build command.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64
from ctypes import c_size_t, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# HIPREC_LIB selects another build of the same sources, e.g. libhiprec_ieee.so (-DHIPREC_IEEE_DIV: ATen's
# correctly rounded sqrt / division in the Adam / RMSprop denominators, op for op)
LIB_PATH = os.path.join(_HERE, os.environ.get("HIPREC_LIB", "libhiprec.so"))

OPT_SGD, OPT_ADAM, OPT_RMSPROP = 0, 1, 2
OPT_KINDS = {"sgd": OPT_SGD, "adam": OPT_ADAM, "rmsprop": OPT_RMSPROP}

STATUS_USER_OOB, STATUS_ITEM_OOB, STATUS_ROW_OOB, STATUS_ROUTE_OVERFLOW = 1, 2, 4, 8
STATUS_NEG_EXHAUSTED = 16
STATUS_LAZY_TABLE = 32
STATUS_TABLE_FULL = 64


class MfTables(Structure):
    """hiprec_mf_tables (include/hiprec.h)."""

    _fields_ = [
        ("user_emb", c_void_p),
        ("item_emb", c_void_p),
        ("user_bias", c_void_p),
        ("item_bias", c_void_p),
        ("global_bias", c_void_p),
        ("n_users", c_int64),
        ("n_items", c_int64),
        ("dim", c_int32),
        ("_pad", c_int32),
    ]


class Stats(Structure):
    """hiprec_stats (include/hiprec.h) — host-side mirror used to decode a device copy."""

    _fields_ = [
        ("loss", c_float),
        ("reg", c_float),
        ("loss_sum", c_double),
        ("reg_sum", c_double),
        ("step", c_int64),
        ("beta1", c_double),
        ("beta2", c_double),
        ("beta1_pow", c_double),
        ("beta2_pow", c_double),
        ("status", c_uint32),
        ("_pad", c_uint32),
    ]


NCF_MAX_LAYERS = 8


class NcfPlan(Structure):
    """hiprec_ncf_plan (include/hiprec.h)."""

    _fields_ = (
        [(n, c_void_p) for n in ("user_mlp", "item_mlp", "user_mf", "item_mf",
                                 "g_user_mlp", "g_item_mlp", "g_user_mf", "g_item_mf")]
        + [("n_users", c_int64), ("n_items", c_int64),
           ("dim_mlp", c_int32), ("dim_mf", c_int32), ("n_layers", c_int32), ("relu_input", c_int32),
           ("layer_in", c_int32 * NCF_MAX_LAYERS), ("layer_out", c_int32 * NCF_MAX_LAYERS),
           ("fc_w", c_void_p * NCF_MAX_LAYERS), ("fc_b", c_void_p * NCF_MAX_LAYERS),
           ("g_fc_w", c_void_p * NCF_MAX_LAYERS), ("g_fc_b", c_void_p * NCF_MAX_LAYERS),
           ("out_w", c_void_p), ("out_b", c_void_p), ("g_out_w", c_void_p), ("g_out_b", c_void_p),
           ("max_batch", c_int64),
           ("act", c_void_p * (NCF_MAX_LAYERS + 1)), ("dact", c_void_p * (NCF_MAX_LAYERS + 1)),
           ("mf", c_void_p), ("dmf", c_void_p), ("scores", c_void_p),
           ("keep", c_void_p * NCF_MAX_LAYERS), ("keep_scale", c_float), ("_pad", c_int32)]
    )


class Csr(Structure):
    """hiprec_csr (include/hiprec.h)."""

    _fields_ = [("rowptr", c_void_p), ("col", c_void_p), ("val", c_void_p), ("eid", c_void_p),
                ("n_rows", c_int64), ("nnz", c_int64), ("slice_row", c_void_p)]


class SlicedCsr(Structure):
    """hiprec_sliced_csr (include/hiprec.h)."""

    _fields_ = [("chunks", c_void_p), ("col16", c_void_p), ("val", c_void_p), ("eid", c_void_p),
                ("sub_row", c_void_p), ("sub_chunk", c_void_p), ("spill_row", c_void_p), ("spill_ptr", c_void_p),
                ("empty_row", c_void_p), ("empty_ptr", c_void_p), ("row_scale", c_void_p), ("col_scale", c_void_p),
                ("n_rows", c_int64), ("n_slots", c_int64),
                ("n_groups", c_int32), ("subs_per_group", c_int32), ("n_chunks", c_int32), ("row_cap", c_int32),
                ("lane_slots", c_int32), ("pad_slot", c_int32)]


class LightGcnPlan(Structure):
    """hiprec_lightgcn_plan (include/hiprec.h)."""

    _fields_ = [("a", Csr), ("at", Csr), ("n_users", c_int64), ("n_items", c_int64),
                ("dim", c_int32), ("n_layers", c_int32), ("decay", c_float), ("_pad", c_int32)] + \
               [(n, c_void_p) for n in ("e0", "g", "xa", "xb", "acc", "da", "db", "zero_ws")] + \
               [("zero_ws_floats", c_int64), ("sa", SlicedCsr), ("sat", SlicedCsr), ("slice_w", c_int32),
                ("dropped_ready", c_int32), ("sliced_ws", c_void_p), ("sliced_ws_floats", c_int64)]


class PgmfTables(Structure):
    """hiprec_pgmf_tables (include/hiprec.h)."""

    _fields_ = [("user_memory", c_void_p), ("item_memory", c_void_p), ("v", c_void_p),
                ("n_users", c_int64), ("n_items", c_int64), ("dim", c_int32), ("_pad", c_int32)]


class T2vTables(Structure):
    """hiprec_t2v_tables (include/hiprec.h)."""

    _fields_ = [("user_emb", c_void_p), ("item_emb1", c_void_p), ("item_emb2", c_void_p),
                ("user_bias", c_void_p), ("item_bias", c_void_p),
                ("n_users", c_int64), ("n_items", c_int64), ("dim", c_int32), ("_pad", c_int32)]


class ShardPlan(Structure):
    """hiprec_shard_plan (include/hiprec.h)."""

    _fields_ = [("world", c_int32), ("rank", c_int32), ("n_steps", c_int64), ("cap", c_int64),
                ("local_batch", c_int64), ("n_local", c_int64), ("users", c_void_p), ("pos_slot", c_void_p),
                ("neg_slot", c_void_p), ("own", c_void_p), ("total", c_void_p), ("total_stride", c_int64),
                ("in_idx", c_void_p), ("ex_req", c_void_p), ("ex_in", c_void_p), ("in_off_host", c_void_p),
                ("n_slots_host", c_void_p), ("req_cnt_host", c_void_p), ("in_cnt_host", c_void_p),
                ("slot_shared", c_void_p), ("slot_stride", c_int64), ("dup_bits", c_void_p), ("dup_words", c_int64),
                ("cidx", c_void_p), ("rows", c_void_p), ("row_cap", c_int64), ("counts", c_void_p)]


class ShardBufs(Structure):
    """hiprec_shard_bufs (include/hiprec.h)."""

    _fields_ = [("w_flat", c_void_p), ("n_users_local", c_int64), ("n_items_local", c_int64), ("dim", c_int32),
                ("_pad", c_int32), ("payload", c_void_p), ("g_recv", c_void_p), ("fetched", c_void_p),
                ("g_send", c_void_p), ("arrived", c_void_p), ("acc", c_void_p), ("scratch", c_void_p),
                ("g_flat", c_void_p), ("m_flat", c_void_p), ("v_flat", c_void_p),
                ("stamp_u", c_void_p), ("stamp_i", c_void_p), ("lazy_scalars", c_void_p), ("lazy_scalars_cap", c_int64),
                ("cbuf", c_void_p), ("cbias", c_void_p)]


class LazyState(Structure):
    """hiprec_lazy_state (include/hiprec.h)."""

    _fields_ = [("w", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("n_users", c_int64),
                ("n_items", c_int64), ("dim", c_int32), ("kind", c_int32), ("stamp_u", c_void_p), ("stamp_i", c_void_p),
                ("scalars", c_void_p), ("scalars_cap", c_int32), ("_pad", c_int32), ("lr", c_double),
                ("beta1", c_double), ("beta2", c_double), ("eps", c_double)]


class LazyRows(Structure):
    """hiprec_lazy_rows (include/hiprec.h)."""

    _fields_ = [("users", c_void_p), ("n_users", c_int64), ("items_a", c_void_p), ("n_items_a", c_int64),
                ("items_b", c_void_p), ("n_items_b", c_int64), ("items_c", c_void_p), ("n_items_c", c_int64)]


SHARD_EXCHANGE_SELF = 1   # include/hiprec.h HIPREC_SHARD_EXCHANGE_SELF


class NcclFns(Structure):
    """hiprec_nccl_fns (include/hiprec.h)."""

    _fields_ = [("send", c_void_p), ("recv", c_void_p), ("group_start", c_void_p), ("group_end", c_void_p)]


NGCF_MAX_LAYERS = 6


class NgcfPlan(Structure):
    """hiprec_ngcf_plan (include/hiprec.h)."""

    _fields_ = ([("a", Csr), ("at", Csr), ("n_users", c_int64), ("n_items", c_int64), ("n_layers", c_int32),
                 ("dim", c_int32 * (NGCF_MAX_LAYERS + 1)), ("decay", c_float), ("inv_reg_batch", c_float),
                 ("e0", c_void_p), ("g_e0", c_void_p)]
                + [(n, c_void_p * NGCF_MAX_LAYERS) for n in ("gc_w", "gc_b", "bi_w", "bi_b", "g_gc_w", "g_gc_b", "g_bi_w",
                                                "g_bi_b", "side", "bi_in", "sum_pre", "bi_pre", "ego", "nrm")]
                + [("all", c_void_p), ("keep", c_void_p * NGCF_MAX_LAYERS), ("keep_scale", c_float * NGCF_MAX_LAYERS),
                   ("keep_prob", c_float * NGCF_MAX_LAYERS), ("keep_seed", ctypes.c_uint64),
                   ("keep_step", ctypes.c_uint64), ("keep_gen", c_int32), ("_pad", c_int32),
                   ("d_all", c_void_p), ("d_sum", c_void_p), ("d_bi", c_void_p), ("d_side", c_void_p),
                   ("d_bi_in", c_void_p), ("d_ego", c_void_p * 2), ("spmm_tmp", c_void_p * NGCF_MAX_LAYERS),
                   ("zero_ws", c_void_p), ("zero_ws_floats", c_int64), ("sa", SlicedCsr), ("sat", SlicedCsr),
                   ("slice_w", c_int32), ("_pad2", c_int32), ("sliced_src", c_void_p),
                   ("sliced_src_floats", c_int64), ("d_sum_l", c_void_p * NGCF_MAX_LAYERS),
                   ("d_bi_l", c_void_p * NGCF_MAX_LAYERS)])


class FusedStep(Structure):
    """hiprec_fused_step (include/hiprec.h)."""

    _fields_ = [("kind", c_int32), ("dim", c_int32), ("n_users", c_int64), ("n_items", c_int64)] + \
               [(n, c_void_p) for n in ("w_read", "g_prev", "m_read", "v_read", "w_write", "m_write", "v_write",
                                        "g_cur", "g_zero", "scratch_prev", "scratch_cur")] + \
               [("lr", c_double), ("beta1", c_double), ("beta2", c_double), ("eps", c_double),
                ("reg_coef", c_float), ("_pad", c_int32)]


class DpStep(Structure):
    """hiprec_dp_step (include/hiprec.h)."""

    _fields_ = [("w", MfTables), ("g", MfTables), ("stats", c_void_p), ("scratch", c_void_p),
                ("scratch_bytes", c_size_t), ("loss_reg_out", c_void_p), ("w_flat", c_void_p),
                ("g_flat", c_void_p), ("m_flat", c_void_p), ("v_flat", c_void_p), ("n_flat", c_int64),
                ("lr", c_double), ("beta1", c_double), ("beta2", c_double), ("eps", c_double),
                ("reg_coef", c_float), ("loss_kind", c_int32), ("opt_kind", c_int32), ("_pad", c_int32)]


# name -> (restype, argtypes); every symbol of include/hiprec.h must be listed here
_P = c_void_p
_T = POINTER(MfTables)
SIGNATURES = {
    "hiprec_version": (c_int, []),
    "hiprec_source_hash": (ctypes.c_char_p, []),
    "hiprec_last_error": (c_char_p, []),
    "hiprec_stats_bytes": (c_size_t, []),
    "hiprec_scratch_bytes": (c_size_t, [c_int64]),
    "hiprec_stats_reset": (c_int, [_P, c_double, c_double, _P]),
    "hiprec_stats_advance_step": (c_int, [_P, _P]),
    "hiprec_stats_set_step": (c_int, [_P, c_int64, c_double, c_double, _P]),
    "hiprec_stats_begin_epoch": (c_int, [_P, _P]),
    "hiprec_gather_rows": (c_int, [_P, c_int64, c_int32, _P, c_int64, _P, _P, _P]),
    "hiprec_route_bucket": (c_int, [_P, c_int64, c_int32, c_int64, _P, _P, _P, _P]),
    "hiprec_shard_route_triples": (c_int, [_P, _P, _P, c_int64, c_int32, c_int64, _P, _P, _P, _P]),
    "hiprec_shard_route_items": (c_int, [_P, c_int64, c_int32, c_int64, _P, _P, _P, _P, _P, _P, _P]),
    "hiprec_shard_gather_payload": (c_int, [_P, _P, c_int64, c_int32, _P, c_int64, c_int32, _P, _P, _P, _P]),
    "hiprec_shard_split_rows": (c_int, [_P, c_int64, c_int32, _P, _P, _P]),
    "hiprec_shard_join_rows": (c_int, [_P, _P, c_int64, c_int32, _P, _P]),
    "hiprec_scatter_add_rows": (c_int, [_P, c_int64, c_int32, _P, _P, c_int64, c_int64, _P, _P]),
    "hiprec_mf_predict": (c_int, [_T, _P, _P, c_int64, _P, _P, _P]),
    "hiprec_mf_forward": (c_int, [_T, _P, _P, c_int64, _P, _P, _P, _P]),
    "hiprec_mf_bpr_grad": (
        c_int,
        [_T, _T, _P, _P, _P, _P, c_int64, c_float, c_float, _P, _P, c_size_t, _P],
    ),
    "hiprec_mf_bce_grad": (
        c_int,
        [_T, _T, _P, _P, _P, _P, c_int64, c_float, c_float, _P, _P, c_size_t, _P],
    ),
    "hiprec_finalize_stats": (c_int, [_P, _P, _P, _P, _P]),
    "hiprec_opt_dense_step": (
        c_int,
        [c_int, _P, _P, _P, _P, c_int64, c_double, c_double, c_double, c_double, _P, _P, c_int64,
         _P],
    ),
    "hiprec_mf_sgd_rows": (
        c_int,
        [_T, _T, _P, _P, _P, _P, c_int64, c_double, _P, _P, c_int32, _P, _P, _P],
    ),
    "hiprec_ncf_plan_bytes": (c_size_t, []),
    "hiprec_gemm_f32": (
        c_int,
        [c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P],
    ),
    "hiprec_ncf_forward": (c_int, [POINTER(NcfPlan), _P, _P, c_int64, _P, _P]),
    "hiprec_ncf_grad": (
        c_int,
        [POINTER(NcfPlan), _P, _P, _P, c_int64, c_float, _P, _P, c_size_t, _P],
    ),
    "hiprec_ncf_step": (
        c_int,
        [POINTER(NcfPlan), _P, _P, _P, c_int64, c_float, c_int, _P, _P, _P, _P, c_int64, c_int64, c_int64, c_double,
         c_double, c_double, c_double, _P, _P, c_size_t, _P],
    ),
    "hiprec_lightgcn_plan_bytes": (c_size_t, []),
    "hiprec_spmm_csr": (c_int, [POINTER(Csr), _P, c_float, _P, _P, _P, c_int32, _P]),
    "hiprec_sliced_width": (c_int32, [c_int64, c_int32]),
    "hiprec_sliced_row_cap": (c_int32, [c_int64, c_int32]),
    "hiprec_to_sliced": (c_int, [_P, c_int64, c_int32, c_int32, _P, _P, _P]),
    "hiprec_from_sliced": (c_int, [_P, c_int64, c_int32, c_int32, _P, c_int32, _P]),
    "hiprec_sliced_drop_values": (c_int, [POINTER(SlicedCsr), _P, _P, _P]),
    "hiprec_spmm_sliced": (c_int, [POINTER(SlicedCsr), _P, c_float, _P, _P, _P, c_int32, c_int32, c_int32, _P]),
    "hiprec_edge_dropout_mask": (c_int, [_P, c_int64, c_float, ctypes.c_uint64, ctypes.c_uint64, _P]),
    "hiprec_lightgcn_step_values": (c_int, [POINTER(LightGcnPlan), _P, c_float, c_int32, ctypes.c_uint64,
                                            ctypes.c_uint64, _P]),
    "hiprec_lightgcn_opt_stage": (c_int, [POINTER(LightGcnPlan), c_int32, _P, _P, _P, c_double, c_double, c_double,
                                          c_double, _P, _P, c_float, ctypes.c_uint64, ctypes.c_uint64, _P]),
    "hiprec_lightgcn_propagate": (c_int, [POINTER(LightGcnPlan), _P, c_float, _P]),
    "hiprec_lightgcn_predict": (c_int, [POINTER(LightGcnPlan), _P, _P, c_int64, _P, _P, _P]),
    "hiprec_lightgcn_grad": (
        c_int,
        [POINTER(LightGcnPlan), _P, c_float, _P, _P, _P, c_int64, c_float, _P, _P, c_size_t, _P],
    ),
    "hiprec_random_permutation": (c_int, [_P, c_int64, ctypes.c_uint64, _P]),
    "hiprec_fused_step_bytes": (c_size_t, []),
    "hiprec_mf_bpr_fused_step": (c_int, [POINTER(FusedStep), _P, _P, _P, c_int64, c_int64, c_float, _P, _P]),
    "hiprec_dp_step_bytes": (c_size_t, []),
    "hiprec_mf_dp_step_begin": (c_int, [POINTER(DpStep), _P, _P, _P, c_int64, c_float, _P]),
    "hiprec_mf_dp_step_end": (c_int, [POINTER(DpStep), _P]),
    "hiprec_sample_negatives": (
        c_int,
        [_P, _P, c_int64, c_int64, _P, c_int64, c_int32, ctypes.c_uint64, _P, _P, _P],
    ),
    "hiprec_pgmf_workspace_bytes": (c_size_t, [c_int32]),
    "hiprec_pgmf_bpr_grad": (
        c_int,
        [POINTER(PgmfTables), POINTER(PgmfTables), _P, _P, _P, c_int64, c_float, c_float, _P, _P, c_size_t,
         _P, c_size_t, _P],
    ),
    "hiprec_pgmf_epoch": (
        c_int,
        [POINTER(PgmfTables), POINTER(PgmfTables), _P, _P, _P, c_int64, c_int64, c_float, c_float, c_int,
         c_double, c_double, c_double, c_double, _P, _P, _P, _P, c_int64, _P, _P, c_size_t, _P, c_size_t,
         _P, c_size_t, _P],
    ),
    "hiprec_t2v_epoch": (
        c_int,
        [POINTER(T2vTables), POINTER(T2vTables), _P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int32, c_float,
         c_int, c_double, c_double, c_double, c_double, _P, _P, _P, _P, c_int64, _P, _P, c_size_t, _P],
    ),
    "hiprec_clip_workspace_bytes": (c_size_t, []),
    "hiprec_clip_grad_norm": (c_int, [_P, c_int64, c_float, _P, c_size_t, _P]),
    "hiprec_clip_opt_dense_step": (
        c_int,
        [c_int, _P, _P, _P, _P, c_int64, c_double, c_double, c_double, c_double, _P, _P, c_int64, c_float, _P, c_size_t,
         _P],
    ),
    "hiprec_t2v_grad": (
        c_int,
        [POINTER(T2vTables), POINTER(T2vTables), _P, _P, _P, _P, _P, _P, c_int64, c_int32, c_float, _P, _P,
         c_size_t, _P],
    ),
    "hiprec_t2v_predict": (c_int, [POINTER(T2vTables), _P, _P, c_int64, _P, _P, _P]),
    "hiprec_alias_sample": (c_int, [_P, _P, _P, c_int64, ctypes.c_uint64, _P, c_int64, _P]),
    "hiprec_ngcf_plan_bytes": (c_size_t, []),
    "hiprec_ngcf_forward": (c_int, [POINTER(NgcfPlan), c_int, _P]),
    "hiprec_ngcf_predict": (c_int, [POINTER(NgcfPlan), _P, _P, c_int64, _P, _P, _P]),
    "hiprec_ngcf_grad": (c_int, [POINTER(NgcfPlan), _P, _P, _P, c_int64, c_float, _P, _P, c_size_t, _P]),
    "hiprec_csr_n_slices": (c_int64, [c_int64]),
    "hiprec_csr_slice_rows": (c_int, [POINTER(Csr), _P, c_int64, _P]),
    "hiprec_rank_metrics_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "hiprec_rank_metrics": (
        c_int,
        [_P, c_int64, _P, _P, POINTER(c_int32), c_int32, _P, c_size_t, _P, _P],
    ),
    "hiprec_stage_epoch": (c_int, [_P, _P, _P, c_int32, _P, c_int64, c_int64, _P, _P, _P, _P]),
    "hiprec_stage_epoch_shuffled": (c_int, [_P, _P, _P, c_int32, ctypes.c_uint64, c_int64, c_int64, _P, _P, _P, _P]),
    "hiprec_mf_bce_epoch": (
        c_int,
        [_T, _T, _P, _P, _P, _P, c_int64, c_int64, c_float, c_int]
        + [c_double, c_double, c_double, c_double]
        + [_P, _P, _P, _P, c_int64, _P, _P, c_int32, _P, _P, c_size_t, _P],
    ),
    "hiprec_mf_bpr_epoch_fused": (
        c_int,
        [c_int, _P, _P, _P, _P, _P, c_int64, c_int64, c_int32, _P, _P, _P, c_int64, c_int64, c_float,
         c_double, c_double, c_double, c_double, _P, POINTER(c_int32), _P],
    ),
    "hiprec_mf_bpr_epoch_fused_range": (
        c_int,
        [c_int, _P, _P, _P, _P, _P, c_int64, c_int64, c_int32, _P, _P, _P, c_int64, c_int64, c_int64, c_int64,
         c_float, c_double, c_double, c_double, c_double, _P, POINTER(c_int32), _P],
    ),
    "hiprec_mf_bpr_dp_epoch_fused_range": (
        c_int,
        [c_int, _P, _P, c_int64, _P, _P, c_int64, c_int64, c_int64, c_int32, _P, _P, _P, c_int64, c_int64, c_int64,
         c_int64, c_int32, c_float, c_double, c_double, c_double, c_double, _P, _P, _P, POINTER(c_int32), _P],
    ),
    "hiprec_mf_bpr_epoch_owned": (
        c_int,
        [_P, c_int64, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, c_int64, _P, _P, c_int64,
         c_int64, c_int64, c_int64, c_float, c_double, _P, _P],
    ),
    "hiprec_ownership_table_bits": (c_int32, [c_int64]),
    "hiprec_ownership_ws_ints": (c_int64, [c_int64, c_int64, c_int32]),
    "hiprec_batch_row_ownership": (
        c_int, [_P, _P, _P, c_int64, c_int64, c_int64, c_int64, c_int32, _P, _P, _P, _P]),
    "hiprec_stage_grouped_ws_ints": (c_int64, [c_int64, c_int64, c_int64]),
    "hiprec_stage_epoch_grouped": (
        c_int, [_P, _P, _P, _P, c_int32, ctypes.c_uint64, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P]),
    "hiprec_mf_pull_chunk": (c_int32, [c_int32]),
    "hiprec_contrib_row_cap": (c_int64, [c_int64, c_int32]),
    "hiprec_batch_row_contrib": (
        c_int, [_P, _P, _P, c_int64, c_int64, c_int64, c_int64, c_int32, c_int32, c_int32, _P, _P, _P, c_int64, _P, _P]),
    "hiprec_mf_bpr_epoch_pull": (
        c_int,
        [_P, c_int64, c_int64, c_int32, _P, _P, _P, _P, c_int64, _P, c_int64, _P, _P, _P, _P, c_int64, c_int64,
         c_int64, c_int64, c_float, c_double, _P, _P],
    ),
    "hiprec_mf_bpr_owned_remote_step": (
        c_int,
        [_P, c_int64, c_int64, c_int32, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_float,
         c_float, c_double, _P, _P, _P],
    ),
    "hiprec_mf_bpr_pull_remote_step": (
        c_int,
        [_P, c_int64, c_int64, c_int32, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, c_int32,
         c_int64, c_float, c_float, c_double, _P, _P, _P],
    ),
    "hiprec_batch_row_ownership_tables": (
        c_int, [_P, _P, _P, c_int64, c_int64, c_int64, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "hiprec_plan_route_tiles": (c_int64, [c_int64, c_int64]),
    "hiprec_plan_route_triples": (
        c_int, [_P, _P, _P, _P, c_int64, c_int64, c_int32, c_int64, c_int64, _P, _P, _P, _P, _P]),
    "hiprec_plan_place_triples": (c_int, [_P, c_int64, _P, c_int32, c_int64, c_int64, _P, _P, _P, _P, _P]),
    "hiprec_plan_slot_ws_ints": (c_int64, [c_int64, c_int32, c_int32]),
    "hiprec_plan_item_slots": (
        c_int, [_P, c_int64, c_int64, c_int32, c_int64, c_int32] + [_P] * 16 + [_P, _P, c_int64, _P, _P]),
    "hiprec_plan_place_requests": (c_int, [_P, c_int64, _P, c_int32, c_int64, _P, _P, _P, _P, c_int64, _P, _P]),
    "hiprec_shard_payload_zero": (
        c_int, [_P, _P, c_int64, c_int32, _P, c_int64, c_int64, c_int64, _P, _P, _P, c_int64, _P, _P, _P]),
    "hiprec_mf_bpr_grad_remote_step": (
        c_int,
        [_P, _P, c_int64, c_int64, c_int32, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, c_int64, c_float, c_float,
         _P, _P, _P],
    ),
    "hiprec_shard_apply_finish": (
        c_int, [_P, _P, c_int64, c_int32, _P, _P, c_int64, c_int64, c_int64, _P, c_double, _P, c_int32, _P, c_double,
                c_int32, _P, _P, _P]),
    "hiprec_shard_planned_steps": (
        c_int, [POINTER(ShardPlan), POINTER(ShardBufs), c_int64, c_int64, c_int32, c_float, c_double, c_double,
                c_double, c_double, POINTER(NcclFns), _P, _P, _P]),
    "hiprec_shard_planned_steps_ex": (
        c_int, [POINTER(ShardPlan), POINTER(ShardBufs), c_int64, c_int64, c_int32, c_float, c_double, c_double,
                c_double, c_double, POINTER(NcclFns), _P, ctypes.c_uint32, _P, _P]),
    "hiprec_gather_epoch": (c_int, [_P, _P, _P, _P, c_int32, ctypes.c_uint64, _P, c_int64, _P, _P, _P, _P]),
    "hiprec_stage_sort_keys": (c_int, [_P, _P, c_int32, ctypes.c_uint64, c_int64, c_int64, c_int64, c_int32, _P, _P]),
    "hiprec_group_epoch_by_item": (c_int, [_P, _P, _P, c_int64, c_int64, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P, _P,
                                           _P, _P]),
    "hiprec_lazy_state_bytes": (c_size_t, []),
    "hiprec_lazy_catchup": (c_int, [POINTER(LazyState), POINTER(LazyRows), _P, _P]),
    "hiprec_lazy_update": (c_int, [POINTER(LazyState), POINTER(LazyRows), _P, _P, _P]),
    "hiprec_lazy_flush": (c_int, [POINTER(LazyState), _P, _P]),
    "hiprec_lazy_mark_current": (c_int, [POINTER(LazyState), _P, _P]),
    "hiprec_mf_epoch_lazy": (c_int, [POINTER(LazyState), _T, _T, _P, _P, _P, c_int32, c_int64, c_int64, c_int32, c_float,
                                     _P, _P, c_size_t, _P]),
    "hiprec_mf_bpr_grad_owned": (c_int, [_P, _P, c_int64, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P, c_int64, c_float,
                                         c_float, _P, _P, _P]),
    "hiprec_mf_epoch_lazy_pull": (c_int, [POINTER(LazyState), _P, _P, _P, _P, c_int64, _P, c_int64, _P, _P, _P, c_int64,
                                          c_int64, c_int32, c_float, _P, _P, _P]),
    "hiprec_mf_epoch_lazy_owned": (c_int, [POINTER(LazyState), _P, _P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int64,
                                           c_int32, c_float, _P, _P, _P]),
    "hiprec_shard_plan_bytes": (c_size_t, []),
    "hiprec_shard_bufs_bytes": (c_size_t, []),
    "hiprec_shard_publish_partials": (c_int, [_P, _P, c_int32, _P, c_int32, _P]),
    "hiprec_shard_apply_rows": (c_int, [_P, _P, c_int64, c_int32, _P, _P, c_int64, c_double, _P, _P]),
    "hiprec_shard_finish_step": (c_int, [_P, c_int32, _P, c_int32, _P, c_double, c_int32, _P, _P]),
    "hiprec_mf_bpr_epoch_sgd_fused": (
        c_int,
        [_P, _P, _P, c_int64, c_int64, c_int32, _P, _P, _P, c_int64, c_int64, c_float, c_double, _P,
         POINTER(c_int32), _P],
    ),
    "hiprec_mf_bpr_epoch": (
        c_int,
        [_T, _T, _P, _P, _P, _P, c_int64, c_int64, c_float, c_int]
        + [c_double, c_double, c_double, c_double]
        + [_P, _P, _P, _P, c_int64, _P, _P, c_int32, _P, _P, c_size_t, _P],
    ),
}

_lib = None


class HiprecError(RuntimeError):
    """A libhiprec call returned a non-zero code."""


def load():
    """Load libhiprec.so (once) and attach the prototypes.  Fails loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libhiprec.so not found at {LIB_PATH}: the HIP extension is the only compute path of "
            "this package (there is no CPU fallback). Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950)."
        )
    # torch must be imported first so that libamdhip64.so.7 resolves to the runtime torch already
    # loaded (one HIP runtime per process: streams and device pointers are shared with torch).
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise RuntimeError(f"libhiprec.so does not export {name}; rebuild it") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.hiprec_stats_bytes() != ctypes.sizeof(Stats):
        raise RuntimeError("hiprec_stats layout mismatch between _lib.py and libhiprec.so")
    if lib.hiprec_lightgcn_plan_bytes() != ctypes.sizeof(LightGcnPlan):
        raise RuntimeError("hiprec_lightgcn_plan layout mismatch between _lib.py and libhiprec.so")
    if lib.hiprec_fused_step_bytes() != ctypes.sizeof(FusedStep):
        raise RuntimeError("hiprec_fused_step layout mismatch between _lib.py and libhiprec.so")
    if lib.hiprec_dp_step_bytes() != ctypes.sizeof(DpStep):
        raise RuntimeError("hiprec_dp_step layout mismatch between _lib.py and libhiprec.so")
    if lib.hiprec_ngcf_plan_bytes() != ctypes.sizeof(NgcfPlan):
        raise RuntimeError("hiprec_ngcf_plan layout mismatch between _lib.py and libhiprec.so")
    if lib.hiprec_lazy_state_bytes() != ctypes.sizeof(LazyState):
        raise RuntimeError("hiprec_lazy_state layout mismatch between _lib.py and libhiprec.so")
    if lib.hiprec_shard_bufs_bytes() != ctypes.sizeof(ShardBufs):
        raise RuntimeError("hiprec_shard_bufs layout mismatch between _lib.py and libhiprec.so")
    if lib.hiprec_ncf_plan_bytes() != ctypes.sizeof(NcfPlan):
        raise RuntimeError("hiprec_ncf_plan layout mismatch between _lib.py and libhiprec.so")
    _lib = _DeviceGuardedLib(lib)
    return _lib


class _Stream(c_void_p):
    """A hipStream_t that remembers which HIP device it belongs to (see _DeviceGuardedLib)."""

    device_index = None


class _DeviceGuardedLib:
    """libhiprec's entry points behind a HIP-device guard.

    A kernel launch goes to the CURRENT HIP device, and torch's default stream has the handle 0 on
    every device, so a call for tensors on ``cuda:1`` made while device 0 is current would run on
    GPU 0's null stream against GPU 1's pointers.  The reference never calls ``set_device`` and
    ``TrainEngine.get_device`` hands out ``cuda:N``, so every stream-taking entry point is wrapped:
    the stream argument built by :func:`stream_ptr` carries its device index, and the wrapper makes
    that device current for the duration of the call.
    """

    def __init__(self, cdll):
        self._cdll = cdll
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(cdll, name)
            setattr(self, name, self._guard(fn) if (restype is c_int and argtypes) else fn)

    @staticmethod
    def _guard(fn):
        import torch

        def call(*args):
            want = None
            for a in reversed(args):  # the stream is the last argument of almost every entry point
                if type(a) is _Stream:
                    want = a.device_index
                    break
            if want is None or want == torch.cuda.current_device():
                return fn(*args)
            with torch.cuda.device(want):
                return fn(*args)

        call.__name__ = fn.__name__
        call.argtypes, call.restype = fn.argtypes, fn.restype
        return call


def check(rc):
    """Raise HiprecError for a non-zero return code of a libhiprec call."""
    if rc != 0:
        msg = load().hiprec_last_error()  # (thread-local buffer of the C library)
        raise HiprecError(f"libhiprec call failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor as c_void_p (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream_ptr(device):
    """Current torch stream of `device` as a hipStream_t (c_void_p)."""
    import torch

    device = torch.device(device)
    st = _Stream(torch.cuda.current_stream(device).cuda_stream)
    st.device_index = device.index if device.index is not None else torch.cuda.current_device()
    return st


LAZY_SCALARS_CAP = 1 << 16     # steps whose Adam bias corrections are tabulated for the lazy replays (csrc/lazy_opt.hip)


def lazy_betas_converge(opt, cap=LAZY_SCALARS_CAP):
    """Beyond the scalars table a lazy replay reads its LAST entry: that is only right once the bias corrections no
    longer move -- 1 - beta^t == 1.0 in double from t = cap - 1 on (default betas: t ~ 36 800).  RMSprop has no
    bias correction.  Checked when the state is set up, not after 65 536 steps of training."""
    if opt.name != "adam":
        return True
    return all(1.0 - float(b) ** (cap - 1) == 1.0 for b in (opt.beta1, opt.beta2))


def lazy_scalars_table(opt, device, cap=LAZY_SCALARS_CAP):
    """Adam's per-step scalars table [cap, 2], zero except for the last entry, which holds the CONVERGED pair
    (lr, 1): steps at or beyond the table are checked against it and replays of such steps read it -- whether or not
    step cap - 1 itself was ever taken by a lazy update (a restored clock beyond the table, or a dense sweep at that
    step, used to leave it zero: a spurious HIPREC_STATUS_LAZY_TABLE)."""
    import torch

    if not lazy_betas_converge(opt, cap):
        raise ValueError(
            f"lazy Adam tabulates the bias corrections of {cap} steps and needs them converged by then "
            f"(1 - beta^{cap - 1} == 1 in double); betas ({opt.beta1}, {opt.beta2}) do not: use dense_opt='sweep'")
    table = torch.zeros((cap, 2), dtype=torch.float32, device=device)
    if opt.name == "adam":
        # (float)(lr / (1 - 0)) and the rcp of (float)sqrt(1 - 0): exactly what a step beyond convergence derives
        table[cap - 1, 0] = float(opt.lr)
        table[cap - 1, 1] = 1.0
    return table


def grow(buf, numel, dtype, device):
    """A work-space tensor of at least `numel` elements: `buf` if it is big enough (same device / dtype), else a new one.
    (Staging buffers are allocated once and kept, not re-made every epoch.)"""
    import torch

    if buf is not None and buf.numel() >= numel and buf.dtype == dtype and buf.device == torch.device(device):
        return buf
    return torch.empty(max(int(numel), 1), dtype=dtype, device=device)
