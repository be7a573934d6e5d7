"""ctypes binding of libhgt_hip.so (the C ABI declared in include/hgt_hip.h).

The library is built in-tree (pyhgt_amd/lib/libhgt_hip.so, see __graft_entry__.build() or
`make -C pyhgt_amd/csrc`).  There is NO fallback: if the shared object is missing or a call
returns an error code, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HGT_LIB_PATH", os.path.join(_HERE, "lib", "libhgt_hip.so"))   # override: development A/B builds

HGT_RTE_LEN = 240
HGT_N_PHASE_EVENTS = 7
HGT_FLAG_NO_FUSED_UPDATE = 1
HGT_FLAG_VALU_AGGREGATE = 2
HGT_FLAG_MFMA_LOGITS = 4
HGT_FLAG_VALU_LOGITS = 8
HGT_FLAG_ITEM_AGGREGATE = 16
HGT_FLAG_NO_ITEM_AGGREGATE = 32
HGT_FLAG_FUSED_ANY_SIZE = 64
HGT_FLAG_DETERMINISTIC_HUBS = 128
HGT_FLAG_SINGLE_PASS = 256
HGT_FLAG_RING_AGGREGATE = 512
HGT_FLAG_XS_GEMM_ALWAYS = 1024
HGT_FLAG_XS_GEMM_NEVER = 2048
HGT_FLAG_NO_TILE_GEMM = 4096
HGT_FLAG_NO_MERGE_UPDATE = 8192
HGT_FLAG_NO_COOP_EDGE = 16384
HGT_FLAG_COOP_EDGE_ALWAYS = 32768
HGT_LINEAR_FORCE_XS = 0x100
HGT_LINEAR_NO_XS = 0x200
HGT_LINEAR_NO_TILE = 0x400
HGT_LINEAR_TANH = 0x1000
HGT_FEATURE_LAB_KERNELS = 1


class HgtLayout(C.Structure):
    _fields_ = [("d_k", C.c_int32), ("dk_pad", C.c_int32), ("d_pad", C.c_int32), ("vec", C.c_int32), ("heads", C.c_int32)]


class HgtPlanSizes(C.Structure):
    _fields_ = [("plan_bytes", C.c_uint64), ("tmp_bytes", C.c_uint64), ("max_items", C.c_int64), ("n_bins", C.c_int64)]


class HgtPlanRows(C.Structure):
    _fields_ = [("rows_all", C.c_void_p), ("off_all", C.c_void_p), ("rows_q", C.c_void_p), ("off_q", C.c_void_p)]


class HgtConvArgs(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int64), ("n_edges", C.c_int64),
        ("in_dim", C.c_int32), ("out_dim", C.c_int32), ("n_types", C.c_int32), ("n_relations", C.c_int32),
        ("n_heads", C.c_int32),
        ("use_norm", C.c_int32), ("use_rte", C.c_int32), ("precision", C.c_int32), ("want_att", C.c_int32),
        ("n_q_rows", C.c_int64),
        ("x", C.c_void_p), ("node_type", C.c_void_p), ("plan", C.c_void_p),
        ("w_qkv", C.c_void_p), ("b_qkv", C.c_void_p), ("w_a", C.c_void_p), ("b_a", C.c_void_p),
        ("relation_att", C.c_void_p), ("relation_msg", C.c_void_p), ("relation_pri", C.c_void_p),
        ("skip", C.c_void_p), ("ln_w", C.c_void_p), ("ln_b", C.c_void_p),
        ("rte_emb", C.c_void_p), ("rte_w", C.c_void_p), ("rte_b", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_uint64),
        ("out", C.c_void_p), ("att_out", C.c_void_p),
        ("phase_events", C.c_void_p),
        ("update_mode", C.c_int32),
        ("mid_w", C.c_void_p), ("mid_b", C.c_void_p), ("out_w", C.c_void_p), ("out_b", C.c_void_p),
        ("out_ln_w", C.c_void_p), ("out_ln_b", C.c_void_p),
        ("stage", C.c_int32), ("proj_rows", C.c_void_p), ("proj_off", C.c_void_p), ("proj_n", C.c_int64),
        ("prepared", C.c_void_p), ("prepared_bytes", C.c_uint64), ("prepared_valid", C.c_int32),
        ("plan_no_hubs", C.c_int32),
        ("flags", C.c_int32),
        ("slice_index", C.c_int32),
        ("slice_count", C.c_int32),
        ("q_begin", C.c_int64), ("q_end", C.c_int64),
        ("item_begin", C.c_int32), ("item_end", C.c_int32),
        ("proj_c24", C.c_void_p), ("proj_c24_row0", C.c_int64),
    ]


ABI_VERSION = 7          # HGT_ABI_VERSION of include/hgt_hip.h this binding was written against

_i32, _i64, _u64, _vp = C.c_int32, C.c_int64, C.c_uint64, C.c_void_p

# name -> (restype, argtypes); every symbol include/hgt_hip.h declares
SIGNATURES = {
    "hgt_strerror": (C.c_char_p, [C.c_int]),
    "hgt_abi_version": (C.c_int, []),
    "hgt_build_features": (C.c_int, []),
    "hgt_layout_for": (C.c_int, [_i32, _i32, C.POINTER(HgtLayout)]),
    "hgt_plan_sizes_for": (C.c_int, [_i64, _i64, _i32, _i32, C.POINTER(HgtPlanSizes)]),
    "hgt_plan_constants": (C.c_int, [C.POINTER(_i32), C.POINTER(_i32)]),
    "hgt_plan_item_edges": (C.c_int, [_i64, C.POINTER(_i32)]),
    "hgt_plan_row_lists": (C.c_int, [_vp, _i64, _i64, _i32, _i32, C.POINTER(HgtPlanRows)]),
    "hgt_plan_tile_items_offset": (C.c_int, [_i64, _i64, _i32, _i32, C.POINTER(_u64), C.POINTER(_i64)]),
    "hgt_plan_build": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _u64, _vp, _u64, _vp]),
    "hgt_plan_from_sorted": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _u64, _vp, _u64, _vp]),
    "hgt_typed_linear": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp,
                                   _i32, _i32, _i32, _i32, _vp]),
    "hgt_split_weights_bytes": (C.c_int, [_i32, _i32, _i32, C.POINTER(_u64)]),
    "hgt_split_weights": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "hgt_typed_linear_bf16x3": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp,
                                          _i32, _i32, _i32, _vp]),
    "hgt_linear_update_bf16x3": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp,
                                           _i32, _vp, _vp]),
    "hgt_split_weights_f16": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "hgt_typed_linear_xs_schedule": (C.c_int, [_vp, _i32, _i64, _i32, _i32, _vp, _i64, C.POINTER(_i64)]),
    "hgt_typed_linear_f16x3": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp,
                                         _i32, _i32, _i32, _vp]),
    "hgt_linear_update_f16x3": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp,
                                          _i32, _vp, _vp]),
    "hgt_zero_rows": (C.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "hgt_relation_pack": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "hgt_edge_logits": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hgt_edge_logits_mfma": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "hgt_edge_logits_range": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp]),
    "hgt_edge_aggregate_update_range": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp,
                                                  _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i64, _i64, _i32, _i32]),
    "hgt_hub_workspace_bytes_ex": (C.c_int, [_i64, _i32, _i32, _i32, _i32, C.POINTER(_u64)]),
    "hgt_edge_aggregate_ex": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _vp, _i32, _vp]),
    "hgt_edge_logits_slice": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "hgt_edge_aggregate_slice": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp,
                                           _i32, _i32, _vp, _i32, _i32, _vp]),
    "hgt_edge_softmax": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp]),
    "hgt_edge_aggregate": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "hgt_edge_aggregate_f16x3": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "hgt_relation_frag_pack_f16": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "hgt_edge_aggregate_items_bytes": (C.c_int, [_i64, _i32, _i32, C.POINTER(_u64)]),
    "hgt_edge_spmm_items": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _u64, _vp]),
    "hgt_edge_aggregate_items_update": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _u64, _vp, _vp, _i32,
                                                  _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "hgt_edge_aggregate_items": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _vp, _u64, _vp]),
    "hgt_edge_single_pass_items": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _i32,
                                             _vp, _u64, _vp]),
    "hgt_plan_header_to_host": (C.c_int, [_vp, _vp, _vp]),
    "hgt_relation_frag_bytes": (C.c_int, [_i32, _i32, _i32, C.POINTER(_u64)]),
    "hgt_relation_frag_pack": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "hgt_hub_workspace_bytes": (C.c_int, [_i64, _i32, _i32, C.POINTER(_u64)]),
    "hgt_att_export": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "hgt_edge_aggregate_update": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp,
                                            _vp, _vp, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "hgt_edge_aggregate_update_f16x3": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp,
                                                  _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "hgt_edge_spmm": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "hgt_edge_softmax_bwd": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _vp]),
    "hgt_edge_gather_sorted": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "hgt_head_dot": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "hgt_relation_outer": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hgt_node_update_bwd_ex": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp,
                                         _vp]),
    "hgt_single_group_offsets": (C.c_int, [_vp, _i32, _vp, _vp]),
    "hgt_node_update_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "hgt_gelu_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "hgt_mul_inplace": (C.c_int, [_vp, _vp, _i64, _vp]),
    "hgt_typed_wgrad": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _i64, _vp]),
    "hgt_typed_wgrad_bf16x3": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp]),
    "hgt_typed_colsum": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _i64, _i32, _vp, _i64, _vp]),
    "hgt_log_softmax_rows": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "hgt_row_dot": (C.c_int, [_vp, _vp, _i64, _i32, C.c_float, _vp, _vp]),
    "hgt_node_update": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp]),
    "hgt_node_update_ex": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _i32, _vp, _vp]),
    "hgt_tanh_inplace": (C.c_int, [_vp, _i64, _vp]),
    "hgt_gather_rows_c24": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp]),
    "hgt_unpack_rows_c24": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp]),
    "hgt_gather_rows": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp]),
    "hgt_conv_workspace_bytes": (C.c_int, [_i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(_u64)]),
    "hgt_conv_workspace_bytes_ex": (C.c_int, [_i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(_u64)]),
    "hgt_conv_prepared_bytes": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(_u64)]),
    "hgt_conv_forward": (C.c_int, [C.POINTER(HgtConvArgs), _vp]),
}

_lib = None


def load():
    """Load libhgt_hip.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                "pyhgt_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C pyhgt_amd/csrc`; there is no CPU/PyTorch fallback for HGTConv" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        if lib.hgt_abi_version() != ABI_VERSION:
            raise RuntimeError("pyhgt_amd: ABI version mismatch in %s" % LIB_PATH)
        _lib = lib
    return _lib


def check(code, what):
    if code != 0:
        msg = load().hgt_strerror(code).decode()
        raise RuntimeError("pyhgt_amd: %s failed: %s (code %d)" % (what, msg, code))


def layout_for(d_out, n_heads):
    lay = HgtLayout()
    check(load().hgt_layout_for(d_out, n_heads, C.byref(lay)), "hgt_layout_for(d=%d, heads=%d)" % (d_out, n_heads))
    return lay
