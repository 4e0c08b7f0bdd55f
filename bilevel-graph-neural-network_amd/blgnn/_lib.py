"""ctypes binding of libblgnn.so (include/blgnn.h).  No torch types cross the ABI:
device pointers travel as integers, sizes as C ints, the stream as a void*.

The library is the product.  If it is missing, or if a compute entry point is
called without a gfx950 device, this module raises -- there is no CPU fallback.
"""

import ctypes
import os

# A process must hold ONE HIP runtime.  PyTorch-ROCm bundles its own libamdhip64; if libblgnn
# were dlopen()ed first it would bind the system copy and device pointers / streams handed over
# by torch would be foreign to it (every call then fails with "invalid device pointer").
# Importing torch first makes the loader resolve libblgnn's libamdhip64 to the one torch holds.
try:
    import torch  # noqa: F401
except ImportError:      # a host without torch (plain C / ctypes caller) uses the system runtime
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libblgnn.so")

OK, ERR_ARG, ERR_DEVICE, ERR_LAUNCH, ERR_UNSUPPORTED = 0, -1, -2, -3, -4
ACT_IDS = {"identity": 0, "relu": 1, "prelu": 2, "sigmoid": 3, "tanh": 4}
REDUCE_SUM, REDUCE_MEAN = 0, 1
GAT_GROUP = {"src": 0, "dst": 1}

_p, _i32, _i64, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float

# name -> argtypes, exactly as declared in include/blgnn.h
SIGNATURES = {
    "blgnn_csr_from_coo": [_p, _p, _i64, _i32, _p, _p, _p, _p],
    "blgnn_segments_from_batch": [_p, _i64, _i32, _p, _p],
    "blgnn_spmm_sum": [_p, _p, _i64, _p, _p, _i64, _p, _i64, _i32, _i32, _f32, _i32, _p],
    "blgnn_spmm_sum_sp": [_p, _p, _i64, _p, _i64, _i32, _i32, _f32, _i32, _p, _i64, _p, _i64, _p],
    "blgnn_gin_layer": [_p, _p, _i64, _p, _i64, _i32, _i32, _p, _p, _p, _p, _i32, _f32, _i32, _p, _p, _p,
                        _p, _i64, _p],
    "blgnn_pack_weight_x6": [_p, _i64, _i32, _i32, _p, _p],
    "blgnn_pack_weight_h3": [_p, _i64, _i32, _i32, _p, _p],
    "blgnn_gin_layer_h3": [_p, _p, _i64, _p, _i64, _i32, _i32, _p, _p, _p, _p, _i32, _f32, _i32, _p, _p, _p,
                           _p, _i64, _p, _p, _p, _i64, _i32, _p, _p, _i64, _p, _i64, _p],
    "blgnn_gin_layer_h3_codes": [_p, _p, _i64, _p, _p, _i32, _i32, _p, _p, _p, _p, _i32, _f32, _i32, _p, _p, _p,
                                 _p, _i64, _p, _p, _p, _i64, _i32, _p, _p],
    "blgnn_gin_layer_x6": [_p, _p, _i64, _p, _i64, _i32, _i32, _p, _p, _p, _p, _i32, _f32, _i32, _p, _p, _p,
                           _p, _i64, _p],
    "blgnn_gin_stack": [_p, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _p, _p,
                        _p, _p, _p, _i32, _p, _i64, _p, _p, _i64, _p],
    "blgnn_gin_stack_h3": [_p, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _p, _p,
                           _p, _p, _p, _i32, _p, _i64, _p, _p, _i64, _p],
    "blgnn_gcn_deg_inv_sqrt": [_p, _p, _p, _i32, _p, _p],
    "blgnn_gcn_aggregate": [_p, _p, _i64, _p, _p, _p, _i64, _p, _i64, _i32, _i32, _p, _i32, _p, _p, _p, _i32, _p],
    "blgnn_gat_scores": [_p, _i64, _p, _i32, _i32, _p, _p, _p],
    "blgnn_gat_softmax_stats": [_p, _p, _i64, _p, _p, _i32, _i32, _f32, _p, _p, _p, _p],
    "blgnn_gat_aggregate": [_p, _p, _i64, _p, _i64, _p, _p, _p, _p, _i32, _f32, _p, _i64, _i32, _i32,
                            _p, _i32, _p, _p, _p, _i32, _p, _i32, _p, _p],
    "blgnn_edge_relu_sum": [_p, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i32, _i32, _p, _p, _i64, _p, _i64, _p],
    "blgnn_row_degree": [_p, _i32, _p, _p],
    "blgnn_segment_reduce": [_p, _i32, _p, _i64, _i32, _i32, _p, _i64, _p, _p],
    "blgnn_segment_reduce_multi": [_p, _i32, _i32, _p, _p, _p, _i32, _p, _i64, _p, _p],
    "blgnn_linear": [_p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _i32, _p, _p, _p, _p],
    "blgnn_gat_transform": [_p, _i64, _p, _i64, _p, _i32, _i32, _i32, _p, _i64, _p, _p, _p],
    "blgnn_linear_x6": [_p, _i64, _p, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _i32, _p, _p, _p, _p],
    "blgnn_pack_weight_h3g": [_p, _i64, _i32, _i32, _p, _p],
    "blgnn_row_absmax": [_p, _i64, _i32, _i32, _p, _p],
    "blgnn_linear_h3": [_p, _i64, _p, _p, _p, _i64, _p, _i32, _i32, _i32, _p, _p, _p, _i32, _p, _p, _p, _p],
    "blgnn_pack_weight_sp": [_p, _i64, _i32, _i32, _p, _p],
    "blgnn_split_rows_sp": [_p, _i64, _i32, _i32, _p, _i64, _p, _i64, _p],
    "blgnn_join_rows_sp": [_p, _i64, _p, _i64, _i32, _i32, _p, _i64, _p],
    "blgnn_linear_sp": [_p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _i32, _p,
                        _p, _p, _p, _i64, _p, _i32, _p, _p, _p, _i64, _i32, _p],
    "blgnn_hadamard": [_p, _i64, _p, _i64, _i32, _i32, _p, _i64, _p],
    "blgnn_gru_gates": [_p, _i64, _p, _i64, _p, _i64, _i32, _i32, _p, _i64, _p],
    "blgnn_gru_gates_backward": [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _p],
    "blgnn_gmn_match": [_p, _i32, _p, _i64, _p, _i64, _i32, _p, _i32, _i32, _p, _p, _p],
    "blgnn_gmn_match_sp": [_p, _i32, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i32, _p, _p, _i32, _p],
    "blgnn_split_rows_sp_pairs": [_p, _i64, _p, _p, _i32, _i32, _p, _i64, _p, _i64, _p],
    "blgnn_pair_gather_normalize": [_p, _i64, _p, _p, _i32, _i32, _f32, _p, _i64, _p],
    "blgnn_link_pred_head": [_p, _i64, _p, _p, _i32, _i32, _f32, _i32, _p, _p, _p, _i32, _i32, _p, _i64, _p],
    "blgnn_l2_normalize_rows": [_p, _i64, _p, _i64, _i32, _i32, _f32, _p],
    "blgnn_scatter_rows": [_p, _i64, _p, _p, _i64, _i32, _i32, _p],
    "blgnn_expand_rows": [_p, _i64, _p, _i64, _p, _i64, _i64, _p],
    "blgnn_stream_create_cu_mask": [_p, _i32, _p],
    "blgnn_stream_destroy": [_p],
    "blgnn_col_reduce": [_p, _i64, _p, _i64, _p, _p, _i32, _i32, _i32, _f32, _p, _p, _p],
    "blgnn_grad_weight": [_p, _i64, _p, _i64, _i32, _i32, _i32, _p, _p, _p],
    "blgnn_bn_finalize": [_p, _p, _i32, _f32, _f32, _f32, _p, _p, _p, _p],
    "blgnn_bn_apply": [_p, _i64, _p, _p, _p, _p, _i32, _i32, _i32, _p, _i64, _p],
    "blgnn_bn_backward": [_p, _i64, _p, _i64, _p, _p, _p, _p, _p, _i32, _i32, _i64, _p, _i64, _p],
    "blgnn_act_backward": [_p, _i64, _p, _i64, _i32, _p, _i32, _i32, _p, _i64, _p],
    "blgnn_act_forward": [_p, _i64, _i32, _p, _i32, _i32, _p, _i64, _p],
    "blgnn_transpose": [_p, _i64, _i32, _i32, _p, _i64, _p],
    "blgnn_segment_broadcast": [_p, _i32, _p, _i64, _p, _i32, _i32, _p, _i64, _i32, _p],
    "blgnn_l2_normalize_backward": [_p, _i64, _p, _i64, _i32, _i32, _f32, _p, _i64, _p],
    "blgnn_pair_scatter_add": [_p, _i64, _i32, _p, _i32, _i32, _p, _i64, _p],
    "blgnn_gat_backward_scores": [_p, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _i32, _i32, _i32, _f32, _p, _p, _p],
    "blgnn_edge_relu_sum_backward": [_p, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i32, _i32, _p, _i64, _p, _i64, _p],
    "blgnn_gmn_match_backward": [_p, _i32, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _i64, _i32, _i64, _p],
    "blgnn_gat_backward_features": [_p, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _f32, _p, _i64, _p],
}
LIBRARY_SYMBOLS = ["blgnn_version", "blgnn_last_error", "blgnn_device_count", "blgnn_packed_weight_x6_elems",
                   "blgnn_packed_weight_h3_elems", "blgnn_packed_weight_h3g_elems", "blgnn_packed_weight_sp_elems",
                   "blgnn_col_reduce_scratch_elems", "blgnn_grad_weight_scratch_elems"]

_lib = None


class BlgnnError(RuntimeError):
    """Launch failure / missing device / unsupported shape reported by libblgnn."""


def load():
    """Load libblgnn.so once.  Raises ImportError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libblgnn.so not found at %s -- build it with "
            "bilevel-graph-neural-network_amd/csrc/build.sh (or __graft_entry__.build()); "
            "there is no CPU fallback for the hot path" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    lib.blgnn_version.restype = ctypes.c_int
    lib.blgnn_last_error.restype = ctypes.c_char_p
    lib.blgnn_device_count.restype = ctypes.c_int
    lib.blgnn_packed_weight_x6_elems.argtypes = [_i32, _i32]
    lib.blgnn_packed_weight_x6_elems.restype = ctypes.c_int64
    lib.blgnn_packed_weight_h3_elems.argtypes = [_i32, _i32]
    lib.blgnn_packed_weight_h3_elems.restype = ctypes.c_int64
    lib.blgnn_packed_weight_h3g_elems.argtypes = [_i32, _i32]
    lib.blgnn_packed_weight_h3g_elems.restype = ctypes.c_int64
    lib.blgnn_packed_weight_sp_elems.argtypes = [_i32, _i32]
    lib.blgnn_packed_weight_sp_elems.restype = ctypes.c_int64
    lib.blgnn_col_reduce_scratch_elems.argtypes = [_i32, _i32]
    lib.blgnn_col_reduce_scratch_elems.restype = ctypes.c_int64
    lib.blgnn_grad_weight_scratch_elems.argtypes = [_i32, _i32, _i32]
    lib.blgnn_grad_weight_scratch_elems.restype = ctypes.c_int64
    _lib = lib
    return lib


def last_error():
    return (load().blgnn_last_error() or b"").decode("utf-8", "replace")


def call(name, *args):
    """Invoke an entry point and translate its status code into the exception the
    reference raises in the same situation: bad arguments ->
    ValueError, unsupported shape -> NotImplementedError, anything else ->
    BlgnnError."""
    rc = getattr(load(), name)(*args)
    if rc == OK:
        return
    msg = "%s failed (%d): %s" % (name, rc, last_error())
    if rc == ERR_ARG:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise BlgnnError(msg)


def device_count():
    return int(load().blgnn_device_count())
