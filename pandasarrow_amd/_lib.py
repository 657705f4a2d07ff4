"""ctypes binding of the C ABI declared in include/pdx/abi.h (libpdx_hip.so).

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is NO
fallback: if the library is missing or a call fails, an exception is raised -- the product path never
routes through the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PDX_LIB_PATH: load another build of the same library (tools/sanitize_cpu.sh: the host side under ASan / UBSan)
LIB_PATH = os.environ.get("PDX_LIB_PATH") or os.path.join(_HERE, "csrc", "libpdx_hip.so")

# enums (include/pdx/abi.h)
OK, INVALID, INDEX_ERROR, OOM, DEVICE, NOT_IMPLEMENTED = range(6)
INT64, FLOAT64, BOOL, UINT64, TIMESTAMP_NS = range(5)
ADD, SUB, MUL, DIV = range(4)
BIT_OR, BIT_AND, BIT_XOR, SHIFT_LEFT, SHIFT_RIGHT = range(4, 9)
EQ, NE, LT, LE, GT, GE = range(6)
AND, OR = range(2)
NEGATE, ABS, SIGN, SQRT, EXP, BIT_NOT = range(6)
SCALAR_NONE, SCALAR_RHS, SCALAR_LHS = range(3)  # pdx_scalar_side: which operand of pdx_binary / pdx_compare is broadcast
AGG_SUM, AGG_MEAN, AGG_MIN, AGG_MAX, AGG_COUNT = range(5)
AGG_VARIANCE, AGG_STDDEV, AGG_PRODUCT, AGG_FIRST, AGG_LAST = range(5, 10)  # group-by only (include/pdx/abi.h)
AGG_ALL, AGG_ANY, AGG_COUNT_DISTINCT = range(10, 13)  # group-by only: all / any need BOOL values
ORIGIN_EPOCH, ORIGIN_START_DAY, ORIGIN_START, ORIGIN_END, ORIGIN_END_DAY, ORIGIN_CUSTOM = range(6)
(UNIT_NANOSECOND, UNIT_MICROSECOND, UNIT_MILLISECOND, UNIT_SECOND, UNIT_MINUTE, UNIT_HOUR, UNIT_DAY, UNIT_WEEK, UNIT_MONTH,
 UNIT_QUARTER) = range(10)  # pdx_calendar_unit
ORIGIN_SHARD = 0x100  # OR-ed into the origin type for a row-range shard of a longer axis (include/pdx/abi.h)


class PdxColumn(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("reserved", C.c_int32), ("length", C.c_int64), ("offset", C.c_int64),
                ("null_count", C.c_int64), ("validity", C.c_void_p), ("values", C.c_void_p)]


class PdxMutColumn(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("reserved", C.c_int32), ("length", C.c_int64), ("null_count", C.c_int64),
                ("validity", C.c_void_p), ("values", C.c_void_p)]


class _ScalarValue(C.Union):
    _fields_ = [("i64", C.c_int64), ("u64", C.c_uint64), ("f64", C.c_double)]


class PdxScalar(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("is_valid", C.c_int32), ("v", _ScalarValue), ("count", C.c_int64)]


class PdxError(RuntimeError):
    """Mirror of the reference's std::runtime_error(status.ToString()) (src/core.h:181-194)."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


# every symbol include/pdx/abi.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_COL = C.POINTER(PdxColumn)
_MUT = C.POINTER(PdxMutColumn)
ABI_SYMBOLS = {
    "pdx_abi_version": (C.c_int, []),
    "pdx_build_info": (C.c_char_p, []),
    "pdx_init": (C.c_int, [C.c_int]),
    "pdx_shutdown": (C.c_int, []),
    "pdx_last_error": (C.c_char_p, []),
    "pdx_malloc": (C.c_int, [C.POINTER(_P), C.c_size_t]),
    "pdx_free": (C.c_int, [_P]),
    "pdx_to_device": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "pdx_to_host": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "pdx_stream_synchronize": (C.c_int, [_P]),
    "pdx_trim_pool": (C.c_int, []),
    "pdx_profile_enable": (C.c_int, [C.c_int]),
    "pdx_profile_reset": (C.c_int, []),
    "pdx_profile_report": (C.c_int, [C.c_char_p, C.c_size_t]),
    "pdx_synth_keys": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, _P, _P]),
    "pdx_synth_vals": (C.c_int, [C.c_int64, C.c_int64, C.c_uint64, _P, _P]),
    "pdx_synth_ts": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P, _P]),
    "pdx_binary": (C.c_int, [C.c_int, _COL, _COL, C.c_int, _MUT, _P]),
    "pdx_compare": (C.c_int, [C.c_int, _COL, _COL, C.c_int, _MUT, _P]),
    "pdx_logical": (C.c_int, [C.c_int, _COL, _COL, _MUT, _P]),
    "pdx_invert": (C.c_int, [_COL, _MUT, _P]),
    "pdx_if_else": (C.c_int, [_COL, _COL, _COL, C.c_int, _MUT, _P]),
    "pdx_unary": (C.c_int, [C.c_int, _COL, _MUT, _P]),
    "pdx_cast_f64": (C.c_int, [_COL, C.c_int, _MUT, _P]),
    "pdx_power": (C.c_int, [_COL, C.c_double, _MUT, _P]),
    "pdx_aggregate": (C.c_int, [C.c_int, _COL, C.POINTER(PdxScalar), _P]),
    "pdx_filter_count": (C.c_int, [_COL, C.c_int, C.POINTER(C.c_int64), _P]),
    "pdx_filter": (C.c_int, [_COL, C.c_int, _COL, C.c_int, _MUT, _P]),
    "pdx_take": (C.c_int, [_COL, C.c_int, _COL, _MUT, _P]),
    "pdx_scatter": (C.c_int, [_COL, C.c_int, _COL, _MUT, _P]),
    "pdx_groupby_create": (C.c_int, [_COL, _P, C.POINTER(_P)]),
    "pdx_groupby_destroy": (C.c_int, [_P]),
    "pdx_groupby_num_groups": (C.c_int64, [_P]),
    "pdx_groupby_num_rows": (C.c_int64, [_P]),
    "pdx_groupby_unique_keys": (C.c_int, [_P, _MUT, _P]),
    "pdx_groupby_group_ids": (C.c_int, [_P, _P, _P]),
    "pdx_groupby_map_ids": (C.c_int, [_P, _P, _P, _P]),
    "pdx_groupby_first_rows": (C.c_int, [_P, _P, _P]),
    "pdx_groupby_groupings": (C.c_int, [_P, _P, _P, _P]),
    "pdx_groupby_agg": (C.c_int, [_P, _COL, C.POINTER(C.c_int), C.c_int, _MUT, _P]),
    "pdx_groupby_bind": (C.c_int, [_P, _COL, _P]),
    "pdx_groupby_unbind": (C.c_int, [_P, _COL]),
    "pdx_groupby_bind_limit": (C.c_int, [_P, C.c_size_t]),
    "pdx_groupby_bound_bytes": (C.c_int64, [_P]),
    "pdx_groupby_last_plan": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "pdx_groupby_group_values": (C.c_int, [_P, _COL, _P, C.POINTER(_P)]),
    "pdx_grouped_destroy": (C.c_int, [_P]),
    "pdx_grouped_counts": (C.c_int, [_P, _P, _P]),
    "pdx_grouped_partial_plan": (C.c_int, [_P, _P, _P, C.POINTER(C.c_int64), _P]),
    "pdx_grouped_partial_fill": (C.c_int, [_P, _P, _P, _P, _P]),
    "pdx_grouped_record_cuts": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P, _P]),
    "pdx_replay_partials": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int64, _P, _P]),
    "pdx_resample_create": (C.c_int, [_COL, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, _P, C.POINTER(_P)]),
    "pdx_resample_grid": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pdx_resample_row_labels": (C.c_int, [_P, _P, _P]),
    "pdx_round_temporal": (C.c_int, [C.c_int, _COL, C.c_int64, C.c_int, C.c_int, C.c_int, _MUT, _P]),
    "pdx_downsample_create": (C.c_int, [_COL, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, _P, C.POINTER(_P)]),
    "pdx_concat": (C.c_int, [_COL, C.c_int, _MUT, _P]),
    "pdx_ipc_open": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "pdx_ipc_destroy": (C.c_int, [_P]),
    "pdx_ipc_num_columns": (C.c_int, [_P]),
    "pdx_ipc_num_rows": (C.c_int64, [_P]),
    "pdx_ipc_column_name": (C.c_char_p, [_P, C.c_int]),
    "pdx_ipc_num_metadata": (C.c_int, [_P]),
    "pdx_ipc_metadata_key": (C.c_char_p, [_P, C.c_int]),
    "pdx_ipc_metadata_value": (C.c_char_p, [_P, C.c_int]),
    "pdx_ipc_load": (C.c_int, [_P, _P]),
    "pdx_ipc_column": (C.c_int, [_P, C.c_int, _COL]),
    "pdx_ipc_write": (C.c_int, [_COL, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int, _P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "pdx_ipc_free_blob": (C.c_int, [_P]),
    "pdx_dist_unique_id": (C.c_int, [_P]),
    "pdx_dist_init": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    "pdx_dist_init_custom": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    "pdx_dist_destroy": (C.c_int, [_P]),
    "pdx_dist_world": (C.c_int, [_P]),
    "pdx_dist_rank": (C.c_int, [_P]),
    "pdx_dist_groupby_sum_mean_count": (C.c_int, [_P, _COL, _COL, C.c_int64, _P, C.POINTER(_P)]),
    "pdx_dist_groupby_num_groups": (C.c_int64, [_P]),
    "pdx_dist_groupby_num_records": (C.c_int64, [_P]),
    "pdx_dist_groupby_fetch": (C.c_int, [_P, _MUT, _P, _P, _P, _P, _P]),
    "pdx_dist_groupby_destroy": (C.c_int, [_P]),
    "pdx_dist_concat": (C.c_int, [_P, _COL, _MUT, _P]),
    "pdx_dist_groupby_order_free": (C.c_int, [_P, _COL, _COL, C.POINTER(C.c_int), C.c_int, C.c_int64, _P, C.POINTER(_P)]),
    "pdx_dist_agg_num_groups": (C.c_int64, [_P]),
    "pdx_dist_agg_fetch": (C.c_int, [_P, _MUT, _P, _MUT, _P]),
    "pdx_dist_agg_destroy": (C.c_int, [_P]),
    "pdx_groupby_order_free_chunked": (C.c_int, [_COL, _COL, C.POINTER(C.c_int), C.c_int, C.c_int64, _P, C.POINTER(_P)]),
    "pdx_dist_resample": (C.c_int, [_P, _COL, _COL, C.POINTER(C.c_int), C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, _P, C.POINTER(_P)]),
    "pdx_dist_resampled_num_bins": (C.c_int64, [_P]),
    "pdx_dist_resampled_fetch": (C.c_int, [_P, _MUT, _MUT, _P]),
    "pdx_dist_resampled_destroy": (C.c_int, [_P]),
    "pdx_groupby_sum_mean_count_chunked": (C.c_int, [_COL, _COL, C.c_int64, _P, C.POINTER(_P)]),
    "pdx_parquet_open": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "pdx_parquet_destroy": (C.c_int, [_P]),
    "pdx_parquet_num_columns": (C.c_int, [_P]),
    "pdx_parquet_num_rows": (C.c_int64, [_P]),
    "pdx_parquet_column_name": (C.c_char_p, [_P, C.c_int]),
    "pdx_parquet_num_metadata": (C.c_int, [_P]),
    "pdx_parquet_metadata_key": (C.c_char_p, [_P, C.c_int]),
    "pdx_parquet_metadata_value": (C.c_char_p, [_P, C.c_int]),
    "pdx_parquet_load": (C.c_int, [_P, _P]),
    "pdx_parquet_column": (C.c_int, [_P, C.c_int, _COL]),
    "pdx_parquet_write": (C.c_int, [_COL, C.POINTER(C.c_char_p), C.c_int, C.c_int, _P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "pdx_parquet_free_blob": (C.c_int, [_P]),
    "pdx_index_union": (C.c_int, [_COL, _COL, C.c_int, _MUT, _P]),
    "pdx_index_intersection": (C.c_int, [_COL, _COL, _MUT, _P]),
    "pdx_argsort": (C.c_int, [_COL, C.c_int, _MUT, _P]),
    "pdx_reindex_indices": (C.c_int, [_COL, _COL, _MUT, _P]),
}

_lib = None


def load():
    """Load libpdx_hip.so and bind every ABI symbol.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and owns the device memory this binding hands to the
    # library, so torch must be loaded FIRST -- libpdx_hip.so's NEEDED libamdhip64.so.7 then resolves to the already-loaded copy.
    # (Loaded the other way round, the ROCm and the torch runtimes both end up in the process and the second one to initialise
    # reports "no ROCm-capable device".)  Pure C/C++ hosts link the system runtime as usual.
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise PdxError(DEVICE, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in ABI_SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != OK:
        msg = load().pdx_last_error().decode("utf-8", "replace")
        raise PdxError(status, msg)
