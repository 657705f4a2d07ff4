"""CPU ORACLE bindings -- TEST INFRASTRUCTURE ONLY.

ctypes/numpy wrapper over ``oracle/_build/libpdx_oracle.so`` (built from ``pdx_oracle.c``,
a plain-C restatement of the Arrow-CPU behaviour the reference forwards to; see the header
of ``pdx_oracle.h``).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package.  The product (``pandasarrow_amd``) never does.

Conventions: values are numpy arrays; ``valid`` is an optional numpy bool array (True =
valid) that is packed LSB-first into an Arrow validity bitmap before the C call; ``offset``
shifts both the values and the bitmap (Arrow slice semantics).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("PDX_ORACLE_SO") or os.path.join(_HERE, "_build", "libpdx_oracle.so")  # (override: tools/sanitize_cpu.sh)

ADD, SUB, MUL, DIV = 0, 1, 2, 3
BIT_OR, BIT_AND, BIT_XOR, SHIFT_LEFT, SHIFT_RIGHT = 4, 5, 6, 7, 8  # int64 operands only
EQ, NE, LT, LE, GT, GE = 0, 1, 2, 3, 4, 5
AND, OR = 0, 1
AGG_SUM, AGG_MEAN, AGG_MIN, AGG_MAX, AGG_COUNT = 0, 1, 2, 3, 4
AGG_VARIANCE, AGG_STDDEV, AGG_PRODUCT, AGG_FIRST, AGG_LAST = 5, 6, 7, 8, 9
OK, INVALID, INDEX_ERROR = 0, 1, 2
ORIGIN_EPOCH, ORIGIN_START_DAY, ORIGIN_START, ORIGIN_END, ORIGIN_END_DAY, ORIGIN_CUSTOM = range(6)


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    if os.environ.get("PDX_ORACLE_SO"):
        return _SO
    src = os.path.join(_HERE, "pdx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(
        os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "pdx_oracle.h"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_splitmix64.restype = C.c_uint64
        _lib.orc_splitmix64.argtypes = [C.c_uint64]
        for name in ("orc_count", "orc_filter_count", "orc_group_ids_i64", "orc_groupby_sum_mean_count",
                     "orc_resample_group_info"):
            getattr(_lib, name).restype = C.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i64(x):
    return C.c_int64(int(x))


def pack_bits(b, offset=0):
    """bool array -> LSB-first bitmap whose bit ``offset`` is b[0]."""
    if b is None:
        return None
    b = np.asarray(b, dtype=bool)
    if offset:
        b = np.concatenate([np.zeros(offset, dtype=bool), b])
    out = np.packbits(b, bitorder="little")
    return np.ascontiguousarray(np.concatenate([out, np.zeros(8, np.uint8)]))


def unpack_bits(bits, n, offset=0):
    return np.unpackbits(np.asarray(bits, dtype=np.uint8), bitorder="little")[offset:offset + n].astype(bool)


def _shift(a, offset):
    """values with ``offset`` junk elements in front (to exercise Arrow slice offsets)."""
    a = np.ascontiguousarray(a)
    if not offset:
        return a
    pad = np.full(offset, 77, dtype=a.dtype) if a.dtype.kind in "iu" else np.full(offset, 7.5, dtype=a.dtype)
    return np.ascontiguousarray(np.concatenate([pad, a]))


# ------------------------------------------------------------------ synthetic inputs
def splitmix64(x):
    return lib().orc_splitmix64(C.c_uint64(int(x) & 0xFFFFFFFFFFFFFFFF))


def synth_keys(start, n, num_keys):
    out = np.empty(n, np.int64)
    lib().orc_synth_keys(_i64(start), _i64(n), _i64(num_keys), _p(out))
    return out


def synth_vals(start, n, seed_off=0):
    out = np.empty(n, np.float64)
    lib().orc_synth_vals(_i64(start), _i64(n), C.c_uint64(seed_off), _p(out))
    return out


def synth_ts(start, n, t0_ns, step_ns):
    out = np.empty(n, np.int64)
    lib().orc_synth_ts(_i64(start), _i64(n), _i64(t0_ns), _i64(step_ns), _p(out))
    return out


# ------------------------------------------------------------------ aggregates
def _is_float(a):
    return np.asarray(a).dtype == np.float64


def agg(kind, v, valid=None, offset=0):
    """Whole-array aggregate.  Returns (value, count); value is None when Arrow returns null."""
    v = np.asarray(v)
    n = len(v)
    vs = _shift(v.astype(np.float64 if _is_float(v) else np.int64), offset)
    vb = pack_bits(valid, offset)
    cnt = C.c_int64(0)
    L = lib()
    if kind == AGG_COUNT:
        return int(L.orc_count(_p(vb), _i64(offset), _i64(n))), None
    if _is_float(v):
        if kind in (AGG_SUM, AGG_MEAN):
            out = C.c_double(0)
            (L.orc_sum_f64 if kind == AGG_SUM else L.orc_mean_f64)(_p(vs), _p(vb), _i64(offset), _i64(n), C.byref(out), C.byref(cnt))
            return (out.value if cnt.value else None), cnt.value
        mn, mx = C.c_double(0), C.c_double(0)
        L.orc_minmax_f64(_p(vs), _p(vb), _i64(offset), _i64(n), C.byref(mn), C.byref(mx), C.byref(cnt))
        return ((mn.value if kind == AGG_MIN else mx.value) if cnt.value else None), cnt.value
    if kind == AGG_SUM:
        out = C.c_int64(0)
        L.orc_sum_i64(_p(vs), _p(vb), _i64(offset), _i64(n), C.byref(out), C.byref(cnt))
        return (out.value if cnt.value else None), cnt.value
    if kind == AGG_MEAN:
        out = C.c_double(0)
        L.orc_mean_i64(_p(vs), _p(vb), _i64(offset), _i64(n), C.byref(out), C.byref(cnt))
        return (out.value if cnt.value else None), cnt.value
    mn, mx = C.c_int64(0), C.c_int64(0)
    L.orc_minmax_i64(_p(vs), _p(vb), _i64(offset), _i64(n), C.byref(mn), C.byref(mx), C.byref(cnt))
    return ((mn.value if kind == AGG_MIN else mx.value) if cnt.value else None), cnt.value


# ------------------------------------------------------------------ element-wise
def _binary_like(fn_f64, fn_i64, op, a, b, va, vb, offset, out_dtype, bits):
    sa, sb = np.ndim(a) == 0, np.ndim(b) == 0  # a python scalar on either side (never both)
    a = np.atleast_1d(np.asarray(a))
    b = np.atleast_1d(np.asarray(b))
    isf = a.dtype == np.float64 or b.dtype == np.float64  # implicit promotion int64 (+) double -> double
    dt = np.float64 if isf else np.int64
    n = len(b) if sa else len(a)
    if isf and a.dtype != b.dtype:
        # the promotion is Arrow's CHECKED cast: a valid int64 value outside +-2^53 fails the call (pinned live against Arrow C++ 25 by
        # tests/cpp/arrow_bridge_test.cpp and tests/test_oracle_golden_r4.py)
        iv, ivalid = (a, va) if a.dtype != np.float64 else (b, vb)
        sel = iv.astype(np.int64)
        ok = np.ones(len(sel), bool) if ivalid is None else np.asarray(ivalid, bool)[:len(sel)]
        bad = ok & ((sel > 2**53) | (sel < -2**53))
        if bad.any():
            raise OracleError(INVALID, f"Integer value {int(sel[np.flatnonzero(bad)[0]])} not in range: -9007199254740992 to 9007199254740992")
    A = a.astype(dt) if sa else _shift(a.astype(dt), offset)
    B = b.astype(dt) if sb else _shift(b.astype(dt), offset)
    VA = pack_bits(va, 0 if sa else offset)
    VB = pack_bits(vb, 0 if sb else offset)
    need_valid = va is not None or vb is not None
    out_valid = np.zeros((n + 7) // 8 + 8, np.uint8) if need_valid else None
    if bits:
        out = np.zeros((n + 7) // 8 + 8, np.uint8)
    else:
        out = np.empty(n, dt if out_dtype is None else out_dtype)
    rc = (fn_f64 if isf else fn_i64)(C.c_int(op), _p(A), _p(VA), _i64(0 if sa else offset), _p(B), _p(VB), _i64(0 if sb else offset),
                                     C.c_int(2 if sa else 1 if sb else 0), _i64(n), _p(out), _p(out_valid))
    if rc == INVALID:
        raise OracleError(INVALID, "divide by zero")
    res = unpack_bits(out, n) if bits else out
    return res, (unpack_bits(out_valid, n) if need_valid else None)


def binary(op, a, b, va=None, vb=None, offset=0):
    """a (op) b; a OR b may be a python scalar (Scalar lhs: src/scalar.cpp:24-36; a null scalar: pass va/vb = [False]).
    Returns (values, valid|None)."""
    L = lib()
    return _binary_like(L.orc_binary_f64, L.orc_binary_i64, op, a, b, va, vb, offset, None, False)


def compare(op, a, b, va=None, vb=None, offset=0):
    L = lib()
    return _binary_like(L.orc_compare_f64, L.orc_compare_i64, op, a, b, va, vb, offset, None, True)


def logical(op, a, b, va=None, vb=None, offset=0):
    n = len(a)
    out = np.zeros((n + 7) // 8 + 8, np.uint8)
    need_valid = va is not None or vb is not None
    out_valid = np.zeros((n + 7) // 8 + 8, np.uint8) if need_valid else None
    lib().orc_logical(C.c_int(op), _p(pack_bits(a, offset)), _p(pack_bits(va, offset)), _i64(offset), _p(pack_bits(b, offset)),
                      _p(pack_bits(vb, offset)), _i64(offset), _i64(n), _p(out), _p(out_valid))
    return unpack_bits(out, n), (unpack_bits(out_valid, n) if need_valid else None)


def invert(a, offset=0):
    n = len(a)
    out = np.zeros((n + 7) // 8 + 8, np.uint8)
    lib().orc_invert(_p(pack_bits(a, offset)), _i64(offset), _i64(n), _p(out))
    return unpack_bits(out, n)


def if_else(cond, a, b, cond_valid=None, va=None, vb=None):
    """arrow::compute::IfElse(cond, a, b) (src/series.cpp:1203-1209, 1247-1253): a or b may be a python scalar (None = null scalar).
    int64 / float64 operands, mixed -> float64.  Returns (values, valid bool[n])."""
    cond = np.asarray(cond, bool)
    n = len(cond)

    def prep(x, vx):
        if x is None:
            return np.zeros(1, np.int64), np.zeros(1, bool), True
        if np.isscalar(x):
            return np.array([x]), (None if vx is None else np.asarray(vx, bool)), True
        return np.ascontiguousarray(x), (None if vx is None else np.asarray(vx, bool)), False

    a, va, sa = prep(a, va)
    b, vb, sb = prep(b, vb)
    isf = a.dtype == np.float64 or b.dtype == np.float64
    dt = np.float64 if isf else np.int64
    a, b = a.astype(dt), b.astype(dt)
    out = np.zeros(max(n, 1), np.uint64)
    ov = np.zeros((n + 7) // 8 + 8, np.uint8)
    lib().orc_if_else(_p(pack_bits(cond, 0)), _p(pack_bits(cond_valid, 0)), _p(a.view(np.uint64)), _p(pack_bits(va, 0)), C.c_int(sa), _p(b.view(np.uint64)),
                      _p(pack_bits(vb, 0)), C.c_int(sb), _i64(n), _p(out), _p(ov))
    return out[:n].view(dt), unpack_bits(ov, n)


UNARY_NEGATE, UNARY_ABS, UNARY_SIGN, UNARY_SQRT, UNARY_EXP, UNARY_BIT_NOT, UNARY_POWER = 0, 1, 2, 3, 4, 5, 100


def unary(op, a, valid=None, expo=0.0):
    """negate / abs / sign / sqrt / exp / bit_wise_not / power(a, expo) of one array (src/dataframe.cpp:251-275, 919-935).
    Returns the result array (float64 for sqrt / exp / power, int64 for the sign of integers, else the input type); validity
    passes through.  Raises ValueError with Arrow's message when an integer cannot be cast to float64 exactly."""
    a = np.ascontiguousarray(a)
    dt = {"int64": 0, "uint64": 1, "float64": 2}[a.dtype.name]
    out = np.zeros(max(len(a), 1), np.uint64)
    bad = np.zeros(1, np.uint64)
    rc = lib().orc_unary(C.c_int(op), C.c_int(dt), _p(a.view(np.uint64)), _p(pack_bits(valid, 0)), _i64(0), _i64(len(a)), C.c_double(expo), _p(out), _p(bad))
    if rc == 1:
        v = int(bad[0]) if dt == 1 else int(bad.view(np.int64)[0])
        raise ValueError(f"Integer value {v} not in range: {'0' if dt == 1 else '-9007199254740992'} to 9007199254740992")
    if rc != 0:
        raise TypeError("no kernel matching input types")
    out = out[: len(a)]
    if op in (UNARY_SQRT, UNARY_EXP, UNARY_POWER) or dt == 2:
        return out.view(np.float64)
    return out.view(np.int64) if (dt == 0 or op == UNARY_SIGN) else out


# ------------------------------------------------------------------ filter / take
def _as_u64(v):
    v = np.ascontiguousarray(v)
    assert v.dtype.itemsize == 8
    return v.view(np.uint64)


def filter(v, mask, valid=None, mask_valid=None, emit_null=True, offset=0):
    """Returns (values, valid|None)."""
    v = np.ascontiguousarray(v)
    n = len(v)
    L = lib()
    mb, mv = pack_bits(mask, offset), pack_bits(mask_valid, offset)
    m = int(L.orc_filter_count(_p(mb), _p(mv), _i64(offset), _i64(n), C.c_int(emit_null)))
    out = np.zeros(m, np.uint64)
    ov = np.zeros((m + 7) // 8 + 8, np.uint8)
    nulls = C.c_int64(0)
    L.orc_filter_64(_p(_shift(_as_u64(v), offset)), _p(pack_bits(valid, offset)), _i64(offset), _p(mb), _p(mv), _i64(offset),
                    _i64(n), C.c_int(emit_null), _p(out), _p(ov), C.byref(nulls))
    return out.view(v.dtype), (unpack_bits(ov, m) if (valid is not None or mask_valid is not None) else None)


def take(v, idx, valid=None, idx_valid=None, offset=0):
    v = np.ascontiguousarray(v)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    n, m = len(v), len(idx)
    out = np.zeros(m, np.uint64)
    ov = np.zeros((m + 7) // 8 + 8, np.uint8)
    nulls, bad = C.c_int64(0), C.c_int64(0)
    rc = lib().orc_take_64(_p(_shift(_as_u64(v), offset)), _p(pack_bits(valid, offset)), _i64(offset), _i64(n), _p(idx),
                           _p(pack_bits(idx_valid)), _i64(0), _i64(m), _p(out), _p(ov), C.byref(nulls), C.byref(bad))
    if rc == INDEX_ERROR:
        raise OracleError(INDEX_ERROR, f"Index {bad.value} out of bounds")
    return out.view(v.dtype), (unpack_bits(ov, m) if (valid is not None or idx_valid is not None) else None)


# ------------------------------------------------------------------ group-by
def group_ids(keys, valid=None, offset=0):
    """Returns (ids uint32[n], uniques int64[G], unique_is_null bool[G], first_row int64[G])."""
    keys = np.asarray(keys, dtype=np.int64)
    n = len(keys)
    ids = np.zeros(n, np.uint32)
    uniq = np.zeros(max(n, 1), np.int64)
    isnull = np.zeros(max(n, 1), np.uint8)
    first = np.zeros(max(n, 1), np.int64)
    G = int(lib().orc_group_ids_i64(_p(_shift(keys, offset)), _p(pack_bits(valid, offset)), _i64(offset), _i64(n), _p(ids), _p(uniq),
                                    _p(isnull), _p(first)))
    return ids, uniq[:G].copy(), isnull[:G].astype(bool), first[:G].copy()


def groupings(ids, G):
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    offsets = np.zeros(G + 1, np.int64)
    rows = np.zeros(max(len(ids), 1), np.int64)
    lib().orc_make_groupings(_p(ids), _i64(len(ids)), _i64(G), _p(offsets), _p(rows))
    return offsets, rows[:len(ids)]


def groupby_agg(kind, ids, G, v, valid=None, offset=0, nthreads=1):
    """Per-group aggregate in group-id order.  Returns (values, valid bool[G])."""
    v = np.asarray(v)
    offsets, rows = groupings(ids, G)
    out_f = np.zeros(max(G, 1), np.float64)
    out_i = np.zeros(max(G, 1), np.int64)
    ov = np.zeros(max(G, 1), np.uint8)
    isf = v.dtype == np.float64
    fn = lib().orc_groupby_agg_f64 if isf else lib().orc_groupby_agg_i64
    vs = _shift(v.astype(np.float64 if isf else np.int64), offset)
    fn(C.c_int(kind), _p(offsets), _p(rows), _i64(G), _p(vs), _p(pack_bits(valid, offset)), _i64(offset), _p(out_f), _p(out_i), _p(ov),
       C.c_int(nthreads))
    if kind == AGG_COUNT or (not isf and kind not in (AGG_MEAN, AGG_VARIANCE, AGG_STDDEV)):
        return out_i[:G], ov[:G].astype(bool)
    return out_f[:G], ov[:G].astype(bool)


def groupby_sum_mean_count(keys, vals, nthreads=1):
    keys = np.ascontiguousarray(keys, dtype=np.int64)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    n = len(keys)
    ok = np.zeros(max(n, 1), np.int64)
    os_ = np.zeros(max(n, 1), np.float64)
    om = np.zeros(max(n, 1), np.float64)
    oc = np.zeros(max(n, 1), np.int64)
    G = int(lib().orc_groupby_sum_mean_count(_p(keys), _p(vals), _i64(n), _p(ok), _p(os_), _p(om), _p(oc), C.c_int(nthreads)))
    return ok[:G], os_[:G], om[:G], oc[:G]


# ------------------------------------------------------------------ resample
def resample_group_info(ts, freq_ns, closed_right=False, label_right=False, origin=ORIGIN_START_DAY, origin_custom_ns=0,
                        offset_ns=0):
    """Returns (bins int64[nb], labels int64[nb])."""
    ts = np.ascontiguousarray(ts, dtype=np.int64)
    n = len(ts)
    if n == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    cap = int((int(ts.max()) - int(ts.min())) // freq_ns + 4)
    bins = np.zeros(cap, np.int64)
    labels = np.zeros(cap, np.int64)
    nb = int(lib().orc_resample_group_info(_p(ts), _i64(n), _i64(freq_ns), C.c_int(closed_right), C.c_int(label_right), C.c_int(origin),
                                           _i64(origin_custom_ns), _i64(offset_ns), _p(bins), _p(labels), _i64(cap)))
    if nb < 0:
        raise OracleError(INVALID, {-1: "Invalid length for values or for binner", -2: "Values falls before first bin",
                                    -3: "Values falls after last bin", -4: "cap", -5: "start date has to be less than end date"}[nb])
    return bins[:nb].copy(), labels[:nb].copy()


def resample_row_labels(ts, freq_ns, shard=False, **kw):
    """shard=True: `ts` is one row-range shard of a longer axis whose rows < bins test was done by the caller (multi-rank tests)."""
    bins, labels = resample_group_info(ts, freq_ns, **kw)
    out = np.zeros(len(ts), np.int64)
    if len(bins):
        if bins[-1] < len(labels) and not shard:
            raise OracleError(INVALID, "upSampling is not implemented.")
        lib().orc_resample_expand(_p(bins), _p(labels), _i64(len(bins)), _p(out))
    return out


def resample_agg(kind, ts, v, freq_ns, valid=None, **kw):
    """Resampler::<agg> (src/group_by.h:249-299): group on per-row labels; empty bins vanish.
    Returns (labels int64[G], values, valid bool[G])."""
    row_labels = resample_row_labels(ts, freq_ns, **kw)
    ids, uniq, _, _ = group_ids(row_labels)
    vals, ok = groupby_agg(kind, ids, len(uniq), v, valid)
    return uniq, vals, ok


# ------------------------------------------------------------------ concat
def concat(parts, valids=None):
    """Row-concat of 8-byte columns.  Returns (values, valid|None)."""
    parts = [np.ascontiguousarray(p) for p in parts]
    k = len(parts)
    total = sum(len(p) for p in parts)
    dtype = parts[0].dtype if parts else np.dtype(np.float64)
    u = [_as_u64(p) for p in parts]
    vb = [None if (valids is None or valids[i] is None) else pack_bits(valids[i]) for i in range(k)]
    pp = (C.c_void_p * max(k, 1))(*[x.ctypes.data for x in u])
    pv = (C.c_void_p * max(k, 1))(*[(None if b is None else b.ctypes.data) for b in vb])
    offs = np.zeros(max(k, 1), np.int64)
    lens = np.array([len(p) for p in parts] + ([0] if not k else []), np.int64)
    out = np.zeros(total, np.uint64)
    ov = np.zeros((total + 7) // 8 + 8, np.uint8)
    nulls = C.c_int64(0)
    lib().orc_concat_64(pp, pv, _p(offs), _p(lens), C.c_int(k), _p(out), _p(ov), C.byref(nulls))
    return out.view(dtype), (None if valids is None else unpack_bits(ov, total))


# ------------------------------------------------------------------ sort
def argsort(v, valid=None, ascending=True):
    """Series::argsort / the indices of Series::sort (src/series.cpp:864-868, 978-992): Arrow's array_sort_indices restated --
    stable in both orders (equal values keep their row order, -0.0 == 0.0), NaNs behind every number, nulls behind the NaNs, in
    BOTH orders.  Returns uint64 indices."""
    v = np.asarray(v)
    n = len(v)
    valid = np.ones(n, bool) if valid is None else np.asarray(valid, bool)
    isnan = (v != v) if v.dtype.kind == "f" else np.zeros(n, bool)
    cls = np.where(~valid, 2, np.where(isnan, 1, 0))
    num = np.flatnonzero(cls == 0)
    vals = v[num]
    if v.dtype.kind == "f":
        vals = vals + 0.0  # -0.0 -> 0.0 is not needed for ordering: numpy compares them equal and the sort is stable
    order = np.argsort(vals, kind="stable")
    if not ascending:
        # stable descending: sort ascending on the negated order of values = stable argsort of the reversed ranks; restate via keys
        # rank = position in the ascending stable order of DISTINCT values, then a stable argsort of (-rank)
        _, inv = np.unique(vals, return_inverse=True)
        order = np.argsort(-inv.astype(np.int64), kind="stable")
    out = np.concatenate([num[order], np.flatnonzero(cls == 1), np.flatnonzero(cls == 2)])
    return out.astype(np.uint64)


# ------------------------------------------------------------------ index alignment
def index_union(a, b, sort=True):
    """Series::broadcast's new index (src/series.cpp:212-227): Unique(Concatenate(a, b)) sorted ascending; sort=False is
    Series::union_ (src/series.cpp:782-798): the distinct labels in first-occurrence order."""
    cat = np.concatenate([np.asarray(a), np.asarray(b)])
    if sort:
        return np.unique(cat)
    _, first = np.unique(cat, return_index=True)
    return cat[np.sort(first)]


def index_intersection(a, b):
    """Series::intersection (src/series.cpp:763-780): for every distinct label of a that occurs in b, its (last) position in a;
    positions sorted ascending; Take."""
    a, b = np.asarray(a), np.asarray(b)
    last = {}
    for i, v in enumerate(a.tolist()):
        last[v] = i
    inb = set(b.tolist())
    pos = sorted(p for v, p in last.items() if v in inb)
    return a[np.array(pos, dtype=np.int64)] if pos else a[:0]


def reindex_indices(old_index, new_index):
    """Series::reindex (src/series.cpp:1255-1309): position of every new label in the old index -- insert_or_assign keeps the
    LAST position of a duplicated label -- and a bool mask of the labels that are present (absent -> AppendNull)."""
    old_index, new_index = np.asarray(old_index), np.asarray(new_index)
    if len(old_index) == 0:
        return np.zeros(len(new_index), np.int64), np.zeros(len(new_index), bool)
    order = np.argsort(old_index, kind="stable")
    so = old_index[order]
    j = np.searchsorted(so, new_index, side="right") - 1
    present = (j >= 0) & (so[np.maximum(j, 0)] == new_index)
    return np.where(present, order[np.maximum(j, 0)], 0).astype(np.int64), present


def reindex(values, valid, old_index, new_index, fill=None):
    """Series::reindex(newIndex, fillValue) (src/series.cpp:1255-1309): the value at the LAST position of every new label (a null value
    stays null); a label the old index lacks gives `fill` (AppendScalar(*fillValue)) or null.  Returns (values, valid)."""
    values = np.asarray(values)
    valid = np.ones(len(values), bool) if valid is None else np.asarray(valid, bool)
    pos, present = reindex_indices(old_index, new_index)
    if len(values) == 0:
        out, ok = np.zeros(len(pos), values.dtype), np.zeros(len(pos), bool)
    else:
        out, ok = values[pos].copy(), valid[pos] & present
    out[~ok] = 0
    if fill is not None:
        out[~present] = fill
        ok = ok | ~present
    return out, ok


# ------------------------------------------------------------------ temporal rounding / DataFrame::downsample
UNIT_NANOSECOND, UNIT_MICROSECOND, UNIT_MILLISECOND, UNIT_SECOND, UNIT_MINUTE, UNIT_HOUR, UNIT_DAY, UNIT_WEEK, UNIT_MONTH, UNIT_QUARTER = range(10)
UNIT_NAMES = ["nanosecond", "microsecond", "millisecond", "second", "minute", "hour", "day", "week", "month", "quarter"]
NS_PER_DAY = 86400 * 10**9


def round_temporal(ts, multiple, unit, ceil=False, week_starts_monday=True, calendar_based_origin=False, valid=None, offset=0):
    """arrow::compute::FloorTemporal / CeilTemporal on timestamp[ns] (no tz) with RoundTemporalOptions(multiple, unit,
    week_starts_monday, ceil_is_strictly_greater=false, calendar_based_origin) -- src/dataframe.cpp:1271-1276.
    Returns (int64 ns, valid|None)."""
    ts = np.asarray(ts, np.int64)
    n = len(ts)
    T = _shift(ts, offset)
    vb = pack_bits(valid, offset)
    out = np.empty(n, np.int64)
    ov = np.zeros((n + 7) // 8 + 8, np.uint8) if valid is not None else None
    rc = lib().orc_round_temporal(C.c_int(int(bool(ceil))), _p(T), _p(vb), _i64(offset), _i64(n), _i64(multiple), C.c_int(unit),
                                  C.c_int(int(bool(week_starts_monday))), C.c_int(int(bool(calendar_based_origin))), _p(out), _p(ov))
    if rc != OK:
        raise OracleError(rc, "round_temporal: bad multiple / unit")
    return out, (None if valid is None else unpack_bits(ov, n))


_DOWNSAMPLE_UNITS = {"n": UNIT_NANOSECOND, "u": UNIT_MICROSECOND, "m": UNIT_MILLISECOND, "S": UNIT_SECOND, "T": UNIT_MINUTE,
                     "H": UNIT_HOUR, "D": UNIT_DAY, "Q": UNIT_QUARTER, "W": UNIT_WEEK, "M": UNIT_MONTH}


def downsample_labels(ts, rule, closed_label_right=True, week_starts_monday=True, start_epoch=True):
    """The binned index of DataFrame::downsample (src/dataframe.cpp:1265-1290): splitTimeSpan(rule) (src/core.cpp:110-133),
    getCalendarUnit(first letter) (src/core.cpp:135-172), Ceil/FloorTemporal, and one day less for rules whose unit ends with
    "E" or is M / W / Y / Q (the label becomes the period's last day)."""
    k = 0
    while k < len(rule) and not rule[k].isalpha():
        k += 1
    mult, unit_s = (int(rule[:k]) if k else 1), rule[k:]
    if not unit_s or unit_s[0] not in _DOWNSAMPLE_UNITS:
        raise OracleError(INVALID, "invalid unit got " + unit_s[:1])
    out, _ = round_temporal(ts, mult, _DOWNSAMPLE_UNITS[unit_s[0]], closed_label_right, week_starts_monday, start_epoch)
    if unit_s.endswith("E") or unit_s in ("M", "W", "Y", "Q"):
        out = out - NS_PER_DAY
    return out


def downsample_agg(kind, ts, v, valid=None, rule="1T", **kw):
    """Resampler(DataFrame{values, binned index}) -> GroupBy on the binned labels (first-occurrence order, hash grouping: the
    index need not be sorted) -> per-group aggregate.  Returns (labels, values, ok)."""
    labels = downsample_labels(ts, rule, **kw)
    ids, uniq, _, _ = group_ids(labels)
    vals, ok = groupby_agg(kind, ids, len(uniq), v, valid)
    return uniq, vals, ok


# ------------------------------------------------------------------ Arrow C++ itself as the timed CPU baseline (bench.py cpu_baseline)
_ARROW_SEQ = os.path.join(_HERE, "_build", "arrow_seq")


def arrow_seq_build(force: bool = False):
    """g++ oracle/arrow_seq.cpp against the pyarrow wheel's headers + libarrow / libarrow_compute (third-party library the
    reference forwards to; nothing of the reference is compiled).  Returns the binary's path, or raises when the wheel (or its
    headers) is not on this box."""
    src = os.path.join(_HERE, "arrow_seq.cpp")
    if not force and os.path.exists(_ARROW_SEQ) and os.path.getmtime(_ARROW_SEQ) >= os.path.getmtime(src):
        return _ARROW_SEQ
    import glob

    import pyarrow as pa

    inc, libdirs = pa.get_include(), pa.get_library_dirs()
    libdir = next(d for d in libdirs if glob.glob(os.path.join(d, "libarrow.so*")))

    def so(name):  # the wheel ships versioned names only (libarrow.so.2500): link them by file name
        hits = sorted(glob.glob(os.path.join(libdir, name + ".so.*")) or glob.glob(os.path.join(libdir, name + ".so")))
        if not hits:
            raise FileNotFoundError(name)
        return "-l:" + os.path.basename(hits[0])

    os.makedirs(os.path.dirname(_ARROW_SEQ), exist_ok=True)
    cmd = ["g++", "-std=c++20", "-O2", "-pthread", f"-I{inc}", src, f"-L{libdir}", so("libarrow_compute"), so("libarrow"),
           f"-Wl,-rpath,{libdir}", "-o", _ARROW_SEQ]
    subprocess.check_call(cmd)
    return _ARROW_SEQ


def arrow_order_run(rows, nkeys, seed=1, timeout=600):
    """oracle/arrow_order.cpp: (keys int64[n], ids uint32[n]) -- the group ids Arrow C++'s Grouper::Consume hands out for a seeded key
    column (one Consume call over the whole column, as GroupBy::makeGroups does).  Raises when the pyarrow wheel is not on this box."""
    import glob
    import tempfile

    import pyarrow as pa

    src, exe = os.path.join(_HERE, "arrow_order.cpp"), os.path.join(_HERE, "_build", "arrow_order")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        inc = pa.get_include()
        libdir = next(d for d in pa.get_library_dirs() if glob.glob(os.path.join(d, "libarrow.so*")))

        def so(name):
            return "-l:" + os.path.basename(sorted(glob.glob(os.path.join(libdir, name + ".so.*")) or glob.glob(os.path.join(libdir, name + ".so")))[0])

        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++20", "-O2", f"-I{inc}", src, f"-L{libdir}", so("libarrow_compute"), so("libarrow"), f"-Wl,-rpath,{libdir}", "-o", exe])
    with tempfile.NamedTemporaryFile(suffix=".bin") as tf:
        r = subprocess.run([exe, "--rows", str(int(rows)), "--keys", str(int(nkeys)), "--seed", str(int(seed)), "--out", tf.name], capture_output=True, text=True,
                           timeout=timeout)
        if r.returncode != 0:
            raise RuntimeError("arrow_order failed: " + r.stderr[-500:])
        raw = np.fromfile(tf.name, dtype=np.uint8)
    n = int(rows)
    return raw[:n * 8].view(np.int64).copy(), raw[n * 8:n * 12].view(np.uint32).copy()


def arrow_seq_run(rows, nkeys, threads, timeout=1800):
    """Run the harness on the first `rows` rows of the synthetic workload; -> dict(seconds, phases, arrow_version, keys, sum, mean, count)."""
    import json
    import tempfile

    exe = arrow_seq_build()
    with tempfile.NamedTemporaryFile(suffix=".bin") as tf:
        r = subprocess.run([exe, "--rows", str(int(rows)), "--keys", str(int(nkeys)), "--threads", str(int(threads)), "--out", tf.name],
                           capture_output=True, text=True, timeout=timeout)
        if r.returncode != 0:
            raise RuntimeError("arrow_seq failed: " + r.stderr[-500:])
        info = json.loads(r.stdout.strip().splitlines()[-1])
        G = int(info["groups"])
        raw = np.fromfile(tf.name, dtype=np.uint8)
    a = raw.view(np.int64)
    info.update(keys=a[:G].copy(), sum=raw.view(np.float64)[G:2 * G].copy(), mean=raw.view(np.float64)[2 * G:3 * G].copy(), count=a[3 * G:4 * G].copy())
    return info


# ------------------------------------------------------------------ group-by all / any / count_distinct / min_max
def groupby_all_any(ids, G, b, valid=None, offset=0):
    """GROUPBY_NUMERIC_AGG(all | any, bool) (src/dataframe.cpp:1520-1522).  b: bool array.  -> (all bool[G], any bool[G], ok bool[G])."""
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    n = len(ids)
    oa, oy, ok = (np.zeros(max(G, 1), np.uint8) for _ in range(3))
    lib().orc_groupby_all_any(_p(ids), _i64(n), _i64(G), _p(pack_bits(np.asarray(b, bool), offset)), _p(pack_bits(valid, offset)), _i64(offset),
                              _p(oa), _p(oy), _p(ok))
    return oa[:G].astype(bool), oy[:G].astype(bool), ok[:G].astype(bool)


def groupby_count_distinct(ids, G, v, valid=None, offset=0):
    """GROUPBY_NUMERIC_AGG(count_distinct, int64_t) (src/dataframe.cpp:1526): distinct VALID values per group, by bit pattern."""
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    out = np.zeros(max(G, 1), np.int64)
    vs = _shift(_as_u64(np.ascontiguousarray(v)), offset)
    lib().orc_groupby_count_distinct(_p(ids), _i64(len(ids)), _i64(G), _p(vs), _p(pack_bits(valid, offset)), _i64(offset), _p(out))
    return out[:G]


def groupby_min_max(ids, G, v, valid=None):
    """GroupBy::min_max (src/dataframe.cpp:1602-1696): arrow::compute::MinMax per group -> (min, max, ok)."""
    mn, ok = groupby_agg(AGG_MIN, ids, G, v, valid)
    mx, _ = groupby_agg(AGG_MAX, ids, G, v, valid)
    return mn, mx, ok


# ------------------------------------------------------------------ frame-level aggregates (NDFrame<DataFrame>, src/ndframe.h:329-335)
def frame_agg(kind, cols, valids=None):
    """NDFrame::sum/mean/min/max/count on a DataFrame: GetInternalArray() is ONE ChunkedArray whose chunks are the columns
    (src/ndframe.h:329-335), so the aggregate runs over every value of the frame (src/ndframe.cpp:119-220).  Arrow sums each chunk
    with its pairwise tree and adds the chunk totals in chunk (= column) order; mean = that total / total valid count (int64
    chunks are summed as doubles); min / max keep the first of ties across chunks; count adds up.  -> value | None."""
    valids = valids or [None] * len(cols)
    if kind == AGG_COUNT:
        return sum(agg(AGG_COUNT, c, v)[0] for c, v in zip(cols, valids))
    isf = any(np.asarray(c).dtype == np.float64 for c in cols)
    if kind in (AGG_SUM, AGG_MEAN):
        tot, cnt, wrap = 0.0 if (isf or kind == AGG_MEAN) else 0, 0, not isf and kind == AGG_SUM
        for c, v in zip(cols, valids):
            c = np.asarray(c)
            s, k = agg(AGG_SUM, c.astype(np.float64) if (kind == AGG_MEAN and c.dtype != np.float64) else c, v)
            if s is None:
                continue
            tot = ((tot + s + 2**63) % 2**64 - 2**63) if wrap else tot + s
            cnt += k
        if cnt == 0:
            return None
        return tot / cnt if kind == AGG_MEAN else tot
    best = None
    for c, v in zip(cols, valids):
        x, _ = agg(kind, c, v)
        if x is None:
            continue
        if best is None or (best != best and x == x) or (x < best if kind == AGG_MIN else x > best):
            best = x
    return best
