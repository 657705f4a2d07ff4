"""Device-resident Arrow-layout columns and the thin Python calls into the C ABI.

A ``Column`` owns torch CUDA tensors (PyTorch is used only for device memory and streams): a contiguous
values buffer, an optional validity bitmap (1 bit/row, LSB first) and an element ``offset`` -- the fields
of ``arrow::ArrayData`` the reference's kernels read (src/ndframe.h:140-157).  Every operation below is one
call through ``libpdx_hip.so``; nothing is computed in Python/torch.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L

_TORCH_DT = {L.INT64: torch.int64, L.FLOAT64: torch.float64, L.UINT64: torch.int64, L.TIMESTAMP_NS: torch.int64, L.BOOL: torch.uint8}


def _device():
    if not torch.cuda.is_available():
        raise L.PdxError(L.DEVICE, "no GPU visible: pandasarrow_amd runs on MI355X (gfx950) only; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _bitmap_bytes(n):
    return (n + 7) // 8 + 16  # slack so kernels may store whole 64-bit words


def pack_bits_host(b, offset=0):
    b = np.asarray(b, dtype=bool)
    if offset:
        b = np.concatenate([np.zeros(offset, bool), b])
    bits = np.packbits(b, bitorder="little")
    return np.concatenate([bits, np.zeros(16, np.uint8)])


def unpack_bits_host(bits, n, offset=0):
    return np.unpackbits(np.asarray(bits, np.uint8), bitorder="little")[offset:offset + n].astype(bool)


class Column:
    """One Arrow-layout column in HBM."""

    __slots__ = ("dtype", "length", "offset", "values", "validity", "null_count")

    def __init__(self, dtype, length, values, validity=None, offset=0, null_count=None):
        self.dtype = dtype
        self.length = int(length)
        self.offset = int(offset)
        self.values = values      # torch tensor (int64/float64, or uint8 bitmap for BOOL)
        self.validity = validity  # torch uint8 bitmap or None
        self.null_count = (0 if validity is None else -1) if null_count is None else int(null_count)

    # ---- construction -------------------------------------------------------------------------------------
    @staticmethod
    def from_numpy(a, valid=None, dtype=None, offset=0):
        """Host -> device.  ``offset`` prepends junk rows to exercise Arrow slice offsets."""
        a = np.asarray(a)
        dev = _device()
        if dtype is None:
            if a.dtype == np.bool_:
                dtype = L.BOOL
            elif a.dtype == np.float64:
                dtype = L.FLOAT64
            elif a.dtype == np.uint64:
                dtype = L.UINT64
            elif a.dtype.kind in "iu":
                dtype = L.INT64
            elif a.dtype.kind == "M":
                dtype = L.TIMESTAMP_NS
                a = a.astype("datetime64[ns]").astype(np.int64)
            else:
                raise L.PdxError(L.INVALID, f"unsupported numpy dtype {a.dtype}")
        n = len(a)
        if dtype == L.BOOL:
            vals = torch.from_numpy(pack_bits_host(a.astype(bool), offset)).to(dev)
        else:
            host = a.astype(np.float64 if dtype == L.FLOAT64 else np.int64 if dtype != L.UINT64 else np.uint64)
            if offset:
                host = np.concatenate([np.full(offset, 77, host.dtype), host])
            vals = torch.from_numpy(np.ascontiguousarray(host).view(np.float64 if dtype == L.FLOAT64 else np.int64).copy()).to(dev)
        vbits = None
        if valid is not None:
            vbits = torch.from_numpy(pack_bits_host(valid, offset)).to(dev)
        return Column(dtype, n, vals, vbits, offset)

    @staticmethod
    def empty(dtype, n, with_validity=False):
        dev = _device()
        if dtype == L.BOOL:
            vals = torch.zeros(_bitmap_bytes(n), dtype=torch.uint8, device=dev)
        else:
            vals = torch.empty(max(n, 1), dtype=_TORCH_DT[dtype], device=dev)
        vb = torch.zeros(_bitmap_bytes(n), dtype=torch.uint8, device=dev) if with_validity else None
        return Column(dtype, n, vals, vb)

    # ---- host views ---------------------------------------------------------------------------------------
    def to_numpy(self):
        """-> (values ndarray, valid bool ndarray | None)."""
        torch.cuda.current_stream().synchronize()
        if self.dtype == L.BOOL:
            vals = unpack_bits_host(self.values.cpu().numpy(), self.length, self.offset)
        else:
            host = self.values[self.offset:self.offset + self.length].cpu().numpy()
            vals = host.view(np.uint64) if self.dtype == L.UINT64 else host
        valid = None
        if self.validity is not None:
            valid = unpack_bits_host(self.validity.cpu().numpy(), self.length, self.offset)
        return vals, valid

    # ---- ABI structs --------------------------------------------------------------------------------------
    def c(self):
        return L.PdxColumn(self.dtype, 0, self.length, self.offset, self.null_count,
                           None if self.validity is None else self.validity.data_ptr(), self.values.data_ptr())

    def mut(self):
        return L.PdxMutColumn(self.dtype, 0, self.length, -1, None if self.validity is None else self.validity.data_ptr(),
                              self.values.data_ptr())

    def _adopt(self, m):
        self.length = int(m.length)
        self.null_count = int(m.null_count)
        return self

    def has_nulls(self):
        return self.validity is not None and self.null_count != 0

    def slice(self, start, length):
        return Column(self.dtype, length, self.values, self.validity, self.offset + start, None if self.validity is not None else 0)


def _scalar_column(x, like_float):
    if isinstance(x, Column):
        return x
    if x is None:
        return Column.from_numpy(np.zeros(1, np.float64 if like_float else np.int64), valid=np.zeros(1, bool))
    if isinstance(x, (float, np.floating)):
        return Column.from_numpy(np.array([x], np.float64))
    return Column.from_numpy(np.array([x], np.int64))


def _promoted(a, b):
    return L.FLOAT64 if L.FLOAT64 in (a.dtype, b.dtype) else L.INT64


# ---------------------------------------------------------------- element-wise
def _scalar_sides(a, b, scalar):
    """-> (a column, b column, pdx_scalar_side, result length).  A non-Column operand on either side is a scalar: on the right
    `series op 2` (src/series.cpp:25-28), on the left `2 op series` (Scalar::operator op(Series), src/scalar.cpp:24-56)."""
    side = L.SCALAR_RHS if scalar is True else int(scalar)
    if not isinstance(a, Column):
        a, side = _scalar_column(a, b.dtype == L.FLOAT64), L.SCALAR_LHS
    elif not isinstance(b, Column):
        b, side = _scalar_column(b, a.dtype == L.FLOAT64), L.SCALAR_RHS
    return a, b, side, (b.length if side == L.SCALAR_LHS else a.length)


def binary(op, a, b, scalar=False) -> Column:
    lib = L.load()
    a, b, side, n = _scalar_sides(a, b, scalar)
    out = Column.empty(_promoted(a, b), n, with_validity=a.has_nulls() or b.has_nulls())
    ca, cb, m = a.c(), b.c(), out.mut()
    L.check(lib.pdx_binary(op, C.byref(ca), C.byref(cb), side, C.byref(m), _stream()))
    return out._adopt(m)


def compare(op, a, b, scalar=False) -> Column:
    lib = L.load()
    a, b, side, n = _scalar_sides(a, b, scalar)
    out = Column.empty(L.BOOL, n, with_validity=a.has_nulls() or b.has_nulls())
    ca, cb, m = a.c(), b.c(), out.mut()
    L.check(lib.pdx_compare(op, C.byref(ca), C.byref(cb), side, C.byref(m), _stream()))
    return out._adopt(m)


def logical(op, a: Column, b: Column) -> Column:
    lib = L.load()
    out = Column.empty(L.BOOL, a.length, with_validity=a.has_nulls() or b.has_nulls())
    ca, cb, m = a.c(), b.c(), out.mut()
    L.check(lib.pdx_logical(op, C.byref(ca), C.byref(cb), C.byref(m), _stream()))
    return out._adopt(m)


def invert(a: Column) -> Column:
    lib = L.load()
    out = Column.empty(L.BOOL, a.length, with_validity=a.has_nulls())
    ca, m = a.c(), out.mut()
    L.check(lib.pdx_invert(C.byref(ca), C.byref(m), _stream()))
    return out._adopt(m)


def null_column(dtype, n) -> Column:
    """n nulls of `dtype` (device-side zeros: values and validity)."""
    nbytes = _bitmap_bytes(n)
    vdt = torch.uint8 if dtype == L.BOOL else (torch.float64 if dtype == L.FLOAT64 else torch.int64)
    vals = torch.zeros(max(nbytes if dtype == L.BOOL else n, 1), dtype=vdt, device=_device())
    return Column(dtype, n, vals, torch.zeros(nbytes, dtype=torch.uint8, device=_device()), 0, n)


def if_else(cond: Column, a, b) -> Column:
    """cond ? a : b (pdx_if_else); a or b may be a python scalar / None (null scalar)."""
    lib = L.load()
    like_float = any(isinstance(x, Column) and x.dtype == L.FLOAT64 for x in (a, b)) or any(isinstance(x, float) for x in (a, b))
    side = L.SCALAR_NONE
    if not isinstance(b, Column):
        b, side = _scalar_column(b, like_float), L.SCALAR_RHS
    elif not isinstance(a, Column):
        a, side = _scalar_column(a, like_float), L.SCALAR_LHS
    out = Column.empty(_promoted(a, b), cond.length, with_validity=cond.has_nulls() or a.has_nulls() or b.has_nulls())
    cc, ca, cb, m = cond.c(), a.c(), b.c(), out.mut()
    L.check(lib.pdx_if_else(C.byref(cc), C.byref(ca), C.byref(cb), side, C.byref(m), _stream()))
    return out._adopt(m)


def unary(op, a: Column) -> Column:
    """negate / abs / sign / sqrt / exp / bit_wise_not of one column (pdx_unary)."""
    lib = L.load()
    to_f64 = op in (L.SQRT, L.EXP)
    out_dt = L.FLOAT64 if to_f64 else (L.INT64 if (op == L.SIGN and a.dtype != L.FLOAT64) else a.dtype)
    out = Column.empty(out_dt, a.length, with_validity=a.has_nulls())
    ca, m = a.c(), out.mut()
    L.check(lib.pdx_unary(int(op), C.byref(ca), C.byref(m), _stream()))
    return out._adopt(m)


def cast_f64(a: Column, checked=True) -> Column:
    """Cast(int64 -> float64) (pdx_cast_f64): checked = Arrow's safe cast (pd::concat's promotion), unchecked = static_cast per value."""
    out = Column.empty(L.FLOAT64, a.length, with_validity=a.has_nulls())
    ca, m = a.c(), out.mut()
    L.check(L.load().pdx_cast_f64(C.byref(ca), int(bool(checked)), C.byref(m), _stream()))
    return out._adopt(m)


def power(a: Column, exponent: float) -> Column:
    lib = L.load()
    out = Column.empty(L.FLOAT64, a.length, with_validity=a.has_nulls())
    ca, m = a.c(), out.mut()
    L.check(lib.pdx_power(C.byref(ca), float(exponent), C.byref(m), _stream()))
    return out._adopt(m)


# ---------------------------------------------------------------- aggregates
def aggregate(kind, a: Column):
    """-> (python value | None, count)."""
    lib = L.load()
    s = L.PdxScalar()
    ca = a.c()
    L.check(lib.pdx_aggregate(kind, C.byref(ca), C.byref(s), _stream()))
    if not s.is_valid:
        return None, int(s.count)
    if s.dtype == L.FLOAT64:
        return float(s.v.f64), int(s.count)
    return int(s.v.i64), int(s.count)


# ---------------------------------------------------------------- filter / take / concat
def _col_array(cols):
    arr = (L.PdxColumn * len(cols))(*[c.c() for c in cols])
    return arr


def _mut_array(cols):
    return (L.PdxMutColumn * len(cols))(*[c.mut() for c in cols])


def filter_count(mask: Column, emit_null=True) -> int:
    lib = L.load()
    out = C.c_int64(0)
    cm = mask.c()
    L.check(lib.pdx_filter_count(C.byref(cm), int(emit_null), C.byref(out), _stream()))
    return int(out.value)


def filter(cols, mask: Column, emit_null=True):
    lib = L.load()
    cm = mask.c()
    if cols and cols[0].length != mask.length:  # same check the kernel would make, before sizing outputs
        L.check(lib.pdx_filter(_col_array(cols), len(cols), C.byref(cm), int(emit_null), _mut_array(cols), _stream()))
    m = filter_count(mask, emit_null)
    nullable = mask.has_nulls() and emit_null
    outs = [Column.empty(c.dtype, m, with_validity=c.has_nulls() or nullable) for c in cols]
    marr = _mut_array(outs)
    L.check(lib.pdx_filter(_col_array(cols), len(cols), C.byref(cm), int(emit_null), marr, _stream()))
    return [o._adopt(marr[i]) for i, o in enumerate(outs)]


def take(cols, idx: Column):
    lib = L.load()
    outs = [Column.empty(c.dtype, idx.length, with_validity=c.has_nulls() or idx.has_nulls()) for c in cols]
    marr = _mut_array(outs)
    ci = idx.c()
    L.check(lib.pdx_take(_col_array(cols), len(cols), C.byref(ci), marr, _stream()))
    return [o._adopt(marr[i]) for i, o in enumerate(outs)]


def scatter(cols, idx: Column, outs):
    """outs[c][idx[j]] = cols[c][j] in place (distinct indices)."""
    lib = L.load()
    marr = _mut_array(outs)
    ci = idx.c()
    L.check(lib.pdx_scatter(_col_array(cols), len(cols), C.byref(ci), marr, _stream()))
    return outs


def concat(parts) -> Column:
    lib = L.load()
    total = sum(p.length for p in parts)
    out = Column.empty(parts[0].dtype, total, with_validity=any(p.has_nulls() for p in parts))
    m = out.mut()
    L.check(lib.pdx_concat(_col_array(parts), len(parts), C.byref(m), _stream()))
    return out._adopt(m)


# ---------------------------------------------------------------- temporal rounding (DataFrame::downsample)
def round_temporal(ts: Column, multiple, unit, ceil=False, week_starts_monday=True, calendar_based_origin=False) -> Column:
    """floor_temporal / ceil_temporal of a timestamp[ns] column (pdx_round_temporal)."""
    out = Column.empty(L.TIMESTAMP_NS, ts.length, with_validity=ts.has_nulls())
    m = out.mut()
    ct = ts.c()
    L.check(L.load().pdx_round_temporal(int(bool(ceil)), C.byref(ct), int(multiple), int(unit), int(bool(week_starts_monday)),
                                        int(bool(calendar_based_origin)), C.byref(m), _stream()))
    return out._adopt(m)


# ---------------------------------------------------------------- index alignment (Series::broadcast / reindex)
def index_union(a: Column, b: Column, sort=True) -> Column:
    """distinct labels of both indexes, sorted ascending or in first-occurrence order (pdx_index_union)."""
    out = Column.empty(a.dtype, a.length + b.length)
    m = out.mut()
    ca, cb = a.c(), b.c()
    L.check(L.load().pdx_index_union(C.byref(ca), C.byref(cb), int(bool(sort)), C.byref(m), _stream()))
    return out._adopt(m)


def index_intersection(a: Column, b: Column) -> Column:
    """labels of a that occur in b, one per distinct label, in a's (last-)position order (pdx_index_intersection)."""
    out = Column.empty(a.dtype, a.length)
    m = out.mut()
    ca, cb = a.c(), b.c()
    L.check(L.load().pdx_index_intersection(C.byref(ca), C.byref(cb), C.byref(m), _stream()))
    return out._adopt(m)


def argsort(a: Column, ascending=True) -> Column:
    """uint64 take indices of the stable sort of a (array_sort_indices: NaN behind the numbers, nulls last, in both orders)."""
    out = Column.empty(L.UINT64, a.length)
    m = out.mut()
    ca = a.c()
    L.check(L.load().pdx_argsort(C.byref(ca), int(bool(ascending)), C.byref(m), _stream()))
    return out._adopt(m)


def reindex_indices(old_index: Column, new_index: Column) -> Column:
    """int64 take indices (LAST position of every new label in old_index, null where absent)."""
    out = Column.empty(L.INT64, new_index.length, with_validity=True)
    m = out.mut()
    co, cn = old_index.c(), new_index.c()
    L.check(L.load().pdx_reindex_indices(C.byref(co), C.byref(cn), C.byref(m), _stream()))
    return out._adopt(m)


# ---------------------------------------------------------------- group-by / resample handles
_AGG_OUT_DT = {L.AGG_MEAN: lambda dt: L.FLOAT64, L.AGG_COUNT: lambda dt: L.INT64, L.AGG_SUM: lambda dt: dt, L.AGG_MIN: lambda dt: dt,
               L.AGG_MAX: lambda dt: dt, L.AGG_VARIANCE: lambda dt: L.FLOAT64, L.AGG_STDDEV: lambda dt: L.FLOAT64,
               L.AGG_PRODUCT: lambda dt: dt, L.AGG_FIRST: lambda dt: dt, L.AGG_LAST: lambda dt: dt,
               L.AGG_ALL: lambda dt: L.BOOL, L.AGG_ANY: lambda dt: L.BOOL, L.AGG_COUNT_DISTINCT: lambda dt: L.INT64}


class GroupByHandle:
    """Owner of a pdx_groupby* (hash group-by or resample segments)."""

    def __init__(self, handle, key_dtype, keep_alive=None):
        self._h = handle
        self.key_dtype = key_dtype
        self._keep = keep_alive  # the resample handle reads the timestamp buffer lazily

    @staticmethod
    def create(key: Column):
        lib = L.load()
        h = C.c_void_p()
        ck = key.c()
        L.check(lib.pdx_groupby_create(C.byref(ck), _stream(), C.byref(h)))
        return GroupByHandle(h, key.dtype)

    @staticmethod
    def resample(ts: Column, freq_ns, closed_right=False, label_right=False, origin=L.ORIGIN_START_DAY, origin_custom_ns=0, offset_ns=0):
        lib = L.load()
        h = C.c_void_p()
        ct = ts.c()
        L.check(lib.pdx_resample_create(C.byref(ct), int(freq_ns), int(closed_right), int(label_right), int(origin), int(origin_custom_ns),
                                        int(offset_ns), _stream(), C.byref(h)))
        return GroupByHandle(h, L.TIMESTAMP_NS, keep_alive=ts)

    @staticmethod
    def downsample(ts: Column, multiple, unit, ceil=False, week_starts_monday=True, calendar_based_origin=False, label_shift_ns=0):
        """Groups of equal Ceil/FloorTemporal labels (+ label_shift_ns) of a timestamp index -- DataFrame::downsample's GroupBy."""
        lib = L.load()
        h = C.c_void_p()
        ct = ts.c()
        L.check(lib.pdx_downsample_create(C.byref(ct), int(multiple), int(unit), int(bool(ceil)), int(bool(week_starts_monday)),
                                          int(bool(calendar_based_origin)), int(label_shift_ns), _stream(), C.byref(h)))
        return GroupByHandle(h, L.TIMESTAMP_NS)

    def close(self):
        if self._h is not None and self._h.value:
            L.load().pdx_groupby_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_groups(self):
        return int(L.load().pdx_groupby_num_groups(self._h))

    @property
    def num_rows(self):
        return int(L.load().pdx_groupby_num_rows(self._h))

    def unique_keys(self) -> Column:
        out = Column.empty(self.key_dtype if self.key_dtype != L.BOOL else L.INT64, self.num_groups, with_validity=True)
        m = out.mut()
        L.check(L.load().pdx_groupby_unique_keys(self._h, C.byref(m), _stream()))
        return out._adopt(m)

    def group_ids(self) -> torch.Tensor:
        out = torch.empty(max(self.num_rows, 1), dtype=torch.int32, device=_device())
        L.check(L.load().pdx_groupby_group_ids(self._h, out.data_ptr(), _stream()))
        return out[: self.num_rows]

    def map_ids(self, mapping: torch.Tensor) -> Column:
        """per-row map[group_id(row)] (int64)."""
        out = Column.empty(L.INT64, self.num_rows)
        L.check(L.load().pdx_groupby_map_ids(self._h, mapping.data_ptr(), out.values.data_ptr(), _stream()))
        return out

    def first_rows(self) -> torch.Tensor:
        out = torch.empty(max(self.num_groups, 1), dtype=torch.int64, device=_device())
        L.check(L.load().pdx_groupby_first_rows(self._h, out.data_ptr(), _stream()))
        return out[: self.num_groups]

    def groupings(self):
        """Grouper::MakeGroupings: (rows int64 [n]: the rows of group 0, then group 1, ..., each ascending; offsets int64 [G + 1])."""
        rows = torch.empty(max(self.num_rows, 1), dtype=torch.int64, device=_device())
        off = torch.empty(self.num_groups + 1, dtype=torch.int64, device=_device())
        L.check(L.load().pdx_groupby_groupings(self._h, rows.data_ptr(), off.data_ptr(), _stream()))
        return rows[: self.num_rows], off

    def row_labels(self) -> torch.Tensor:
        out = torch.empty(max(self.num_rows, 1), dtype=torch.int64, device=_device())
        L.check(L.load().pdx_resample_row_labels(self._h, out.data_ptr(), _stream()))
        return out[: self.num_rows]

    def bind(self, values: Column):
        """Group `values` once and keep the layout (+ later the per-group sum / count / min / max) in the handle: the reference's
        GroupBy constructor does this for every column (processEach, src/dataframe.cpp:1539-1554).  The handle keeps the column alive."""
        cv = values.c()
        L.check(L.load().pdx_groupby_bind(self._h, C.byref(cv), _stream()))
        if not hasattr(self, "_bound"):
            self._bound = []
        self._bound.append(values)
        return self

    def unbind(self, values: Column | None = None):
        if values is None:
            L.check(L.load().pdx_groupby_unbind(self._h, None))
            self._bound = []
        else:
            cv = values.c()
            L.check(L.load().pdx_groupby_unbind(self._h, C.byref(cv)))
            self._bound = [b for b in getattr(self, "_bound", []) if b is not values]

    def bound_bytes(self) -> int:
        return int(L.load().pdx_groupby_bound_bytes(self._h))

    def bind_limit(self, max_bytes: int):
        L.check(L.load().pdx_groupby_bind_limit(self._h, int(max_bytes)))

    def last_plan(self) -> dict:
        """The path the last agg() took: {'slots': 'dense', 'sort': 'narrow:7+7', 'layout': 'fused', 'reducer': 'flr_reduce_dense', ...}."""
        buf = C.create_string_buffer(512)
        L.check(L.load().pdx_groupby_last_plan(self._h, buf, 512))
        return dict(w.split("=", 1) for w in buf.value.decode().split() if "=" in w)

    def agg(self, values: Column, kinds):
        """All `kinds` from one grouped pass.  -> list of Columns (G rows, group-id order)."""
        kinds = list(kinds)
        G = self.num_groups
        outs = [Column.empty(_AGG_OUT_DT[k](values.dtype), G, with_validity=k in (L.AGG_ALL, L.AGG_ANY) or (
            values.has_nulls() and k not in (L.AGG_COUNT, L.AGG_COUNT_DISTINCT))) for k in kinds]
        marr = _mut_array(outs)
        karr = (C.c_int * len(kinds))(*kinds)
        cv = values.c()
        L.check(L.load().pdx_groupby_agg(self._h, C.byref(cv), karr, len(kinds), marr, _stream()))
        return [o._adopt(marr[i]) for i, o in enumerate(outs)]


class GroupedValues:
    """Owner of a pdx_grouped*: one float64 column stably sorted by group (for the multi-GPU partial-tree exchange)."""

    def __init__(self, gb: GroupByHandle, values: Column):
        self._gb = gb  # keep the group-by handle alive
        self._h = C.c_void_p()
        cv = values.c()
        L.check(L.load().pdx_groupby_group_values(gb._h, C.byref(cv), _stream(), C.byref(self._h)))
        self.total = None

    def close(self):
        if self._h is not None and self._h.value:
            L.load().pdx_grouped_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def counts(self) -> torch.Tensor:
        out = torch.empty(max(self._gb.num_groups, 1), dtype=torch.int64, device=_device())
        L.check(L.load().pdx_grouped_counts(self._h, out.data_ptr(), _stream()))
        return out[: self._gb.num_groups]

    def partial_plan(self, prefix: torch.Tensor, order: torch.Tensor | None = None) -> int:
        self._prefix = prefix.contiguous()  # read again by partial_fill
        self._order = None if order is None else order.contiguous()
        total = C.c_int64(0)
        L.check(L.load().pdx_grouped_partial_plan(self._h, self._prefix.data_ptr(), None if self._order is None else self._order.data_ptr(),
                                                  C.byref(total), _stream()))
        self.total = int(total.value)
        return self.total

    def partial_fill(self, gid_map: torch.Tensor):
        """-> (rec_key int64[total], rec_val float64[total]) device tensors."""
        dev = _device()
        key = torch.empty(max(self.total, 1), dtype=torch.int64, device=dev)
        val = torch.empty(max(self.total, 1), dtype=torch.float64, device=dev)
        gm = gid_map.contiguous()
        L.check(L.load().pdx_grouped_partial_fill(self._h, gm.data_ptr(), key.data_ptr(), val.data_ptr(), _stream()))
        return key[: self.total], val[: self.total]


def replay_partials(rec_key: torch.Tensor, rec_val: torch.Tensor, gid_lo: int, n_own: int) -> torch.Tensor:
    out = torch.empty(max(n_own, 1), dtype=torch.float64, device=_device())
    rk, rv = rec_key.contiguous(), rec_val.contiguous()
    L.check(L.load().pdx_replay_partials(rk.data_ptr(), rv.data_ptr(), int(rk.numel()), int(gid_lo), int(n_own), out.data_ptr(), _stream()))
    return out[:n_own]


# ---------------------------------------------------------------- synthetic inputs (bench / tests)
def synth_keys(start, n, num_keys) -> Column:
    out = Column.empty(L.INT64, n)
    L.check(L.load().pdx_synth_keys(int(start), int(n), int(num_keys), out.values.data_ptr(), _stream()))
    return out


def synth_vals(start, n, seed_off=0) -> Column:
    out = Column.empty(L.FLOAT64, n)
    L.check(L.load().pdx_synth_vals(int(start), int(n), int(seed_off), out.values.data_ptr(), _stream()))
    return out


def synth_ts(start, n, t0_ns, step_ns) -> Column:
    out = Column.empty(L.TIMESTAMP_NS, n)
    L.check(L.load().pdx_synth_ts(int(start), int(n), int(t0_ns), int(step_ns), out.values.data_ptr(), _stream()))
    return out


# ---------------------------------------------------------------- Arrow IPC streams <-> device columns (pdx_ipc_*)
class _FrameMemory:
    """A window of device memory owned by a pdx_ipc_frame, exposed through the CUDA array interface so that a torch tensor can
    alias it without a copy (the tensor keeps this object, and through it the frame, alive)."""

    def __init__(self, frame, ptr, count, typestr):
        self._frame = frame
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class IpcFrame:
    """Owner of a pdx_ipc_frame*: the parsed schema / record batch of ONE Arrow IPC stream and, after load(), its device copy."""

    def __init__(self, blob):
        lib = L.load()
        self._blob = bytes(blob) if not isinstance(blob, (bytes, bytearray, memoryview)) else blob  # must outlive load()
        self._h = C.c_void_p()
        buf = (C.c_char * len(self._blob)).from_buffer_copy(self._blob) if not isinstance(self._blob, bytes) else self._blob
        self._buf = buf
        L.check(lib.pdx_ipc_open(buf, len(self._blob), C.byref(self._h)))
        self.names = [lib.pdx_ipc_column_name(self._h, i).decode() for i in range(lib.pdx_ipc_num_columns(self._h))]
        self.num_rows = int(lib.pdx_ipc_num_rows(self._h))
        self.metadata = {lib.pdx_ipc_metadata_key(self._h, i).decode(): lib.pdx_ipc_metadata_value(self._h, i).decode()
                         for i in range(lib.pdx_ipc_num_metadata(self._h))}

    def __del__(self):
        try:
            if self._h is not None and self._h.value:
                L.load().pdx_ipc_destroy(self._h)
            self._h = None
        except Exception:
            pass

    def schema(self):
        """[(name, pdx dtype, null_count)] without touching the GPU."""
        out = []
        for i, nm in enumerate(self.names):
            c = L.PdxColumn()
            L.check(L.load().pdx_ipc_column(self._h, i, C.byref(c)))
            out.append((nm, int(c.dtype), int(c.null_count)))
        return out

    def load(self):
        """One host->device copy of the record batch body (+ widening kernels for narrow types) -> {name: Column} aliasing it."""
        _device()
        L.check(L.load().pdx_ipc_load(self._h, _stream()))
        self._blob = self._buf = None
        cols = []
        n = self.num_rows
        for i, nm in enumerate(self.names):
            c = L.PdxColumn()
            L.check(L.load().pdx_ipc_column(self._h, i, C.byref(c)))
            if c.dtype == L.BOOL:
                vals = torch.as_tensor(_FrameMemory(self, c.values, (n + 7) // 8 + 8, "|u1"), device=_device())
            else:
                vals = torch.as_tensor(_FrameMemory(self, c.values, max(n, 1), "<f8" if c.dtype == L.FLOAT64 else "<i8"), device=_device())
            vb = None
            if c.validity:
                vb = torch.as_tensor(_FrameMemory(self, c.validity, (n + 7) // 8 + 8, "|u1"), device=_device())
            cols.append((nm, Column(int(c.dtype), n, vals, vb, 0, int(c.null_count))))
        return cols


class ParquetFile:
    """Owner of a pdx_parquet_file*: the parsed footer of ONE Parquet file and, after load(), its decoded device columns
    (DataFrame::readParquet, reference src/dataframe.cpp:646-683)."""

    def __init__(self, blob):
        lib = L.load()
        self._blob = blob if isinstance(blob, bytes) else bytes(blob)  # must outlive load()
        self._h = C.c_void_p()
        L.check(lib.pdx_parquet_open(self._blob, len(self._blob), C.byref(self._h)))
        self.names = [lib.pdx_parquet_column_name(self._h, i).decode() for i in range(lib.pdx_parquet_num_columns(self._h))]
        self.num_rows = int(lib.pdx_parquet_num_rows(self._h))
        self.metadata = {lib.pdx_parquet_metadata_key(self._h, i).decode("utf-8", "replace"): lib.pdx_parquet_metadata_value(self._h, i)
                         for i in range(lib.pdx_parquet_num_metadata(self._h))}

    def __del__(self):
        try:
            if self._h is not None and self._h.value:
                L.load().pdx_parquet_destroy(self._h)
            self._h = None
        except Exception:
            pass

    def schema(self):
        """[(name, pdx dtype, null_count from the chunk statistics or -1)] without touching the GPU."""
        out = []
        for i, nm in enumerate(self.names):
            c = L.PdxColumn()
            L.check(L.load().pdx_parquet_column(self._h, i, C.byref(c)))
            out.append((nm, int(c.dtype), int(c.null_count)))
        return out

    def load(self):
        """One host->device copy of the column chunks + the decode kernels -> [(name, Column)] aliasing the file object's memory."""
        _device()
        L.check(L.load().pdx_parquet_load(self._h, _stream()))
        self._blob = None
        cols = []
        n = self.num_rows
        for i, nm in enumerate(self.names):
            c = L.PdxColumn()
            L.check(L.load().pdx_parquet_column(self._h, i, C.byref(c)))
            if c.dtype == L.BOOL:
                vals = torch.as_tensor(_FrameMemory(self, c.values, (n + 7) // 8 + 8, "|u1"), device=_device())
            else:
                vals = torch.as_tensor(_FrameMemory(self, c.values, max(n, 1), "<f8" if c.dtype == L.FLOAT64 else "<i8"), device=_device())
            vb = None
            if c.validity:
                vb = torch.as_tensor(_FrameMemory(self, c.validity, (n + 7) // 8 + 8, "|u1"), device=_device())
            cols.append((nm, Column(int(c.dtype), n, vals, vb, 0, int(c.null_count))))
        return cols


def parquet_write(cols, names) -> bytes:
    """pdx_parquet_write of device columns -> the bytes of a Parquet file (one row group, PLAIN pages)."""
    lib = L.load()
    arr = _col_array(cols)
    cn = (C.c_char_p * len(names))(*[nm.encode() for nm in names])
    out, size = C.c_void_p(), C.c_size_t()
    L.check(lib.pdx_parquet_write(arr, cn, len(cols), 0, _stream(), C.byref(out), C.byref(size)))
    try:
        return C.string_at(out, size.value)
    finally:
        lib.pdx_parquet_free_blob(out)


def ipc_write(cols, names, metadata=None) -> bytes:
    """One schema + one record batch (+ custom metadata) + end-of-stream, the layout DataFrame::toBinary produces."""
    lib = L.load()
    arr = _col_array(cols)
    cn = (C.c_char_p * max(len(names), 1))(*[nm.encode() for nm in names])
    kv = [x.encode() for k, v in (metadata or {}).items() for x in (k, v)]
    ckv = (C.c_char_p * max(len(kv), 1))(*kv)
    out, size = C.c_void_p(), C.c_size_t()
    L.check(lib.pdx_ipc_write(arr, cn, len(cols), ckv, len(kv) // 2, 0, _stream(), C.byref(out), C.byref(size)))
    try:
        return C.string_at(out, size.value)
    finally:
        lib.pdx_ipc_free_blob(out)
