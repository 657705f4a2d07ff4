"""Host-side mirror of the reference's operator interface for the hot path (names, argument meaning, errors).

This is the Python twin of the C++ facade in ``pandasarrow_amd/cpp`` and exists so the parity tests read like the
reference's own Catch2 tests.  Reference surface mirrored (file:line under the reference repository):

  pd::Series      operators + - * / < <= > >= == != & | ~ ; where/take/operator[] ; sum/mean/min/max/count
                  src/series.h:20-516, src/series.cpp:19-33,130-159,229-261 ; src/ndframe.cpp:26-31,119-220,347-350
  pd::DataFrame   element-wise ops over all columns, where/take, group_by, resample        src/dataframe.h:75-709
  pd::GroupBy     groupSize/unique/sum/mean/min/max/count                                    src/group_by.h:22-299
  pd::Resampler   resample(rule, closed_right, label_right, origin, offset) + aggregations   src/resample.h:51-122
  pd::concat      row concat                                                                 src/concat.h:56-64

Every method is a thin call into libpdx_hip.so through ``column.py``; errors surface as ``PdxError`` (a RuntimeError),
the analogue of the reference's ``std::runtime_error(status.ToString())``.  The default index is an implicit
0..n-1 range (the reference materialises a uint64 range per object, src/ndframe.cpp:100-107; here it is lazy).
"""
from __future__ import annotations

import re

import numpy as np
import torch

from . import _lib as L
from . import column as K
from .column import Column

_RULE_NS = {"T": 60 * 10**9, "min": 60 * 10**9, "S": 10**9, "L": 10**6, "ms": 10**6, "U": 10**3, "us": 10**3, "N": 1, "ns": 1}


def _rule_to_ns(rule):
    """splitTimeSpan + unit table of pd::resample (src/resample.h:51-89); calendar offsets (M/Q/Y/W) are out of scope."""
    if isinstance(rule, (int, np.integer)):
        return int(rule)
    m = re.fullmatch(r"(\d*)([A-Za-z]+)", rule)
    if not m or m.group(2) not in _RULE_NS:
        raise L.PdxError(L.NOT_IMPLEMENTED, f"resample rule '{rule}': only fixed-duration rules [T/min S L/ms U/us N/ns] are supported")
    return int(m.group(1) or 1) * _RULE_NS[m.group(2)]


# getCalendarUnit(freq_unit[0]) of DataFrame::downsample (src/core.cpp:135-172): only the FIRST letter of the unit counts,
# so "min" / "ms" both mean millisecond there and upper-case N / U / L are rejected -- kept as the reference has it
_DOWNSAMPLE_UNITS = {"n": L.UNIT_NANOSECOND, "u": L.UNIT_MICROSECOND, "m": L.UNIT_MILLISECOND, "S": L.UNIT_SECOND, "T": L.UNIT_MINUTE,
                     "H": L.UNIT_HOUR, "D": L.UNIT_DAY, "Q": L.UNIT_QUARTER, "W": L.UNIT_WEEK, "M": L.UNIT_MONTH}


def _split_time_span(rule):
    """splitTimeSpan (src/core.cpp:110-133): leading digits = multiple (default 1), the rest = unit."""
    k = 0
    while k < len(rule) and not rule[k].isalpha():
        k += 1
    if k == 0 and any(ch.isdigit() for ch in rule):
        raise L.PdxError(L.INVALID, "Invalid time offset " + rule)
    return (int(rule[:k]) if k else 1), rule[k:]


def _take_filled(cols, idx, fill_value):
    """pdx_take by reindex indices; rows whose INDEX is null (label absent from the old index) get `fill_value` through one
    pdx_if_else per column on the indices' validity bitmap (cond = label present): present-but-null values stay null."""
    outs = K.take(cols, idx)
    if fill_value is None or idx.validity is None:
        return outs
    fv = fill_value.value if isinstance(fill_value, Scalar) else fill_value
    if fv is None:
        return outs
    present = Column(L.BOOL, idx.length, idx.validity, None, idx.offset)
    filled = []
    for c in outs:
        is_float = isinstance(fv, (float, np.floating))
        if c.dtype == L.BOOL or (c.dtype == L.FLOAT64) != is_float:
            have = "double" if is_float else "int64"
            want = {L.FLOAT64: "double", L.BOOL: "bool", L.UINT64: "uint64", L.TIMESTAMP_NS: "timestamp[ns]"}.get(c.dtype, "int64")
            raise L.PdxError(L.INVALID, f"Cannot append scalar of type {have} to builder for type {want}")
        view = Column(L.INT64, c.length, c.values, c.validity, c.offset, c.null_count) if c.dtype in (L.UINT64, L.TIMESTAMP_NS) else c
        r = K.if_else(present, view, fv)
        r.dtype = c.dtype
        filled.append(r)
    return filled


class Scalar:
    """pd::Scalar (src/scalar.h:62-241): a value or null."""

    def __init__(self, value, count=None):
        self.value = value
        self.count = count

    def isValid(self):
        return self.value is not None

    def as_py(self):
        return self.value

    def __eq__(self, other):
        if isinstance(other, (Series, DataFrame)):
            return self._cmp(L.EQ, other)
        return self.value == (other.value if isinstance(other, Scalar) else other)

    def __repr__(self):
        return f"Scalar({self.value!r})"

    __hash__ = None

    # ---- BINARY_OPERATOR_2 (src/scalar.cpp:12-56): Scalar op Series / DataFrame = CallFunction(name, {scalar, array(s)});
    # python scalars on the left reach the same kernels through Series.__radd__ etc.
    def _bin(self, op, other):
        if isinstance(other, DataFrame):
            return other._like([K.binary(op, self.value, c) for c in other.cols])
        if isinstance(other, Series):
            return other._wrap(K.binary(op, self.value, other.col))
        return NotImplemented

    def _cmp(self, op, other):
        if isinstance(other, DataFrame):
            return other._like([K.compare(op, self.value, c) for c in other.cols])
        return other._wrap(K.compare(op, self.value, other.col))

    def __add__(self, o): return self._bin(L.ADD, o)
    def __sub__(self, o): return self._bin(L.SUB, o)
    def __mul__(self, o): return self._bin(L.MUL, o)
    def __truediv__(self, o): return self._bin(L.DIV, o)
    def __lt__(self, o): return self._cmp(L.LT, o)
    def __le__(self, o): return self._cmp(L.LE, o)
    def __gt__(self, o): return self._cmp(L.GT, o)
    def __ge__(self, o): return self._cmp(L.GE, o)
    def __ne__(self, o): return self._cmp(L.NE, o) if isinstance(o, (Series, DataFrame)) else not self.__eq__(o)


class Series:
    def __init__(self, values, valid=None, index=None, name="", is_index=False):
        if isinstance(values, Column):
            self.col = values
        else:
            a = np.asarray(values)
            if a.dtype == np.float64 and valid is None and np.isnan(a).any():
                valid = ~np.isnan(a)  # ArrayT<T>::Make: NaN -> null on construction (src/core.h:404-436)
            self.col = Column.from_numpy(a, valid)
        self.index = index  # Column or None (implicit range)
        self.name = name
        self.is_index = is_index

    # ---- plumbing
    def size(self):
        return self.col.length

    __len__ = size

    def dtype(self):
        return self.col.dtype

    def values(self):
        return self.col.to_numpy()[0]

    def to_numpy(self):
        return self.col.to_numpy()

    def _wrap(self, col, index="same"):
        # ReturnSeriesOrThrowOnError (src/series.cpp:1364-1384): same length -> same index; result name reset to ""
        return Series(col, index=self.index if index == "same" else index, name="")

    def _rhs(self, other):
        if isinstance(other, Series):
            if other.size() != self.size():
                raise L.PdxError(L.INVALID, f"Array arguments must all be the same length: {self.size()} vs {other.size()}")
            return other.col, False
        if isinstance(other, Scalar):
            other = other.value
        return other, True

    # ---- index alignment (Series::broadcast src/series.cpp:212-227, Series::reindex 1255-1309)
    def _explicit_index(self):
        """the index as a column; the implicit range becomes the uint64 0..n-1 the reference materialises (src/ndframe.cpp:100-107)"""
        if self.index is not None:
            return self.index
        return Column(L.UINT64, self.size(), torch.arange(max(self.size(), 1), dtype=torch.int64, device=K._device()), None)

    def _same_index(self, other):
        if self.index is None and other.index is None:
            return self.size() == other.size()
        a, b = self._explicit_index(), other._explicit_index()
        if a.length != b.length or a.dtype != b.dtype:
            return False
        if a.length == 0:
            return True
        ai, bi = (Column(L.INT64, c.length, c.values, None, c.offset) for c in (a, b))  # labels compare as 64-bit patterns
        return K.filter_count(K.compare(L.EQ, ai, bi)) == a.length

    def reindex(self, new_index, fill_value=None):
        """values at the LAST position of every new label; a label the old index lacks gives null, or `fill_value`
        (Series::reindex, src/series.cpp:1255-1309: `fillValue ? AppendScalar(*fillValue) : AppendNull()`, 1295-1302).  A present
        label whose value is null stays null.  The fill value's type must be the column's (Arrow's AppendScalar check)."""
        if not isinstance(new_index, Column):
            new_index = Column.from_numpy(np.asarray(new_index))
        old = self._explicit_index()
        if old.dtype != new_index.dtype:
            raise L.PdxError(L.INVALID, "type(NewIndex) != type(CurrentIndex).")
        idx = K.reindex_indices(old, new_index)
        return Series(_take_filled([self.col], idx, fill_value)[0], index=new_index, name=self.name)

    def broadcast(self, other):
        if self._same_index(other):
            return self, other
        a, b = self._explicit_index(), other._explicit_index()
        if a.dtype != b.dtype:
            raise L.PdxError(L.INVALID, "type(NewIndex) != type(CurrentIndex).")
        union = K.index_union(a, b)
        return self.reindex(union), other.reindex(union)

    def _bin(self, op, other):
        if isinstance(other, Series) and not self._same_index(other):
            x, y = self.broadcast(other)
            return x._wrap(K.binary(op, x.col, y.col, False))
        b, scalar = self._rhs(other)
        return self._wrap(K.binary(op, self.col, b, scalar))

    def _cmp(self, op, other):
        if isinstance(other, Series) and not self._same_index(other):
            x, y = self.broadcast(other)
            return x._wrap(K.compare(op, x.col, y.col, False))
        b, scalar = self._rhs(other)
        return self._wrap(K.compare(op, self.col, b, scalar))

    # ---- Series::operator{+,-,*,/} (src/series.cpp:229-235)
    def __add__(self, o): return self._bin(L.ADD, o)
    def __sub__(self, o): return self._bin(L.SUB, o)
    def __mul__(self, o): return self._bin(L.MUL, o)
    def __truediv__(self, o): return self._bin(L.DIV, o)
    # ---- Scalar::operator{+,-,*,/}(Series) (src/scalar.cpp:24-41): CallFunction(name, {scalar, array}), index = the Series' own
    def _rbin(self, op, o): return self._wrap(K.binary(op, o.value if isinstance(o, Scalar) else o, self.col))
    def __radd__(self, o): return self._rbin(L.ADD, o)
    def __rsub__(self, o): return self._rbin(L.SUB, o)
    def __rmul__(self, o): return self._rbin(L.MUL, o)
    def __rtruediv__(self, o): return self._rbin(L.DIV, o)
    # ---- functions of one column: Series::abs / exp / pow / sign / sqrt (src/series.h:89-109), operator- = "negate"
    def __neg__(self): return self._wrap(K.unary(L.NEGATE, self.col))
    def abs(self): return self._wrap(K.unary(L.ABS, self.col))
    def sign(self): return self._wrap(K.unary(L.SIGN, self.col))
    def sqrt(self): return self._wrap(K.unary(L.SQRT, self.col))
    def exp(self): return self._wrap(K.unary(L.EXP, self.col))
    def pow(self, x): return self._wrap(K.power(self.col, x))
    # ---- comparisons (src/series.cpp:247-257)
    def __lt__(self, o): return self._cmp(L.LT, o)
    def __le__(self, o): return self._cmp(L.LE, o)
    def __gt__(self, o): return self._cmp(L.GT, o)
    def __ge__(self, o): return self._cmp(L.GE, o)
    def __eq__(self, o): return self._cmp(L.EQ, o)  # noqa: E711
    def __ne__(self, o): return self._cmp(L.NE, o)
    __hash__ = None
    # ---- logical (src/series.cpp:259-261,319)
    # `&` / `|` on bool Series are the reference's && / || ("and" / "or", src/series.cpp:259-260); on integers they are its
    # operator& / operator| / ^ / << / >> ("bit_wise_and" ... "shift_right", src/series.cpp:237-245)
    def __and__(self, o): return self._wrap(K.logical(L.AND, self.col, self._rhs(o)[0])) if self.col.dtype == L.BOOL else self._bin(L.BIT_AND, o)
    def __or__(self, o): return self._wrap(K.logical(L.OR, self.col, self._rhs(o)[0])) if self.col.dtype == L.BOOL else self._bin(L.BIT_OR, o)
    def __xor__(self, o): return self._bin(L.BIT_XOR, o)
    def __lshift__(self, o): return self._bin(L.SHIFT_LEFT, o)
    def __rshift__(self, o): return self._bin(L.SHIFT_RIGHT, o)
    def __invert__(self):  # bool: "invert" (src/series.cpp:319); integers: "bit_wise_not" (DataFrame::operator~, src/dataframe.h:500-502)
        return self._wrap(K.invert(self.col) if self.col.dtype == L.BOOL else K.unary(L.BIT_NOT, self.col))

    # ---- NDFrame aggregations (src/ndframe.cpp:119-220)
    def _agg(self, kind):
        v, c = K.aggregate(kind, self.col)
        return Scalar(v, c)

    def sum(self): return self._agg(L.AGG_SUM)
    def mean(self): return self._agg(L.AGG_MEAN)
    def min(self): return self._agg(L.AGG_MIN)
    def max(self): return self._agg(L.AGG_MAX)
    def count(self): return self._agg(L.AGG_COUNT)

    def count_na(self):
        """NDFrame::count_na (src/ndframe.cpp:119-126): CountOptions::ONLY_NULL."""
        return self.size() - int(K.aggregate(L.AGG_COUNT, self.col)[0])

    def _bool_counts(self, what):
        if self.col.dtype != L.BOOL:  # Arrow has no "all" / "any" kernel for non-boolean input: the reference throws
            raise L.PdxError(L.INVALID, f"Function '{what}' has no kernel matching input types")
        true_valid = K.filter_count(self.col, emit_null=False)               # valid AND true
        false_valid = K.filter_count(K.invert(self.col), emit_null=False)    # valid AND false
        if true_valid + false_valid == 0:
            raise L.PdxError(L.INVALID, f"{what}() of a Series without a valid value is null (min_count = 1)")
        return true_valid, false_valid

    def all(self):
        """NDFrame::all (src/ndframe.cpp:110): every VALID value is true (nulls skipped)."""
        return self._bool_counts("all")[1] == 0

    def any(self):
        """NDFrame::any (src/ndframe.cpp:112)."""
        return self._bool_counts("any")[0] > 0

    def unique(self):
        """Series::unique: the distinct values in first-occurrence order (a null, if any, keeps its place) -- the group-by dictionary."""
        key = self.col
        h = K.GroupByHandle.create(key if key.dtype != L.FLOAT64 else Column(L.INT64, key.length, key.values, key.validity, key.offset, key.null_count))
        u = h.unique_keys()
        if key.dtype == L.FLOAT64:
            u = Column(L.FLOAT64, u.length, u.values, u.validity, u.offset, u.null_count)
        return Series(u if u.has_nulls() else Column(u.dtype, u.length, u.values, None, u.offset, 0), name=self.name)

    def nunique(self):
        """Series::nunique: distinct VALID values."""
        u = self.unique().col
        return u.length - (0 if u.validity is None else int(u.length - K.aggregate(L.AGG_COUNT, u)[0]))

    # ---- Series::where / take / operator[] (src/series.cpp:130-159, src/ndframe.cpp:347-350)
    def _index_col(self):
        return self.index

    def if_else(self, cond: "Series", other):
        """Series::if_else / where(cond, other) (src/series.cpp:1203-1209, 1247-1253): cond ? self : other (Series or Scalar)."""
        if cond.col.dtype != L.BOOL:
            raise L.PdxError(L.INVALID, "if_else condition must be boolean")
        if isinstance(other, Series):
            if other.size() != self.size():
                raise L.PdxError(L.INVALID, f"Array arguments must all be the same length: {self.size()} vs {other.size()}")
            other = other.col
        elif isinstance(other, Scalar):
            other = other.value
        if cond.size() != self.size():
            raise L.PdxError(L.INVALID, f"Array arguments must all be the same length: {self.size()} vs {cond.size()}")
        return self._wrap(K.if_else(cond.col, self.col, other))

    def where(self, mask: "Series", other=None):
        if other is not None or isinstance(other, Scalar):
            return self.if_else(mask, other)
        if self.is_index:
            raise L.PdxError(L.INVALID, "where() is not supported on an index Series")
        if mask.col.dtype != L.BOOL:
            raise L.PdxError(L.INVALID, "filter mask must be boolean")
        cols = [self.col] + ([self.index] if self.index is not None else [])
        outs = K.filter(cols, mask.col, emit_null=True)
        return Series(outs[0], index=outs[1] if self.index is not None else None, name=self.name)

    def take(self, idx: "Series"):
        if idx.col.dtype == L.BOOL:
            raise L.PdxError(L.INVALID, "take indices must be integers, not boolean")
        cols = [self.col] + ([self.index] if self.index is not None else [])
        outs = K.take(cols, idx.col)
        return Series(outs[0], index=outs[1] if self.index is not None else None, name=self.name)

    def __getitem__(self, s):
        if isinstance(s, Series):
            return self.where(s) if s.col.dtype == L.BOOL else self.take(s)
        raise TypeError("only Series selectors are on the hot path")

    # ---- sort (src/series.cpp:864-868, 978-992, 1211-1229)
    def argsort(self, ascending=True):
        """Series::argsort: CallFunction("array_sort_indices") -> uint64 indices (re-attached index per ReturnSeriesOrThrowOnError)."""
        return self._wrap(K.argsort(self.col, ascending))

    def sort(self, ascending=True):
        """Series::sort: values AND index taken by the same sort indices."""
        idx = K.argsort(self.col, ascending)
        cols = [self.col, self._explicit_index()]
        outs = K.take(cols, idx)
        return Series(outs[0], index=outs[1], name=self.name)

    def sort_index(self, ascending=True):
        """values and index ordered by the INDEX (the Series form of DataFrame::sort_index, src/dataframe.cpp:1062-1071): a group-by result
        sorted by key compares equal with the reference's whatever order its Grouper numbered the groups in"""
        ix = self._explicit_index()
        idx = K.argsort(ix, ascending)
        outs = K.take([self.col, ix], idx)
        return Series(outs[0], index=outs[1], name=self.name)

    def n_largest(self, n):
        s = self.sort(False)
        return s if s.size() < n else Series(s.col.slice(0, n), index=s.index.slice(0, n), name=self.name)

    def n_smallest(self, n):
        s = self.sort(True)
        return s if s.size() < n else Series(s.col.slice(0, n), index=s.index.slice(0, n), name=self.name)

    def resample(self, rule, closed_right=False, label_right=False, origin=L.ORIGIN_START_DAY, offset_ns=0, origin_custom_ns=0):
        return DataFrame({self.name or "0": self}, index=self.index).resample(rule, closed_right, label_right, origin, offset_ns, origin_custom_ns)


class DataFrame:
    def __init__(self, columns, index=None):
        """columns: dict name -> Series | ndarray | Column;  index: Column | ndarray | None (implicit range)."""
        self.names = list(columns.keys())
        self.cols = []
        for v in columns.values():
            if isinstance(v, Series):
                self.cols.append(v.col)
            elif isinstance(v, Column):
                self.cols.append(v)
            else:
                self.cols.append(Series(v).col)
        n = {c.length for c in self.cols}
        if len(n) > 1:
            raise L.PdxError(L.INVALID, "all columns must have the same length")
        if index is not None and not isinstance(index, Column):
            index = Column.from_numpy(np.asarray(index))
        self.index = index

    def num_rows(self): return self.cols[0].length if self.cols else 0
    def num_columns(self): return len(self.cols)
    def __getitem__(self, name):
        if isinstance(name, Series):
            return self.where(name) if name.col.dtype == L.BOOL else self.take(name)
        return Series(self.cols[self.names.index(name)], index=self.index, name=name)

    def _like(self, cols, index="same"):
        df = DataFrame.__new__(DataFrame)
        df.names, df.cols = list(self.names), cols
        df.index = self.index if index == "same" else index
        return df

    # ---- element-wise over all columns (DataFrame::BinaryFunction, src/dataframe.cpp:233-275)
    def _bin(self, op, other):
        if isinstance(other, DataFrame):
            if other.num_rows() != self.num_rows() or other.num_columns() != self.num_columns():
                raise L.PdxError(L.INVALID, "DataFrame shapes differ")
            return self._like([K.binary(op, a, b) for a, b in zip(self.cols, other.cols)])
        if isinstance(other, Series):
            return self._like([K.binary(op, a, other.col) for a in self.cols])
        if isinstance(other, Scalar):
            other = other.value
        return self._like([K.binary(op, a, other, True) for a in self.cols])

    def __add__(self, o): return self._bin(L.ADD, o)
    def __sub__(self, o): return self._bin(L.SUB, o)
    def __mul__(self, o): return self._bin(L.MUL, o)
    def __truediv__(self, o): return self._bin(L.DIV, o)
    # Scalar::operator op(DataFrame) (src/scalar.cpp:12-29): the scalar stays the left operand for every column
    def _rbin(self, op, o): return self._like([K.binary(op, o.value if isinstance(o, Scalar) else o, c) for c in self.cols])
    def __radd__(self, o): return self._rbin(L.ADD, o)
    def __rsub__(self, o): return self._rbin(L.SUB, o)
    def __rmul__(self, o): return self._rbin(L.MUL, o)
    def __rtruediv__(self, o): return self._rbin(L.DIV, o)
    # BINARY_OPERATOR_DF(> >= < <= == !=) (src/dataframe.cpp:563-573, declared src/dataframe.h:476-520 with DataFrame / Series /
    # Scalar right-hand sides): the compare kernel over every column -> a frame of bit-packed boolean columns
    def _cmp(self, op, other):
        if isinstance(other, DataFrame):
            if other.num_rows() != self.num_rows() or other.num_columns() != self.num_columns():
                raise L.PdxError(L.INVALID, "DataFrame shapes differ")
            return self._like([K.compare(op, a, b) for a, b in zip(self.cols, other.cols)])
        if isinstance(other, Series):
            if other.size() != self.num_rows():
                raise L.PdxError(L.INVALID, f"Array arguments must all be the same length: {self.num_rows()} vs {other.size()}")
            return self._like([K.compare(op, a, other.col) for a in self.cols])
        if isinstance(other, Scalar):
            other = other.value
        return self._like([K.compare(op, a, other, True) for a in self.cols])

    def __lt__(self, o): return self._cmp(L.LT, o)
    def __le__(self, o): return self._cmp(L.LE, o)
    def __gt__(self, o): return self._cmp(L.GT, o)
    def __ge__(self, o): return self._cmp(L.GE, o)
    def __eq__(self, o): return self._cmp(L.EQ, o)  # noqa: E711
    def __ne__(self, o): return self._cmp(L.NE, o)
    __hash__ = None

    # BINARY_OPERATOR_DF(&&, and) / (||, or) (src/dataframe.cpp:575-577): non-Kleene "and" / "or" over boolean frames.  Python has
    # no overloadable && / ||: like Series, `&` / `|` mean these on boolean frames and bit_wise_and / bit_wise_or on integer ones.
    def _logical(self, op, other):
        n = self.num_rows()
        if isinstance(other, DataFrame):
            if other.num_rows() != n or other.num_columns() != self.num_columns():
                raise L.PdxError(L.INVALID, "DataFrame shapes differ")
            rhs = other.cols
        elif isinstance(other, Series):
            if other.size() != n:
                raise L.PdxError(L.INVALID, f"Array arguments must all be the same length: {n} vs {other.size()}")
            rhs = [other.col] * self.num_columns()
        else:  # Scalar / bool / None: Arrow broadcasts the scalar; a null scalar makes every row null
            v = other.value if isinstance(other, Scalar) else other
            rhs = [K.null_column(L.BOOL, n) if v is None else Column.from_numpy(np.full(n, bool(v)))] * self.num_columns()
        if any(c.dtype != L.BOOL for c in list(self.cols) + list(rhs)):
            raise L.PdxError(L.NOT_IMPLEMENTED, 'Function \'and\' / \'or\' has no kernel matching input types (boolean columns expected)')
        return self._like([K.logical(op, a, b) for a, b in zip(self.cols, rhs)])

    def _is_bool(self): return bool(self.cols) and all(c.dtype == L.BOOL for c in self.cols)
    def logical_and(self, o): return self._logical(L.AND, o)
    def logical_or(self, o): return self._logical(L.OR, o)
    # BINARY_OPERATOR_DF(| & ^ << >>) (src/dataframe.cpp:553-561): integer frames
    def __or__(self, o): return self._logical(L.OR, o) if self._is_bool() else self._bin(L.BIT_OR, o)
    def __and__(self, o): return self._logical(L.AND, o) if self._is_bool() else self._bin(L.BIT_AND, o)
    def __xor__(self, o): return self._bin(L.BIT_XOR, o)
    def __lshift__(self, o): return self._bin(L.SHIFT_LEFT, o)
    def __rshift__(self, o): return self._bin(L.SHIFT_RIGHT, o)
    # DataFrame::unary("negate" | "bit_wise_not") and UNARY_FUNCTION(abs | exp | sign | sqrt), pow (src/dataframe.cpp:251-275, 919-935)
    def _unary(self, op): return self._like([K.unary(op, c) for c in self.cols])
    def __neg__(self): return self._unary(L.NEGATE)
    def __invert__(self):  # integer frames: "bit_wise_not" (DataFrame::operator~); boolean frames: "invert" as Series::operator!
        return self._like([K.invert(c) for c in self.cols]) if self._is_bool() else self._unary(L.BIT_NOT)
    def abs(self): return self._unary(L.ABS)
    def sign(self): return self._unary(L.SIGN)
    def sqrt(self): return self._unary(L.SQRT)
    def exp(self): return self._unary(L.EXP)
    def pow(self, x): return self._like([K.power(c, x) for c in self.cols])

    # ---- NDFrame::sum/mean/min/max/count on a DataFrame (src/ndframe.cpp:119-220): GetInternalArray() is ONE ChunkedArray whose
    # chunks are the columns (src/ndframe.h:329-335), so the aggregate runs over every value of the frame.  Arrow reduces a
    # chunked array chunk by chunk (each chunk with its own pairwise tree) and folds the chunk results in chunk = column order.
    def _check_one_dtype(self):
        if len({c.dtype for c in self.cols}) > 1:  # arrow::ChunkedArray needs one type for all its chunks
            raise L.PdxError(L.INVALID, "frame-level aggregates need all columns to have the same dtype")

    def sum(self):
        self._check_one_dtype()
        tot, first = None, True
        for c in self.cols:
            v, _ = K.aggregate(L.AGG_SUM, c)
            if v is None:
                continue
            if first:
                tot = v
            elif isinstance(v, float):
                tot = tot + v
            else:
                tot = (tot + v + 2**63) % 2**64 - 2**63  # int64 wraps
            first = False
        return Scalar(tot)

    def count(self):
        return Scalar(sum(K.aggregate(L.AGG_COUNT, c)[0] for c in self.cols))

    def mean(self):
        """total of the per-chunk pairwise sums (int64 chunks are summed as doubles, like Arrow's mean) / total valid count"""
        self._check_one_dtype()
        tot, cnt = 0.0, 0
        for c in self.cols:
            v, k = K.aggregate(L.AGG_SUM, c if c.dtype == L.FLOAT64 else K.cast_f64(c, checked=False))  # (Arrow's mean: static_cast<double> per value)
            if v is None:
                continue
            tot, cnt = tot + v, cnt + k
        return Scalar(tot / cnt if cnt else None, cnt)

    def _extreme(self, kind):
        self._check_one_dtype()
        best = None
        for c in self.cols:  # the first of ties across chunks wins; a NaN chunk result never replaces a number
            x, _ = K.aggregate(kind, c)
            if x is None:
                continue
            if best is None or (best != best and x == x) or (x < best if kind == L.AGG_MIN else x > best):
                best = x
        return Scalar(best)

    def min(self): return self._extreme(L.AGG_MIN)
    def max(self): return self._extreme(L.AGG_MAX)

    # ---- where / take (src/dataframe.cpp:461-492)
    def where(self, mask: Series):
        if mask.col.dtype != L.BOOL:
            raise L.PdxError(L.INVALID, "filter mask must be boolean")
        cols = self.cols + ([self.index] if self.index is not None else [])
        outs = K.filter(cols, mask.col, emit_null=True)
        return self._like(outs[: len(self.cols)], index=outs[-1] if self.index is not None else None)

    def take(self, idx: Series):
        if idx.col.dtype == L.BOOL:
            raise L.PdxError(L.INVALID, "take indices must be integers, not boolean")
        cols = self.cols + ([self.index] if self.index is not None else [])
        outs = K.take(cols, idx.col)
        return self._like(outs[: len(self.cols)], index=outs[-1] if self.index is not None else None)

    def sort_index(self, ascending=True, ignore_index=False):
        """DataFrame::sort_index (src/dataframe.cpp:1062-1071): the index sorted (Series::sort: array_sort_indices + take), the frame taken
        by the same indices.  Also the order-independent view of a group-by result: group ORDER is first occurrence here and a bounded
        permutation of it in Arrow's Grouper (include/pdx/abi.h at pdx_groupby_create), sorted by key both frames are identical."""
        ix = _frame_index(self)
        idx = K.argsort(ix, ascending)
        outs = K.take(self.cols + [ix], idx)
        return self._like(outs[:-1], index=None if ignore_index else outs[-1])

    def reindex(self, new_index, fill_value=None):
        """DataFrame::reindex / reindexAsync (src/dataframe.cpp:1139-1186, src/dataframe.h:403-406): every column at the LAST position
        of each new label; absent labels -> null or `fill_value`.  One take plan (pdx_reindex_indices) serves all columns."""
        if not isinstance(new_index, Column):
            new_index = Column.from_numpy(np.asarray(new_index))
        old = _frame_index(self)
        if old.dtype != new_index.dtype:
            raise L.PdxError(L.INVALID, "type(NewIndex) != type(CurrentIndex).")
        idx = K.reindex_indices(old, new_index)
        return self._like(_take_filled(self.cols, idx, fill_value), index=new_index)

    reindexAsync = reindex

    # ---- Arrow IPC (src/dataframe.cpp:726-791)
    def toBinary(self, columns=None, index=None, metadata=None) -> bytes:
        """DataFrame::toBinary: one IPC stream (schema + ONE record batch + custom metadata).  Like the reference, every column
        is written whatever `columns` says (it computes the list and then serialises m_array whole); with `index` the index is
        cast to int64 and appended as the LAST column under that name."""
        cols, names = list(self.cols), list(self.names)
        if index is not None:
            ix = _frame_index(self)
            cols.append(Column(L.INT64, ix.length, ix.values, ix.validity, ix.offset, ix.null_count))
            names.append(index)
        return K.ipc_write(cols, names, metadata)

    @staticmethod
    def readBinary(blob, index=None):
        """DataFrame::readBinary: exactly one record batch; the body goes to the device in ONE copy and the columns alias it.
        `index`: that column is taken out of the frame and becomes the index (int64 -> timestamp[ns], as the reference casts)."""
        frame = K.IpcFrame(blob)
        pairs = frame.load()
        idx = None
        if index is not None:
            hit = [i for i, (nm, _) in enumerate(pairs) if nm == index]
            if hit:
                _, c = pairs.pop(hit[0])
                idx = Column(L.TIMESTAMP_NS, c.length, c.values, c.validity, c.offset, c.null_count) if c.dtype == L.INT64 else c
            # (absent: the reference logs `no field "<index>" exist` and carries on without an index)
        df = DataFrame.__new__(DataFrame)
        df.names, df.cols, df.index = [nm for nm, _ in pairs], [c for _, c in pairs], idx
        df.metadata = frame.metadata
        return df

    def toParquet(self, path, index_field=""):
        """DataFrame::toParquet(filepath, indexField) (src/dataframe.cpp:685-724): the columns, and -- when `index_field` is given -- the
        index as a last column of that name (concatenateArraysToRecordBatch), as one row group.  path=None returns the bytes."""
        cols, names = list(self.cols), list(self.names)
        if index_field:
            cols.append(_frame_index(self))
            names.append(index_field)
        blob = K.parquet_write(cols, names)
        if path is None:
            return blob
        with open(path, "wb") as fh:
            fh.write(blob)

    @staticmethod
    def readParquet(path_or_bytes):
        """DataFrame::readParquet (src/dataframe.cpp:646-683): a Parquet file with ONE row group -> a frame on the device.  The
        footer and the page headers are walked on the host; the column chunks travel in one copy and are decoded by kernels
        (Snappy, definition levels, PLAIN / dictionary values).  The reference builds `DataFrame{recordBatch}`: default range index."""
        if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
            blob = bytes(path_or_bytes)
        else:
            try:
                with open(path_or_bytes, "rb") as fh:
                    blob = fh.read()
            except OSError as e:  # arrow::io::ReadableFile::Open fails -> std::runtime_error(status)
                raise L.PdxError(L.INVALID, f"IOError: Failed to open local file '{path_or_bytes}': {e.strerror}") from None
        pf = K.ParquetFile(blob)
        pairs = pf.load()
        df = DataFrame.__new__(DataFrame)
        df.names, df.cols, df.index = [nm for nm, _ in pairs], [c for _, c in pairs], None
        df.metadata = pf.metadata
        return df

    # ---- group_by / resample (src/dataframe.cpp:1227-1262)
    def group_by(self, key):
        return GroupBy(key, self)

    def resample(self, rule, closed_right=False, label_right=False, origin=L.ORIGIN_START_DAY, offset_ns=0, origin_custom_ns=0):
        if self.index is None:
            raise L.PdxError(L.INVALID, "axis must be a TimestampArray but got the implicit range index")
        return Resampler(self, _rule_to_ns(rule), closed_right, label_right, origin, origin_custom_ns, offset_ns)

    def downsample(self, rule, closed_label_right=True, weekStartsMonday=True, startEpoch=True):
        """DataFrame::downsample (src/dataframe.cpp:1265-1290): bin the index with Arrow's ceil_temporal (closed_label_right) or
        floor_temporal -- RoundTemporalOptions(multiple, unit, weekStartsMonday, false, calendar_based_origin = startEpoch) --
        move M / W / Y / Q (and *E) labels back one day to the period's last day, then key a Resampler's GroupBy on the binned
        index (hash grouping in first-occurrence order: the index need not be sorted)."""
        if self.index is None or self.index.dtype != L.TIMESTAMP_NS:
            raise L.PdxError(L.INVALID, "downsample needs a timestamp[ns] index")
        mult, unit_s = _split_time_span(rule)
        if not unit_s or unit_s[0] not in _DOWNSAMPLE_UNITS:  # getCalendarUnit (src/core.cpp:135-172)
            raise L.PdxError(L.INVALID, "invalid unit got " + unit_s[:1])
        unit = _DOWNSAMPLE_UNITS[unit_s[0]]
        one_day_less = unit_s.endswith("E") or unit_s in ("M", "W", "Y", "Q")
        src = self.index

        def binned_index():
            # the frame the reference's Resampler carries is indexed by the rounded labels; nothing on the aggregation path reads
            # that index (the result is indexed by the handle's unique labels), so it is only built when somebody looks at it
            binned = K.round_temporal(src, mult, unit, closed_label_right, weekStartsMonday, startEpoch)
            if one_day_less:
                # Subtract(binned, date32 scalar 1) -> Cast(int64) -> Cast(timestamp[ns]): one day less on the int64 view
                as_i64 = Column(L.INT64, binned.length, binned.values, binned.validity, binned.offset, binned.null_count)
                shifted = K.binary(L.SUB, as_i64, 86400 * 10**9, True)
                binned = Column(L.TIMESTAMP_NS, shifted.length, shifted.values, shifted.validity, 0, shifted.null_count)
            return binned

        framed = _LazyIndexFrame(self.names, self.cols, binned_index)
        handle = K.GroupByHandle.downsample(src, mult, unit, closed_label_right, weekStartsMonday, startEpoch,
                                            -86400 * 10**9 if one_day_less else 0)
        return Resampler(framed, _handle=handle)


class _LazyIndexFrame(DataFrame):
    """A DataFrame whose index column is produced on first access (DataFrame.downsample's rounded index)."""

    def __init__(self, names, cols, make_index):
        self.names, self.cols = list(names), cols
        self._make_index, self._index = make_index, None

    @property
    def index(self):
        if self._index is None and self._make_index is not None:
            self._index, self._make_index = self._make_index(), None
        return self._index

    @index.setter
    def index(self, value):
        self._index, self._make_index = value, None


class GroupBy:
    """pd::GroupBy (src/group_by.h:22-299).  Construction hashes the key column (makeGroups)."""

    def __init__(self, key, df: DataFrame, _handle=None):
        self.df = df
        self.key = key
        self._h = _handle if _handle is not None else K.GroupByHandle.create(df[key].col)
        self._bound_names = set()
        self._groupings_cache = None
        self._keys_host = None

    # ---- walking the groups (src/group_by.h:39-77; dataframe.cpp:1354-1510).  The reference materialises every group's arrays in the
    # constructor; here the groupings (pdx_groupby_groupings) are built on first use and a group's frame is one take.
    def _groupings(self):
        if self._groupings_cache is None:
            rows, off = self._h.groupings()
            self._groupings_cache = (rows, off.cpu().numpy())
        return self._groupings_cache

    def _keys(self):
        if self._keys_host is None:
            vals, ok = self.unique().to_numpy()
            self._keys_host = (vals, ok)
        return self._keys_host

    def GetKeyByIndex(self, i):
        """uniqueKeys->GetScalar(i) (group_by.h:57-60): the key of group i as a python scalar, None for the null key"""
        vals, ok = self._keys()
        if not 0 <= i < len(vals):
            raise L.PdxError(L.INDEX_ERROR, f"Index {i} out of bounds")
        return None if ok is not None and not ok[i] else vals[i].item()

    def _group_index(self, key):
        vals, ok = self._keys()
        if key is None:
            hit = np.flatnonzero(~ok) if ok is not None else np.array([], dtype=np.int64)
        else:
            hit = np.flatnonzero((vals == key) & (ok if ok is not None else True))
        if len(hit) == 0:
            raise KeyError(f"{key} is an invalid key")   # group_by.h:45-49
        return int(hit[0])

    def MakeSubDataFrame(self, groupIndex=None, key=None):
        """MakeSubDataFrame(groupIndex) / MakeSubDataFrame(key) (group_by.h:62-73; python has no overloads, so the key form is spelled
        key=...): the rows of one group, every column and the index, in row order"""
        i = self._group_index(key) if groupIndex is None else int(groupIndex)
        if not 0 <= i < self.groupSize():
            raise L.PdxError(L.INDEX_ERROR, f"Index {i} out of bounds")
        rows, off = self._groupings()
        sel = rows[int(off[i]):int(off[i + 1])]
        idx = Column(L.INT64, int(sel.numel()), sel, None)
        outs = K.take(self.df.cols + [_frame_index(self.df)], idx)
        return self.df._like(outs[:-1], index=outs[-1])

    def group(self, key):
        """GroupBy::group(value) (group_by.h:38-50): the group's column arrays"""
        return self.MakeSubDataFrame(key=key).cols

    def apply(self, fn, per_column=False, _index_keys=False):
        """GroupBy::apply (dataframe.cpp:1430-1510).  fn(DataFrame) -> scalar: Series indexed by the unique keys; fn(DataFrame) -> array of
        the group's length: the arrays concatenated in GROUP order under the frame's own index (as the reference does);
        per_column=True: fn(Series) -> scalar for every column of every group -> DataFrame with one row per group."""
        G = self.groupSize()
        if per_column:
            subs = [self.MakeSubDataFrame(i) for i in range(G)]
            cols = {}
            for nm in self.df.names:
                cols[nm] = _scalars_to_column([fn(Series(sub.cols[sub.names.index(nm)], index=sub.index, name=nm)) for sub in subs])
            return DataFrame(cols, index=self.unique() if _index_keys else None)
        results = [fn(self.MakeSubDataFrame(i)) for i in range(G)]
        if results and isinstance(results[0], (Series, Column, np.ndarray, list)):
            parts = []
            for i, r in enumerate(results):
                c = r.col if isinstance(r, Series) else (r if isinstance(r, Column) else Column.from_numpy(np.asarray(r)))
                rows, off = self._groupings()
                if c.length != int(off[i + 1] - off[i]):
                    raise L.PdxError(L.INVALID, f"Failed to Merge Apply::Functor due to inconsistent Row Length\n{c.length} != {int(off[i + 1] - off[i])}")
                parts.append(c)
            return Series(K.concat(parts), index=self.df.index)
        return Series(_scalars_to_column(results), index=self.unique())

    def apply_async(self, fn, per_column=False):
        """apply_async (dataframe.cpp:1354-1408): the same results (the per-column form is indexed by the unique keys)"""
        return self.apply(fn, per_column, _index_keys=True)

    def apply_chunk(self, fn):
        """apply_chunk (dataframe.cpp:1411-1428): fn(DataFrame) -> DataFrame per group, concatenated along the index"""
        return concat([fn(self.MakeSubDataFrame(i)) for i in range(self.groupSize())], axis="index")

    def getDF(self):
        return self.df

    def groupSize(self):
        return self._h.num_groups

    def unique(self) -> Column:
        return self._h.unique_keys()

    def _col(self, name) -> Column:
        """The frame's column, bound to the handle on first use: the reference's constructor groups every column once (processEach,
        src/dataframe.cpp:1539-1554) and sum() / mean() / count() reuse that (src/group_by.h:85-139); here the first aggregation of a
        column sorts it by group, later ones reuse the layout and the cached per-group results.  The GroupBy holds the frame, so
        the bound buffers stay alive and unchanged."""
        c = self.df.cols[self.df.names.index(name)]
        if c.dtype in (L.INT64, L.FLOAT64) and name not in self._bound_names:
            self._h.bind(c)
            self._bound_names.add(name)
        return c

    def group_ids(self):
        return self._h.group_ids()

    def _agg_frame(self, kind, args):
        single = isinstance(args, str)
        names = [args] if single else list(args)
        uniq = self.unique()
        outs = [self._h.agg(self._col(nm), [kind])[0] for nm in names]
        if single:
            return Series(outs[0], index=uniq, name=names[0])
        return DataFrame(dict(zip(names, outs)), index=uniq)

    def sum(self, args): return self._agg_frame(L.AGG_SUM, args)
    def mean(self, args): return self._agg_frame(L.AGG_MEAN, args)
    def min(self, args): return self._agg_frame(L.AGG_MIN, args)
    def max(self, args): return self._agg_frame(L.AGG_MAX, args)
    def count(self, args): return self._agg_frame(L.AGG_COUNT, args)
    # GROUPBY_NUMERIC_AGG(variance|stddev), GROUPBY_AGG(product), GroupBy::first/last (src/dataframe.cpp:1516-1536, 1698-1810)
    def variance(self, args): return self._agg_frame(L.AGG_VARIANCE, args)
    def stddev(self, args): return self._agg_frame(L.AGG_STDDEV, args)
    def product(self, args): return self._agg_frame(L.AGG_PRODUCT, args)
    def first(self, args): return self._agg_frame(L.AGG_FIRST, args)
    def last(self, args): return self._agg_frame(L.AGG_LAST, args)
    # GROUPBY_NUMERIC_AGG(all | any, bool), GROUPBY_NUMERIC_AGG(count_distinct, int64_t) (src/dataframe.cpp:1520-1526)
    def all(self, args): return self._agg_frame(L.AGG_ALL, args)
    def any(self, args): return self._agg_frame(L.AGG_ANY, args)
    def count_distinct(self, args): return self._agg_frame(L.AGG_COUNT_DISTINCT, args)

    def min_max(self, args):
        """GroupBy::min_max (src/dataframe.cpp:1602-1696): arrow::compute::MinMax per group.  One column name -> frame with
        columns "min", "max"; a list -> "<name>_min", "<name>_max" per column.  Both extremes come from ONE grouped pass."""
        single = isinstance(args, str)
        cols = {}
        for nm in ([args] if single else list(args)):
            mn, mx = self._h.agg(self._col(nm), [L.AGG_MIN, L.AGG_MAX])
            cols["min" if single else nm + "_min"], cols["max" if single else nm + "_max"] = mn, mx
        return DataFrame(cols, index=None)

    def agg(self, name, kinds):
        """sum/mean/count of one column from a single grouped pass (the headline query)."""
        uniq = self.unique()
        outs = self._h.agg(self._col(name), kinds)
        return uniq, outs


class Resampler(GroupBy):
    """pd::Resampler (src/group_by.h:255-299): GroupBy keyed on the per-row bin labels; aggregations run over ALL columns
    and the result is indexed by the labels of the non-empty bins."""

    def __init__(self, df, freq_ns=None, closed_right=False, label_right=False, origin=L.ORIGIN_START_DAY, origin_custom_ns=0, offset_ns=0,
                 _handle=None):
        h = _handle if _handle is not None else K.GroupByHandle.resample(df.index, freq_ns, closed_right, label_right, origin,
                                                                         origin_custom_ns, offset_ns)
        super().__init__("__resampler_idx__", df, _handle=h)

    def index(self) -> Column:
        return self.unique()

    def _all(self, kind):
        return self._agg_frame(kind, list(self.df.names))

    def sum(self, args=None): return self._all(L.AGG_SUM) if args is None else super().sum(args)
    def mean(self, args=None): return self._all(L.AGG_MEAN) if args is None else super().mean(args)
    def min(self, args=None): return self._all(L.AGG_MIN) if args is None else super().min(args)
    def max(self, args=None): return self._all(L.AGG_MAX) if args is None else super().max(args)
    def count(self, args=None): return self._all(L.AGG_COUNT) if args is None else super().count(args)


def concat(frames, axis="index", join="outer", ignore_index=False, sort=False):
    """pd::concat(dfs, axis, join, ignore_index, sort) (src/concat.h:56-64, src/concat.cpp).

    axis="index" (rows, src/concat.cpp:116-190): columns are appended by name (common numeric type for same-named columns, nulls
    where a frame lacks a column; join="inner" keeps the columns without nulls), each frame's index is carried along
    (``[0,1,0,1]``) unless ignore_index.
    axis="columns" (src/concat.cpp:192-244): the frames' indexes are merged pairwise -- Series::union_ (distinct labels in
    first-occurrence order) for join="outer", Series::intersection for "inner" -- optionally sorted, every frame whose index
    differs is reindexed onto the merged index (labels it lacks -> null rows), and the columns are laid side by side
    (duplicate names are kept; ignore_index renames them "0", "1", ...)."""
    if axis in ("columns", 1):
        return _concat_columns(frames, join, ignore_index, sort)
    if join not in ("outer", "inner"):
        raise L.PdxError(L.INVALID, "join must be 'outer' or 'inner'")
    # Concatenator::concatenateRows (src/concat.cpp:116-190): same-named columns are promoted to a common numeric type
    # (resolveDuplicateFieldName + promoteTypes, src/concat.cpp:90-114, src/core.cpp:452-485), the schemas are unified in order of
    # first appearance (ConcatenateTables, unify_schemas) with nulls where a frame lacks a column; join="inner" then drops every
    # column that holds a null (the reference's test for "not present in all frames")
    names = []
    for f in frames:
        names += [nm for nm in f.names if nm not in names]
    cols = []
    for nm in names:
        have = [f.cols[f.names.index(nm)] if nm in f.names else None for f in frames]
        dts = {c.dtype for c in have if c is not None}
        if dts <= {L.INT64, L.FLOAT64}:
            dt = L.FLOAT64 if L.FLOAT64 in dts else L.INT64
        elif len(dts) == 1:
            dt = next(iter(dts))
        else:
            raise L.PdxError(L.NOT_IMPLEMENTED, f"concat: column '{nm}' has types without a common numeric type on this path")
        parts = []
        for f, c in zip(frames, have):
            if c is None:
                c = K.null_column(dt, f.num_rows())
            elif c.dtype != dt:  # arrow::compute::Cast(column, double) with default (safe) options, src/concat.cpp:127
                c = K.cast_f64(c, checked=True)
            parts.append(c)
        cols.append(K.concat(parts))
    if join == "inner":
        keep = [i for i, c in enumerate(cols) if c.validity is None or c.null_count == 0]  # (pdx_concat returns the exact null count)
        names, cols = [names[i] for i in keep], [cols[i] for i in keep]
    index = None
    if not ignore_index:
        idx_parts = [f.index if f.index is not None else Column.from_numpy(np.arange(f.num_rows(), dtype=np.uint64)) for f in frames]
        index = K.concat(idx_parts)
    df = DataFrame.__new__(DataFrame)
    df.names, df.cols, df.index = list(names), cols, index
    return df


def _scalars_to_column(vals):
    """buildArray(ScalarVector): Scalars / python scalars (None = null) -> one column; ints stay int64 unless a float is among them"""
    vals = [v.value if isinstance(v, Scalar) else v for v in vals]
    ok = np.array([v is not None for v in vals], dtype=bool)
    isf = any(isinstance(v, (float, np.floating)) for v in vals)
    isb = bool(vals) and all(isinstance(v, (bool, np.bool_)) or v is None for v in vals)
    dt = bool if isb else (np.float64 if isf else np.int64)
    arr = np.array([dt(0) if v is None else v for v in vals], dtype=dt)
    return Column.from_numpy(arr, None if ok.all() else ok)


def _frame_index(f):
    return f.index if f.index is not None else Column(L.UINT64, f.num_rows(), torch.arange(max(f.num_rows(), 1), dtype=torch.int64, device=K._device()), None)


def _same_labels(a: Column, b: Column):
    if a.length != b.length or a.dtype != b.dtype:
        return False
    if a.length == 0:
        return True
    ai, bi = (Column(L.INT64, c.length, c.values, None, c.offset) for c in (a, b))
    return K.filter_count(K.compare(L.EQ, ai, bi)) == a.length


def _concat_columns(frames, join, ignore_index, sort):
    if join not in ("outer", "inner"):
        raise L.PdxError(L.INVALID, "join must be 'outer' or 'inner'")
    if all(f.index is None for f in frames) and len({f.num_rows() for f in frames}) == 1:
        new_index, explicit = None, False  # equal implicit ranges: nothing to align
    else:
        idxs = [_frame_index(f) for f in frames]
        new_index = idxs[0]
        for other in idxs[1:]:  # Concatenator::mergeIndexes (src/concat.cpp:78-88)
            if new_index.dtype != other.dtype:
                raise L.PdxError(L.INVALID, "type(NewIndex) != type(CurrentIndex).")
            new_index = K.index_intersection(new_index, other) if join == "inner" else K.index_union(new_index, other, sort=False)
        if sort:
            new_index = K.index_union(new_index, new_index.slice(0, 0), sort=True)  # labels are distinct: this only sorts them
        explicit = True
    names, cols = [], []
    for f in frames:
        fcols = f.cols
        if explicit and not _same_labels(_frame_index(f), new_index):
            take_idx = K.reindex_indices(_frame_index(f), new_index)  # DataFrame::reindexAsync: one plan, every column gathered once
            fcols = K.take(f.cols, take_idx)
        names += f.names
        cols += fcols
    if ignore_index:
        names = [str(i) for i in range(len(names))]
    df = DataFrame.__new__(DataFrame)
    df.names, df.cols, df.index = names, cols, new_index
    return df
