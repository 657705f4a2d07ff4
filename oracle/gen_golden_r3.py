#!/usr/bin/env python3
"""Generate tests/golden/arrow_golden_r3.npz -- golden vectors for the rows closed in round 3.

TEST INFRASTRUCTURE (same role and conventions as oracle/gen_golden.py / gen_golden_r2.py, which stay frozen).  Drives Arrow C++
25.0.0 through pyarrow, replaying the reference's call sequences:

  * DataFrame comparisons / logical operators: BINARY_OPERATOR_DF(> >= < <= == != && ||) -> BinaryFunction(name, self, other) =
        CallFunction(name, {self.GetChunkedArray(), other})  with other = the other frame's ChunkedArray, a ChunkedArray holding the
        Series' array once per column, or the Scalar                                    src/dataframe.cpp:233-249, 563-577
  * Series / DataFrame::reindex(newIndex, fillValue): map label -> LAST position, then per new label
        AppendScalar(value at that position) | (fillValue ? AppendScalar(*fillValue) : AppendNull())
                                                                                        src/series.cpp:1255-1309, dataframe.cpp:1139-1186
    (reference-owned loop, no Arrow kernel: restated here with pyarrow builders element by element, exactly as the loop reads)

Run:  python oracle/gen_golden_r3.py
"""
import json
import os
import sys

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "arrow_golden_r3.npz")
store = {}
manifest = {"arrow_version": pa.__version__, "cases": {}}


def put(case, **arrays):
    for k, v in arrays.items():
        store[f"{case}/{k}"] = np.asarray(v)


def arr(v, valid=None, typ=None):
    v = np.asarray(v)
    mask = None if valid is None else ~np.asarray(valid, bool)
    return pa.array(v, type=typ, mask=mask)


def out_np(a, dtype):
    if isinstance(a, pa.ChunkedArray):
        a = a.combine_chunks()
    valid = np.array([x is not None for x in a.to_pylist()], bool) if a.null_count else np.ones(len(a), bool)
    vals = np.asarray(a.fill_null(0 if not pa.types.is_boolean(a.type) else False).to_numpy(zero_copy_only=False)).astype(dtype)
    return vals, valid


CMPS = {"eq": "equal", "ne": "not_equal", "lt": "less", "le": "less_equal", "gt": "greater", "ge": "greater_equal"}


def frame_cols(rng, n, ncols, dt, nulls):
    cols, valids = [], []
    for c in range(ncols):
        if dt == "f64":
            v = np.round(rng.standard_normal(n) * 2.0) / 2.0  # many ties with the rhs
            if n > 4:
                v[1], v[3] = np.nan, -0.0
        elif dt == "i64":
            v = rng.integers(-3, 4, n).astype(np.int64)
        else:
            v = rng.random(n) > 0.5
        ok = (rng.random(n) > 0.25) if nulls else np.ones(n, bool)
        cols.append(v)
        valids.append(ok)
    return cols, valids


def gen_frame_compare():
    rng = np.random.default_rng(20260301)
    cases = []
    for n in (0, 1, 9, 64, 1001):
        for dt in ("f64", "i64"):
            for nulls in (False, True):
                ncols = 3
                a_cols, a_ok = frame_cols(rng, n, ncols, dt, nulls)
                b_cols, b_ok = frame_cols(rng, n, ncols, dt, nulls)
                s_col, s_ok = frame_cols(rng, n, 1, dt, nulls)
                lhs = pa.chunked_array([arr(v, ok) for v, ok in zip(a_cols, a_ok)], type=pa.float64() if dt == "f64" else pa.int64())
                rhs_frame = pa.chunked_array([arr(v, ok) for v, ok in zip(b_cols, b_ok)], type=lhs.type)
                rhs_series = pa.chunked_array([arr(s_col[0], s_ok[0])] * ncols, type=lhs.type)  # BinaryFunction: the Series once per column
                scalars = [pa.scalar(0.5 if dt == "f64" else 1), pa.scalar(None, type=lhs.type)]
                if dt == "f64":
                    scalars.append(pa.scalar(float("nan")))
                name = f"fcmp_{dt}_{n}_{int(nulls)}"
                put(name, **{f"a{c}": a_cols[c] for c in range(ncols)}, **{f"a{c}_valid": a_ok[c] for c in range(ncols)},
                    **{f"b{c}": b_cols[c] for c in range(ncols)}, **{f"b{c}_valid": b_ok[c] for c in range(ncols)},
                    s=s_col[0], s_valid=s_ok[0],
                    scalars=np.array([0.0 if s.as_py() is None else s.as_py() for s in scalars], np.float64),
                    scalars_valid=np.array([s.as_py() is not None for s in scalars], bool))
                for short, fn in CMPS.items():
                    for tag, rhs in (("frame", rhs_frame), ("series", rhs_series)):
                        res = pc.call_function(fn, [lhs, rhs])
                        v, ok = out_np(res, np.bool_)
                        put(name, **{f"{short}_{tag}": v, f"{short}_{tag}_valid": ok})
                    for si, sc in enumerate(scalars):
                        v, ok = out_np(pc.call_function(fn, [lhs, sc]), np.bool_)
                        put(name, **{f"{short}_scalar{si}": v, f"{short}_scalar{si}_valid": ok})
                cases.append(name)
    manifest["cases"]["frame_compare"] = cases


def gen_frame_logical():
    rng = np.random.default_rng(20260302)
    cases = []
    for n in (0, 1, 9, 64, 1001):
        for nulls in (False, True):
            ncols = 2
            a_cols, a_ok = frame_cols(rng, n, ncols, "bool", nulls)
            b_cols, b_ok = frame_cols(rng, n, ncols, "bool", nulls)
            s_col, s_ok = frame_cols(rng, n, 1, "bool", nulls)
            lhs = pa.chunked_array([arr(v, ok) for v, ok in zip(a_cols, a_ok)], type=pa.bool_())
            rhs_frame = pa.chunked_array([arr(v, ok) for v, ok in zip(b_cols, b_ok)], type=pa.bool_())
            rhs_series = pa.chunked_array([arr(s_col[0], s_ok[0])] * ncols, type=pa.bool_())
            scalars = [pa.scalar(True), pa.scalar(False), pa.scalar(None, type=pa.bool_())]
            name = f"flog_{n}_{int(nulls)}"
            put(name, **{f"a{c}": a_cols[c] for c in range(ncols)}, **{f"a{c}_valid": a_ok[c] for c in range(ncols)},
                **{f"b{c}": b_cols[c] for c in range(ncols)}, **{f"b{c}_valid": b_ok[c] for c in range(ncols)}, s=s_col[0], s_valid=s_ok[0])
            for short, fn in (("and", "and"), ("or", "or")):
                for tag, rhs in (("frame", rhs_frame), ("series", rhs_series)):
                    v, ok = out_np(pc.call_function(fn, [lhs, rhs]), np.bool_)
                    put(name, **{f"{short}_{tag}": v, f"{short}_{tag}_valid": ok})
                for si, sc in enumerate(scalars):
                    v, ok = out_np(pc.call_function(fn, [lhs, sc]), np.bool_)
                    put(name, **{f"{short}_scalar{si}": v, f"{short}_scalar{si}_valid": ok})
            v, ok = out_np(pc.call_function("invert", [lhs]), np.bool_)
            put(name, invert=v, invert_valid=ok)
            cases.append(name)
    manifest["cases"]["frame_logical"] = cases


def reindex_loop(values, old_index, new_index, fill):
    """Series::reindex as the reference's loop reads (src/series.cpp:1275-1305): indexer[label] = i overwrites (LAST position wins)."""
    indexer = {}
    for i, lab in enumerate(old_index.to_pylist()):
        indexer[lab] = i
    out = []
    for lab in new_index.to_pylist():
        if lab in indexer:
            out.append(values[indexer[lab]].as_py())  # AppendScalar(GetScalar(i)): a null value stays null
        else:
            out.append(fill)                           # AppendScalar(*fillValue) | AppendNull()
    return pa.array(out, type=values.type)


def gen_reindex_fill():
    rng = np.random.default_rng(20260303)
    cases = []
    for n_old, n_new in ((0, 5), (5, 0), (6, 9), (200, 333), (5000, 7000)):
        for dt in ("f64", "i64"):
            for nulls in (False, True):
                for dup in (False, True):
                    old = rng.integers(-50, 50 + 4 * n_old, n_old).astype(np.int64) if dup else rng.permutation(4 * n_old + 8)[:n_old].astype(np.int64) - 3
                    new = rng.integers(-60, 60 + 4 * n_old, n_new).astype(np.int64)
                    vals = rng.standard_normal(n_old) if dt == "f64" else rng.integers(-1000, 1000, n_old).astype(np.int64)
                    ok = (rng.random(n_old) > 0.3) if nulls else np.ones(n_old, bool)
                    va = arr(vals, ok, pa.float64() if dt == "f64" else pa.int64())
                    name = f"rix_{dt}_{n_old}_{n_new}_{int(nulls)}_{int(dup)}"
                    put(name, old_index=old, new_index=new, values=vals, values_valid=ok)
                    for tag, fill in (("null", None), ("fill", -7.25 if dt == "f64" else -7)):
                        v, vok = out_np(reindex_loop(va, pa.array(old), pa.array(new), fill), np.float64 if dt == "f64" else np.int64)
                        put(name, **{f"out_{tag}": v, f"out_{tag}_valid": vok})
                    put(name, fill=np.array([-7.25 if dt == "f64" else -7], np.float64))
                    cases.append(name)
    manifest["cases"]["reindex_fill"] = cases


if __name__ == "__main__":
    gen_frame_compare()
    gen_frame_logical()
    gen_reindex_fill()
    store["manifest"] = np.array(json.dumps(manifest))
    np.savez_compressed(OUT, **store)
    print(f"wrote {OUT}: {sum(len(v) for v in manifest['cases'].values())} cases, {len(store)} arrays, {os.path.getsize(OUT) / 1e3:.0f} kB")
