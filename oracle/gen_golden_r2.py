#!/usr/bin/env python3
"""Generate tests/golden/arrow_golden_r2.npz -- golden vectors for the rows added after round 1.

TEST INFRASTRUCTURE (same role and conventions as oracle/gen_golden.py, which stays frozen so that
tests/golden/arrow_golden.npz does not churn).  Drives Arrow C++ 25.0.0 through pyarrow, replaying the
reference's call sequences:

  * Scalar op Series:  Scalar::operator{+,-,*,/,<,...}(Series) -> BinaryImpl -> CallFunction(name, {scalar, array})
                       src/scalar.cpp:24-56
  * DataFrame::downsample: FloorTemporal / CeilTemporal(index, RoundTemporalOptions(multiple, unit, week_starts_monday,
                       ceil_is_strictly_greater=false, calendar_based_origin)) [+ Subtract(one day) for M / W / Q rules]
                       src/dataframe.cpp:1265-1290
  * GroupBy all/any/count_distinct/min_max: GROUPBY_NUMERIC_AGG(all|any|count_distinct), GroupBy::min_max
                       src/dataframe.cpp:1520-1526, 1602-1696
  * DataFrame-level sum/mean/min/max/count: NDFrame<>::GetInternalArray = ChunkedArray of all columns
                       src/ndframe.h:329-335, src/ndframe.cpp:119-220

Run:  python oracle/gen_golden_r2.py
"""
import json
import os
import sys

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "arrow_golden_r2.npz")
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
    """pyarrow array -> (values with nulls zeroed, valid bool)."""
    if isinstance(a, pa.ChunkedArray):
        a = a.combine_chunks()
    valid = np.array([x is not None for x in a.to_pylist()], bool) if a.null_count else np.ones(len(a), bool)
    vals = np.asarray(a.fill_null(0 if not pa.types.is_boolean(a.type) else False).to_numpy(zero_copy_only=False)).astype(dtype)
    return vals, valid


def nan_bits(payload, neg=False):
    return np.array([0x7FF8000000000000 | payload | (0x8000000000000000 if neg else 0)], np.uint64).view(np.float64)[0]


# ------------------------------------------------------------------ Scalar op Series (src/scalar.cpp:24-56)
def gen_scalar_lhs():
    rng = np.random.default_rng(20260201)
    cases = []
    ops = {"add": pc.add, "sub": pc.subtract, "mul": pc.multiply, "div": pc.divide}
    cmps = {"eq": pc.equal, "ne": pc.not_equal, "lt": pc.less, "le": pc.less_equal, "gt": pc.greater, "ge": pc.greater_equal}
    for n in (0, 1, 7, 65, 1000):
        for dt in ("f64", "i64", "si_af", "sf_ai"):  # scalar/array dtype mixes: (f,f) (i,i) (int scalar, float array) (float scalar, int array)
            for nulls in (False, True):
                for snull in (False, True):
                    name = f"ewl_{dt}_{n}_{int(nulls)}_{int(snull)}"
                    arr_is_f = dt in ("f64", "si_af")
                    sc_is_f = dt in ("f64", "sf_ai")
                    if arr_is_f:
                        b = rng.standard_normal(n)
                        if n > 3:
                            b[2], b[3] = np.nan, 0.0
                    else:
                        b = rng.integers(1, 9, n, dtype=np.int64) * rng.choice([-1, 1], n)
                        if n > 3 and dt == "i64":
                            b[0], b[1] = 2, -1
                    if sc_is_f:
                        s = 2.5
                    else:
                        s = int(np.iinfo(np.int64).min) if (dt == "i64" and n == 7) else 7  # INT64_MIN / -1 -> 0, INT64_MIN * 2 wraps
                    vb = (rng.random(n) > 0.2) if nulls else None
                    S = pa.scalar(None, pa.float64() if sc_is_f else pa.int64()) if snull else pa.scalar(s)
                    B = arr(b, vb)
                    rec = dict(b=b, vb=np.ones(n, bool) if vb is None else vb, s=np.array(s), s_valid=np.array(not snull))
                    odt = np.float64 if (arr_is_f or sc_is_f) else np.int64
                    for k, f in ops.items():
                        vals, valid = out_np(f(S, B), odt)
                        rec[k], rec[f"{k}_valid"] = vals, valid
                    for k, f in cmps.items():
                        vals, valid = out_np(f(S, B), bool)
                        rec[k], rec[f"{k}_valid"] = vals, valid
                    put(name, **rec)
                    cases.append(name)
    # NaN payloads: which operand's NaN survives in Scalar op Series (x86 SSE semantics as compiled into Arrow 25.0.0)
    b = np.array([1.0, nan_bits(0x111), nan_bits(0x222, True), np.inf, -np.inf, 0.0, -0.0], np.float64)
    for j, s in enumerate((nan_bits(0x999), nan_bits(0x999, True), np.inf, -np.inf, 0.0, 2.0)):
        name = f"ewl_nanbits_{j}"
        rec = dict(b=b, vb=np.ones(len(b), bool), s=np.array(s), s_valid=np.array(True))
        for k, f in ops.items():
            r = f(pa.scalar(float(s)), pa.array(b)).to_numpy(zero_copy_only=False)
            rec[k], rec[f"{k}_valid"] = r, np.ones(len(b), bool)
        put(name, **rec)
        cases.append(name)
    # integer scalar / array holding a zero at a valid slot raises; under a null slot it does not
    try:
        pc.divide(pa.scalar(7), pa.array([2, 0]))
        raise SystemExit("expected divide by zero")
    except pa.ArrowInvalid as e:
        manifest["scalar_lhs_div_by_zero_message"] = str(e)
    assert pc.divide(pa.scalar(7), pa.array([2, 0], mask=np.array([False, True]))).to_pylist() == [3, None]
    manifest["cases"]["scalar_lhs"] = cases


SECTIONS = [gen_scalar_lhs]


def main():
    for f in SECTIONS:
        f()
    store["manifest"] = np.array(json.dumps(manifest))
    np.savez_compressed(OUT, **store)
    print(f"wrote {OUT}: {len(store)} arrays, {os.path.getsize(OUT)} bytes; arrow {pa.__version__}")


if __name__ == "__main__":
    main()
