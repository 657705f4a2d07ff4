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


# ------------------------------------------------------------------ floor_temporal / ceil_temporal + DataFrame::downsample
UNIT_NAMES = ["nanosecond", "microsecond", "millisecond", "second", "minute", "hour", "day", "week", "month", "quarter"]
DAY = 86400 * 10**9


def gen_round_temporal():
    rng = np.random.default_rng(20260202)
    cases = []
    edge = [0, -1, 1, DAY, -DAY, DAY - 1, 7 * DAY, 946684800 * 10**9, 951782400 * 10**9 - 1, 951782400 * 10**9,  # 2000-02-29
            -2208988800 * 10**9, 725753810610691880, -536625883277931502, 1325099338657844886]  # week-origin quirks found by fuzzing
    ts = np.concatenate([rng.integers(-2 * 10**18, 2 * 10**18, 160), rng.integers(-10**13, 10**13, 60),
                         946684800 * 10**9 + rng.integers(0, 400 * DAY, 60), np.array(edge)]).astype(np.int64)
    valid = rng.random(len(ts)) > 0.15
    T = pa.array(ts).cast(pa.timestamp("ns"))
    put("rt_input", ts=ts, valid=valid)
    for ui, unit in enumerate(UNIT_NAMES):
        for mult in (1, 2, 3, 7, 15, 60):
            for cbo in (False, True):
                for wsm in ((True, False) if unit == "week" else (True,)):
                    name = f"rt_{unit}_{mult}_{int(cbo)}_{int(wsm)}"
                    f = pc.floor_temporal(T, multiple=mult, unit=unit, week_starts_monday=wsm, calendar_based_origin=cbo)
                    c = pc.ceil_temporal(T, multiple=mult, unit=unit, week_starts_monday=wsm, calendar_based_origin=cbo,
                                         ceil_is_strictly_greater=False)
                    put(name, unit=ui, multiple=mult, cbo=cbo, wsm=wsm, floor=f.cast(pa.int64()).to_numpy(), ceil=c.cast(pa.int64()).to_numpy())
                    cases.append(name)
    # nulls pass through
    Tn = pa.array(ts, mask=~valid).cast(pa.timestamp("ns"))
    fn = pc.floor_temporal(Tn, multiple=5, unit="minute")
    vals, ok = out_np(fn.cast(pa.int64()), np.int64)
    put("rt_nulls_minute_5", floor=vals, floor_valid=ok)
    # the reference's M / W / Q post-step: Subtract(binned, date32 scalar 1) -> Cast(int64) -> Cast(timestamp[ns])  (src/dataframe.cpp:1277-1285)
    one_day = pa.scalar(1, pa.date32())
    b = pc.ceil_temporal(T, multiple=1, unit="month", calendar_based_origin=True)
    shifted = pc.subtract(b, one_day).cast(pa.int64()).to_numpy()
    assert np.array_equal(shifted, b.cast(pa.int64()).to_numpy() - DAY)
    manifest["month_rule_shift_ns"] = DAY
    manifest["cases"]["round_temporal"] = cases


def reference_downsample(ts, cols, unit, mult, closed_label_right, wsm, start_epoch, minus_day):
    """DataFrame::downsample + Resampler aggregation with Arrow kernels: binned index -> Grouper ids in first-occurrence order
    (dictionary_encode) -> ApplyGroupings (take of each group's rows in row order) -> one scalar aggregate per group."""
    T = pa.array(ts).cast(pa.timestamp("ns"))
    fn = pc.ceil_temporal if closed_label_right else pc.floor_temporal
    kw = dict(multiple=mult, unit=unit, week_starts_monday=wsm, calendar_based_origin=start_epoch)
    if closed_label_right:
        kw["ceil_is_strictly_greater"] = False
    binned = fn(T, **kw)
    if minus_day:
        binned = pc.subtract(binned, pa.scalar(1, pa.date32())).cast(pa.int64()).cast(pa.timestamp("ns"))
    enc = binned.dictionary_encode()
    ids = np.asarray(enc.indices.to_numpy(zero_copy_only=False)).astype(np.uint32)
    labels = enc.dictionary.cast(pa.int64()).to_numpy()
    G = len(labels)
    order = np.argsort(ids, kind="stable")
    offs = np.concatenate([[0], np.cumsum(np.bincount(ids, minlength=G))])
    res = {}
    for cname, (v, valid) in cols.items():
        V = arr(v, valid)
        isf = np.asarray(v).dtype == np.float64
        sums, means, cnts, mins, maxs = [], [], [], [], []
        for g in range(G):
            grp = V.take(pa.array(order[offs[g]:offs[g + 1]]))
            sums.append(pc.sum(grp).as_py())
            means.append(pc.mean(grp).as_py())
            mins.append(pc.min(grp).as_py())
            maxs.append(pc.max(grp).as_py())
            cnts.append(pc.count(grp).as_py())
        dt = np.float64 if isf else np.int64
        nz = lambda xs, d: np.array([0 if x is None else x for x in xs], d)  # noqa: E731
        res[cname] = dict(sum=nz(sums, dt), mean=nz(means, np.float64), min=nz(mins, dt), max=nz(maxs, dt), count=np.array(cnts, np.int64),
                          ok=np.array([x is not None for x in sums], bool))
    return binned.cast(pa.int64()).to_numpy(), labels, res


def gen_downsample():
    rng = np.random.default_rng(20260203)
    cases = []
    t0 = 946684800 * 10**9 + 37 * 10**9 + 123
    for n in (1, 9, 700, 2500):
        for rule, unit, mult, minus_day in (("3T", "minute", 3, False), ("1T", "minute", 1, False), ("250m", "millisecond", 250, False),
                                            ("2H", "hour", 2, False), ("5D", "day", 5, False), ("M", "month", 1, True),
                                            ("W", "week", 1, True), ("2Q", "quarter", 2, True), ("7S", "second", 7, False)):
            for clr in (False, True):
                for shuffled in (False, True):
                    if shuffled != (n == 700):  # the 700-row cases are the shuffled ones
                        continue
                    name = f"ds_{n}_{rule}_{int(clr)}_{int(shuffled)}"
                    span = {"minute": 60 * 10**9, "millisecond": 10**6, "hour": 3600 * 10**9, "day": DAY, "month": 30 * DAY, "week": 7 * DAY,
                            "quarter": 91 * DAY, "second": 10**9}[unit] * mult
                    ts = t0 + np.cumsum(rng.integers(1, max(2, span // 3), n))
                    if n > 4:
                        ts[3] = (ts[3] // span) * span  # exactly on a boundary (for the fixed units)
                        ts = np.sort(ts)
                    if shuffled:  # the reference hash-groups the binned index: order of first occurrence, no sortedness needed
                        ts = ts[rng.permutation(n)]
                    vf = (rng.standard_normal(n) * 1e3).astype(np.float32).astype(np.float64)  # exactly representable in 4 bytes
                    vi = rng.integers(-10**6, 10**6, n, dtype=np.int64)
                    vvalid = rng.random(n) > 0.1
                    binned, labels, res = reference_downsample(ts, {"f": (vf, vvalid), "i": (vi, None)}, unit, mult, clr, True, True, minus_day)
                    # inputs are regenerated by the test from (seed material stored here would double the file): store them compactly
                    rec = dict(ts=ts, vf=vf.astype(np.float32), vi=vi.astype(np.int32), vvalid=vvalid, labels=labels, clr=clr)
                    if n <= 700:
                        rec["binned"] = binned
                    for c in ("f", "i"):
                        for k, v in res[c].items():
                            rec[f"{c}_{k}"] = v
                    put(name, **rec)
                    manifest.setdefault("downsample_rules", {})[name] = rule
                    cases.append(name)
    manifest["cases"]["downsample"] = cases


# ------------------------------------------------------------------ group-by all / any / count_distinct / min_max
def gen_groupby_extra():
    rng = np.random.default_rng(20260204)
    cases = []
    for n in (0, 1, 17, 1000, 20000):
        for card in (1, 7, 1000):
            for nulls in (False, True):
                if n == 0 and (card != 1 or nulls):
                    continue
                name = f"gbx_{n}_{card}_{int(nulls)}"
                keys = rng.integers(-card // 2, card - card // 2, n, dtype=np.int64) * 1000003
                vb = rng.random(n) > (0.3 if card > 7 else 0.02)  # small cards: mostly-true so that `all` is not trivially false
                vf = np.round(rng.standard_normal(n) * 3) / 2.0        # few distinct doubles per group
                if n > 10:
                    vf[rng.integers(0, n, n // 10)] = -0.0
                    vf[rng.integers(0, n, n // 20)] = 0.0
                    vf[rng.integers(0, n, max(1, n // 50))] = np.nan
                    vf[rng.integers(0, n, max(1, n // 50))] = nan_bits(0x77)
                vi = rng.integers(-3, 4, n, dtype=np.int64) * (2**40 + 1)
                vvalid = (rng.random(n) > 0.25) if nulls else None
                K = arr(keys)
                enc = K.dictionary_encode()
                ids = np.asarray(enc.indices.to_numpy(zero_copy_only=False)).astype(np.uint32)
                G = len(enc.dictionary)
                order = np.argsort(ids, kind="stable")
                offs = np.concatenate([[0], np.cumsum(np.bincount(ids, minlength=G))])
                B, F, I = arr(vb, vvalid), arr(vf, vvalid), arr(vi, vvalid)
                alls, anys, cdf, cdi, mnf, mxf = [], [], [], [], [], []
                for g in range(G):
                    rows = pa.array(order[offs[g]:offs[g + 1]])
                    alls.append(pc.all(B.take(rows)).as_py())        # CallFunction("all", {group}, nullptr)
                    anys.append(pc.any(B.take(rows)).as_py())
                    cdf.append(pc.count_distinct(F.take(rows)).as_py())
                    cdi.append(pc.count_distinct(I.take(rows)).as_py())
                    mm = pc.min_max(F.take(rows)).as_py()             # GroupBy::min_max: arrow::compute::MinMax(group)
                    mnf.append(mm["min"])
                    mxf.append(mm["max"])
                nz = lambda xs, d: np.array([0 if x is None else x for x in xs], d)  # noqa: E731
                put(name, keys=keys, vb=vb, vf=vf, vi=vi, vvalid=np.ones(n, bool) if vvalid is None else vvalid, ids=ids,
                    uniq=enc.dictionary.to_numpy(zero_copy_only=False).astype(np.int64),
                    all=nz(alls, bool), any=nz(anys, bool), ok=np.array([x is not None for x in alls], bool),
                    cd_f=np.array(cdf, np.int64), cd_i=np.array(cdi, np.int64), min_f=nz(mnf, np.float64), max_f=nz(mxf, np.float64),
                    ok_mm=np.array([x is not None for x in mnf], bool))
                cases.append(name)
    manifest["cases"]["groupby_extra"] = cases


# ------------------------------------------------------------------ frame-level aggregates: ChunkedArray of all columns
def gen_frame_aggs():
    rng = np.random.default_rng(20260205)
    cases = []
    for ncols in (1, 3, 8):
        for n in (0, 1, 17, 1000):
            for dt in ("f64", "i64"):
                for nulls in (False, True):
                    name = f"fr_{dt}_{ncols}_{n}_{int(nulls)}"
                    if dt == "f64":
                        cols = [rng.standard_normal(n) * 10.0 ** rng.integers(-3, 6, n) if n else np.zeros(0) for _ in range(ncols)]
                        if n > 3:
                            cols[-1][1], cols[0][2] = -0.0, 0.0
                    else:
                        cols = [rng.integers(-2**61, 2**61, n, dtype=np.int64) for _ in range(ncols)]
                    valids = [(rng.random(n) > 0.2) if nulls else np.ones(n, bool) for _ in range(ncols)]
                    if nulls and ncols > 1:
                        valids[1][:] = False  # an all-null column (chunk): contributes nothing
                    ch = pa.chunked_array([arr(c, v) for c, v in zip(cols, valids)], type=pa.float64() if dt == "f64" else pa.int64())
                    res = dict(sum=pc.sum(ch).as_py(), mean=pc.mean(ch).as_py(), min=pc.min(ch).as_py(), max=pc.max(ch).as_py(), count=pc.count(ch).as_py())
                    rec = dict(count=res["count"], isnull=np.array([res[k] is None for k in ("sum", "mean", "min", "max")]),
                               mean=np.float64(np.nan if res["mean"] is None else res["mean"]))
                    odt = np.float64 if dt == "f64" else np.int64
                    for k in ("sum", "min", "max"):
                        rec[k] = np.array(0 if res[k] is None else res[k], odt)
                    for j, (c, v) in enumerate(zip(cols, valids)):
                        rec[f"c{j}"], rec[f"v{j}"] = c, v
                    rec["ncols"] = ncols
                    put(name, **rec)
                    cases.append(name)
    # ties across chunks keep the first: (0.0 | -0.0) and (-0.0 | 0.0)
    for j, (a, b) in enumerate(((0.0, -0.0), (-0.0, 0.0))):
        ch = pa.chunked_array([pa.array([a, 1.0]), pa.array([b, 1.0])])
        put(f"fr_tie_{j}", c0=np.array([a, 1.0]), c1=np.array([b, 1.0]), v0=np.ones(2, bool), v1=np.ones(2, bool), ncols=2,
            min=np.array(pc.min(ch).as_py()), max=np.array(pc.max(ch).as_py()), sum=np.array(pc.sum(ch).as_py()),
            mean=np.float64(pc.mean(ch).as_py()), count=4, isnull=np.zeros(4, bool))
        cases.append(f"fr_tie_{j}")
    manifest["cases"]["frame_aggs"] = cases


SECTIONS = [gen_scalar_lhs, gen_round_temporal, gen_downsample, gen_groupby_extra, gen_frame_aggs]


def main():
    for f in SECTIONS:
        f()
    store["manifest"] = np.array(json.dumps(manifest))
    np.savez_compressed(OUT, **store)
    print(f"wrote {OUT}: {len(store)} arrays, {os.path.getsize(OUT)} bytes; arrow {pa.__version__}")


if __name__ == "__main__":
    main()
