#!/usr/bin/env python3
"""Generate tests/golden/arrow_golden.npz -- golden input/output vectors for the hot path.

TEST INFRASTRUCTURE.  The arithmetic of the reference's vectorized path lives in a
third-party dependency, Apache Arrow C++ (unpinned by the reference; Arrow 25.0.0 is the
version present in this image via the pyarrow wheel).  This script drives that library
through pyarrow, replaying the reference's call sequences:

  * element-wise / compare / logical:  CallFunction("add"...)      src/series.cpp:19-33,229-261
  * whole-array aggregates:            CallFunction("sum"...)      src/ndframe.cpp:26-31
  * filter / take:                     "filter"/"take"             src/dataframe.cpp:461-492
  * group-by: Grouper::Consume (== dictionary_encode: dense ids in first-occurrence order)
              -> ApplyGroupings (== take of each group's rows in row order)
              -> one scalar CallFunction per group                  src/dataframe.cpp:1571-1600,
                                                                    src/pd_core_macros.h:5-147
  * resample binning has no Arrow kernel (reference-owned loops, src/resample.cpp); it is
    pinned by the reference's own tests (tests/golden/kat_reference.json) and cross-checked
    here against pandas' resample bin assignment, which the reference imitates.

Inputs up to a few thousand elements are stored; larger cases are regenerated from the
counter-based generator in oracle/pdx_oracle.c (orc_synth_*), so only expected outputs are
stored.  Run:  python oracle/gen_golden.py
"""
import json
import os
import sys

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle as orc  # noqa: E402  (only for the synthetic input generator)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "arrow_golden.npz")
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
    valid = np.array([x is not None for x in a.to_pylist()], bool) if a.null_count else np.ones(len(a), bool)
    vals = np.asarray(a.fill_null(0 if not pa.types.is_boolean(a.type) else False).to_numpy(zero_copy_only=False)).astype(dtype)
    return vals, valid


def f64_inputs(rng, n, kind):
    if kind == "uniform":
        return rng.random(n)
    if kind == "normal":
        return rng.standard_normal(n) * 1e6
    if kind == "cancel":  # heavy cancellation
        x = rng.standard_normal(n) * 1e12
        x[1::2] = -x[0::2][: len(x[1::2])] + rng.random(len(x[1::2]))
        return x
    if kind == "special":
        x = rng.random(n)
        if n:
            x[rng.integers(0, n, max(1, n // 7))] = -0.0
        if n > 3:
            x[rng.integers(0, n)] = np.inf
        return x
    raise ValueError(kind)


# ------------------------------------------------------------------ aggregates
def gen_aggregates():
    rng = np.random.default_rng(20250101)
    cases = []
    lens = [0, 1, 2, 15, 16, 17, 31, 32, 33, 100, 255, 256, 257, 1000, 4097]
    for n in lens:
        for kind in ("uniform", "normal", "cancel"):
            for nulls in (0.0, 0.2):
                name = f"agg_f64_{kind}_{n}_{int(nulls*100)}"
                v = f64_inputs(rng, n, kind)
                valid = (rng.random(n) >= nulls) if nulls else None
                a = arr(v, valid)
                exp = dict(
                    sum=pc.sum(a).as_py(), mean=pc.mean(a).as_py(), min=pc.min(a).as_py(), max=pc.max(a).as_py(), count=pc.count(a).as_py()
                )
                put(name, v=v, valid=np.ones(n, bool) if valid is None else valid,
                    exp=np.array([np.nan if exp[k] is None else exp[k] for k in ("sum", "mean", "min", "max")], np.float64),
                    isnull=np.array([exp[k] is None for k in ("sum", "mean", "min", "max")]), count=exp["count"])
                cases.append(name)
    # special values: -0.0, inf, nan, all-null, all-nan
    specials = {
        "negzero1": ([-0.0], None), "negzero2": ([-0.0, -0.0], None), "negzero17": ([-0.0] * 17, None),
        "zero_tie_a": ([0.0, -0.0, 0.0], None), "zero_tie_b": ([-0.0, 0.0, -0.0], None),
        "nan_mid": ([1.0, float("nan"), 3.0], None), "all_nan": ([float("nan")] * 3, None),
        "inf_cancel": ([float("inf"), -float("inf"), 1.0], None), "all_null": ([1.0, 2.0], [False, False]),
        "nan_null": ([float("nan"), 5.0, 2.0], [True, False, True]),
    }
    for nm, (v, valid) in specials.items():
        name = f"agg_f64_special_{nm}"
        a = arr(np.array(v, np.float64), valid)
        exp = [pc.sum(a).as_py(), pc.mean(a).as_py(), pc.min(a).as_py(), pc.max(a).as_py()]
        put(name, v=np.array(v, np.float64), valid=np.ones(len(v), bool) if valid is None else np.array(valid),
            exp=np.array([np.nan if e is None else e for e in exp], np.float64), isnull=np.array([e is None for e in exp]),
            count=pc.count(a).as_py())
        cases.append(name)
    # int64 (wrapping sum, mean via double)
    for n in (0, 1, 17, 1000):
        for nulls in (0.0, 0.2):
            name = f"agg_i64_{n}_{int(nulls*100)}"
            v = rng.integers(-2**62, 2**62, n, dtype=np.int64) if n != 17 else rng.integers(-100, 100, n, dtype=np.int64)
            valid = (rng.random(n) >= nulls) if nulls else None
            a = arr(v, valid)
            s, m, lo, hi = pc.sum(a).as_py(), pc.mean(a).as_py(), pc.min(a).as_py(), pc.max(a).as_py()
            put(name, v=v, valid=np.ones(n, bool) if valid is None else valid,
                exp_i=np.array([0 if e is None else e for e in (s, lo, hi)], np.int64), exp_mean=np.float64(np.nan if m is None else m),
                isnull=np.array([e is None for e in (s, m, lo, hi)]), count=pc.count(a).as_py())
            cases.append(name)
    # large: inputs come from the counter-based generator
    for n, seed_off in ((100000, 11), (1000003, 12)):
        name = f"agg_f64_synth_{n}"
        v = orc.synth_vals(0, n, seed_off)
        a = pa.array(v)
        put(name, n=n, seed_off=seed_off, exp=np.array([pc.sum(a).as_py(), pc.mean(a).as_py(), pc.min(a).as_py(), pc.max(a).as_py()]))
        cases.append(name)
    manifest["cases"]["aggregate"] = cases


# ------------------------------------------------------------------ element-wise
def gen_elementwise():
    rng = np.random.default_rng(20250102)
    cases = []
    ops = {"add": pc.add, "sub": pc.subtract, "mul": pc.multiply, "div": pc.divide}
    cmps = {"eq": pc.equal, "ne": pc.not_equal, "lt": pc.less, "le": pc.less_equal, "gt": pc.greater, "ge": pc.greater_equal}
    for n in (0, 1, 7, 64, 65, 1000):
        for dt in ("f64", "i64", "mixed"):
            for nulls in (False, True):
                for scalar in (False, True):
                    name = f"ew_{dt}_{n}_{int(nulls)}_{int(scalar)}"
                    if dt == "i64":
                        a = rng.integers(-50, 50, n, dtype=np.int64)
                        b = rng.integers(1, 9, n, dtype=np.int64) * rng.choice([-1, 1], n)
                        if n > 3:
                            a[0], b[0] = np.iinfo(np.int64).max, 2  # wrap on add/mul
                            a[1], b[1] = np.iinfo(np.int64).min, -1  # INT64_MIN / -1 -> 0
                    else:
                        a = rng.standard_normal(n)
                        b = rng.standard_normal(n) if dt == "f64" else rng.integers(-5, 5, n, dtype=np.int64)
                        if n > 3:
                            a[2] = np.nan
                            if dt == "f64":
                                b[3] = 0.0
                    va = (rng.random(n) > 0.2) if nulls else None
                    vb = (rng.random(n) > 0.2) if nulls else None
                    if scalar:
                        bs = b[0].item() if n else (2 if dt != "f64" else 0.5)
                        B = pa.scalar(bs)
                        vb = None
                    else:
                        B = arr(b, vb)
                    A = arr(a, va)
                    rec = dict(a=a, va=np.ones(n, bool) if va is None else va, vb=np.ones(n, bool) if vb is None else vb)
                    rec["b"] = np.array(bs) if scalar else b
                    odt = np.int64 if dt == "i64" else np.float64
                    for k, f in ops.items():
                        vals, valid = out_np(f(A, B), odt)
                        rec[f"{k}"], rec[f"{k}_valid"] = vals, valid
                    for k, f in cmps.items():
                        vals, valid = out_np(f(A, B), bool)
                        rec[f"{k}"], rec[f"{k}_valid"] = vals, valid
                    put(name, **rec)
                    cases.append(name)
    # integer divide by zero raises
    try:
        pc.divide(pa.array([7, 1]), pa.array([2, 0]))
        raise SystemExit("expected divide by zero")
    except pa.ArrowInvalid as e:
        manifest["div_by_zero_message"] = str(e)
    # a divide-by-zero hidden under a null slot does NOT raise
    r = pc.divide(pa.array([7, None]), pa.array([2, 0]))
    assert r.to_pylist() == [3, None]
    # logical and/or (non-Kleene), invert
    for n in (0, 1, 9, 64, 130):
        name = f"logic_{n}"
        a, b = rng.random(n) > 0.5, rng.random(n) > 0.5
        va, vb = rng.random(n) > 0.2, rng.random(n) > 0.2
        A, B = arr(a, va), arr(b, vb)
        av, avv = out_np(pc.and_(A, B), bool)
        ov, ovv = out_np(pc.or_(A, B), bool)
        iv, ivv = out_np(pc.invert(A), bool)
        put(name, a=a, b=b, va=va, vb=vb, and_=av, and_valid=avv, or_=ov, or_valid=ovv, inv=iv, inv_valid=ivv)
        cases.append(name)
    manifest["cases"]["elementwise"] = cases


# ------------------------------------------------------------------ filter / take
def gen_filter_take():
    rng = np.random.default_rng(20250103)
    cases = []
    for n in (0, 1, 63, 64, 65, 1000, 5000):
        for sel in (0.0, 0.01, 0.5, 1.0):
            for nulls in (False, True):
                name = f"filter_{n}_{int(sel*100)}_{int(nulls)}"
                v = rng.standard_normal(n)
                valid = (rng.random(n) > 0.2) if nulls else None
                mask = rng.random(n) < sel
                mvalid = (rng.random(n) > 0.1) if nulls else None
                V, M = arr(v, valid), arr(mask, mvalid)
                e_vals, e_valid = out_np(pc.filter(V, M, null_selection_behavior="emit_null"), np.float64)
                d_vals, d_valid = out_np(pc.filter(V, M, null_selection_behavior="drop"), np.float64)
                put(name, v=v, valid=np.ones(n, bool) if valid is None else valid, mask=mask,
                    mvalid=np.ones(n, bool) if mvalid is None else mvalid, emit=e_vals, emit_valid=e_valid, drop=d_vals, drop_valid=d_valid)
                cases.append(name)
    for n, m in ((1, 1), (10, 0), (100, 257), (5000, 3000)):
        for nulls in (False, True):
            name = f"take_{n}_{m}_{int(nulls)}"
            v = rng.integers(-1000, 1000, n, dtype=np.int64)
            valid = (rng.random(n) > 0.2) if nulls else None
            idx = rng.integers(0, n, m, dtype=np.int64)
            ivalid = (rng.random(m) > 0.1) if nulls else None
            vals, ok = out_np(pc.take(arr(v, valid), arr(idx, ivalid)), np.int64)
            put(name, v=v, valid=np.ones(n, bool) if valid is None else valid, idx=idx,
                ivalid=np.ones(m, bool) if ivalid is None else ivalid, out=vals, out_valid=ok)
            cases.append(name)
    try:
        pc.take(pa.array([1, 2, 3]), pa.array([0, 5]))
        raise SystemExit("expected index error")
    except pa.ArrowIndexError as e:
        manifest["take_oob_message"] = str(e)
    manifest["cases"]["filter_take"] = cases


# ------------------------------------------------------------------ group-by
def reference_groupby(keys, kvalid, cols):
    """Replay of GroupBy::makeGroups + GROUPBY_AGG/GROUPBY_NUMERIC_AGG using Arrow kernels."""
    K = arr(keys, kvalid)
    enc = K.dictionary_encode(null_encoding="encode")  # first-occurrence dense ids, null = own group
    ids = np.asarray(enc.indices.to_numpy(zero_copy_only=False)).astype(np.uint32)
    uniq = enc.dictionary
    G = len(uniq)
    order = np.argsort(ids, kind="stable")  # MakeGroupings: row ids ascending within group
    counts = np.bincount(ids, minlength=G)
    offs = np.concatenate([[0], np.cumsum(counts)])
    res = {}
    for cname, (v, valid) in cols.items():
        V = arr(v, valid)
        isf = np.asarray(v).dtype == np.float64
        sums, means, mins, maxs, cnts = [], [], [], [], []
        varis, stds, prods, firsts, lasts = [], [], [], [], []
        for g in range(G):
            grp = V.take(pa.array(order[offs[g]:offs[g + 1]]))  # ApplyGroupings
            sums.append(pc.sum(grp).as_py())
            means.append(pc.mean(grp).as_py())
            mins.append(pc.min(grp).as_py())
            maxs.append(pc.max(grp).as_py())
            cnts.append(pc.count(grp).as_py())
            # the "next" aggregations of SURVEY 8(f)-3: CallFunction(name, {group}, nullptr) -> default options
            varis.append(pc.variance(grp).as_py())
            stds.append(pc.stddev(grp).as_py())
            prods.append(pc.product(grp).as_py())
            firsts.append(grp[0].as_py())            # GroupBy::first/last: GetScalar(0) / GetScalar(length - 1)
            lasts.append(grp[len(grp) - 1].as_py())
        dt = np.float64 if isf else np.int64
        nz = lambda xs, d: np.array([0 if x is None else x for x in xs], d)  # noqa: E731
        res[cname] = dict(sum=nz(sums, dt), mean=nz(means, np.float64), min=nz(mins, dt), max=nz(maxs, dt), count=np.array(cnts, np.int64),
                          ok=np.array([x is not None for x in sums], bool),
                          variance=nz(varis, np.float64), stddev=nz(stds, np.float64), product=nz(prods, dt), first=nz(firsts, dt),
                          last=nz(lasts, dt), ok_first=np.array([x is not None for x in firsts], bool),
                          ok_last=np.array([x is not None for x in lasts], bool))
    uvals, uvalid = out_np(uniq, np.int64)
    return ids, uvals, uvalid, res


def gen_groupby():
    rng = np.random.default_rng(20250104)
    cases = []
    for n in (0, 1, 17, 1000, 20000):
        for card in (1, 7, 1000):
            for nulls in (False, True):
                if n == 0 and (card != 1 or nulls):
                    continue
                name = f"gb_{n}_{card}_{int(nulls)}"
                keys = rng.integers(-card // 2, card - card // 2, n, dtype=np.int64) * 1000003
                kvalid = (rng.random(n) > 0.05) if nulls else None
                vf = f64_inputs(rng, n, "normal")
                vi = rng.integers(-10**6, 10**6, n, dtype=np.int64)
                fvalid = (rng.random(n) > 0.2) if nulls else None
                ids, uvals, uvalid, res = reference_groupby(keys, kvalid, {"f": (vf, fvalid), "i": (vi, fvalid)})
                rec = dict(keys=keys, kvalid=np.ones(n, bool) if kvalid is None else kvalid, vf=vf, vi=vi,
                           vvalid=np.ones(n, bool) if fvalid is None else fvalid, ids=ids, uniq=uvals, uniq_valid=uvalid)
                for c in ("f", "i"):
                    for k, v in res[c].items():
                        rec[f"{c}_{k}"] = v
                put(name, **rec)
                cases.append(name)
    # skewed: one hot key + long tail, and sorted keys (new key every few rows)
    for nm, keys in (("hot", np.where(rng.random(30000) < 0.9, 42, rng.integers(0, 500, 30000))),
                     ("sorted", np.repeat(np.arange(3000, dtype=np.int64), 7)),
                     ("desc", np.repeat(np.arange(2000, dtype=np.int64)[::-1], 5))):
        name = f"gb_skew_{nm}"
        keys = keys.astype(np.int64)
        vf = f64_inputs(rng, len(keys), "cancel")
        ids, uvals, uvalid, res = reference_groupby(keys, None, {"f": (vf, None)})
        rec = dict(keys=keys, vf=vf, ids=ids, uniq=uvals)
        for k, v in res["f"].items():
            rec[f"f_{k}"] = v
        put(name, **rec)
        cases.append(name)
    # large synthetic (inputs regenerated from the counter-based generator)
    n, nk = 300000, 1000
    keys, vals = orc.synth_keys(0, n, nk), orc.synth_vals(0, n, 0)
    ids, uvals, uvalid, res = reference_groupby(keys, None, {"f": (vals, None)})
    put("gb_synth_300000_1000", n=n, num_keys=nk, uniq=uvals, f_sum=res["f"]["sum"], f_mean=res["f"]["mean"], f_count=res["f"]["count"],
        f_min=res["f"]["min"], f_max=res["f"]["max"], f_variance=res["f"]["variance"], f_stddev=res["f"]["stddev"],
        f_product=res["f"]["product"], f_first=res["f"]["first"], f_last=res["f"]["last"])
    cases.append("gb_synth_300000_1000")
    manifest["cases"]["groupby"] = cases


# ------------------------------------------------------------------ resample (pandas cross-check of bin assignment)
def gen_resample():
    import pandas as pd

    rng = np.random.default_rng(20250105)
    cases = []
    t0 = 946684800 * 10**9 + 37 * 10**9 + 123  # 2000-01-01 00:00:37.000000123
    for n in (1, 2, 9, 500, 5000):
        for freq in (60 * 10**9, 7 * 10**9, 250 * 10**6):
            for closed_right in (False, True):
                for label_right in (False, True):
                    name = f"rs_{n}_{freq}_{int(closed_right)}_{int(label_right)}"
                    gaps = rng.integers(1, 3 * freq // 2, n)
                    # sprinkle exact edge hits and big gaps (empty bins)
                    ts = t0 + np.cumsum(gaps)
                    day0 = (ts[0] // (86400 * 10**9)) * 86400 * 10**9
                    if n > 4:
                        k = (ts[3] - day0) // freq + 1
                        ts[3] = day0 + k * freq  # exactly on an edge
                        ts[4:] += ts[3] - ts[4] + freq * 5 + 1 if ts[4] <= ts[3] else freq * 5
                        ts = np.sort(ts)
                    v = rng.standard_normal(n)
                    s = pd.Series(v, index=pd.to_datetime(ts, unit="ns"))
                    r = s.resample(pd.Timedelta(int(freq), unit="ns"), closed="right" if closed_right else "left",
                                   label="right" if label_right else "left", origin="start_day")
                    cnt = r.count()
                    nz = cnt[cnt > 0]
                    labels = nz.index.values.astype("datetime64[ns]").astype(np.int64)
                    counts = nz.values.astype(np.int64)
                    # arithmetic = Arrow kernels over each bin's (contiguous) rows
                    offs = np.concatenate([[0], np.cumsum(counts)])
                    V = pa.array(v)
                    means = np.array([pc.mean(V.slice(offs[i], counts[i])).as_py() for i in range(len(counts))], np.float64)
                    sums = np.array([pc.sum(V.slice(offs[i], counts[i])).as_py() for i in range(len(counts))], np.float64)
                    # the reference refuses ("upSampling is not implemented", src/resample.h:14-17,102-105) when the
                    # number of rows is smaller than the number of bins: bins.back() < labels->length()
                    put(name, ts=ts, v=v, freq=freq, closed_right=closed_right, label_right=label_right, labels=labels, counts=counts,
                        mean=means, sum=sums, nbins_total=len(cnt), upsampling=bool(n < len(cnt)))
                    cases.append(name)
    manifest["cases"]["resample"] = cases


def main():
    gen_aggregates()
    gen_elementwise()
    gen_filter_take()
    gen_groupby()
    gen_resample()
    store["manifest"] = np.array(json.dumps(manifest))
    np.savez_compressed(OUT, **store)
    print(f"wrote {OUT}: {os.path.getsize(OUT)/1e6:.2f} MB, {sum(len(v) for v in manifest['cases'].values())} cases")


if __name__ == "__main__":
    main()
