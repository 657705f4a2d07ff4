"""Pins the CPU oracle (oracle/pdx_oracle.c) against (a) Arrow 25.0.0 golden vectors frozen by
oracle/gen_golden.py and (b) the reference's own known-answer test vectors (kat_reference.json).
CPU only."""
import os
import numpy as np
import pytest

import oracle as orc
from conftest import assert_f64_bits, golden

G = golden()
AGG_KINDS = [orc.AGG_SUM, orc.AGG_MEAN, orc.AGG_MIN, orc.AGG_MAX]


# ------------------------------------------------------------------ aggregates
@pytest.mark.parametrize("name", [c for c in G.cases("aggregate") if c.startswith("agg_f64") and "synth" not in c])
@pytest.mark.parametrize("offset", [0, 3])
def test_agg_f64(name, offset):
    c = G.case(name)
    valid = c["valid"] if not c["valid"].all() else None
    for j, kind in enumerate(AGG_KINDS):
        val, cnt = orc.agg(kind, c["v"], valid, offset)
        assert cnt == int(c["count"])
        if c["isnull"][j]:
            assert val is None
        else:
            assert_f64_bits([val], [c["exp"][j]], what=f"{name} kind={kind}")
    assert orc.agg(orc.AGG_COUNT, c["v"], valid, offset)[0] == int(c["count"])


@pytest.mark.parametrize("name", [c for c in G.cases("aggregate") if c.startswith("agg_i64")])
def test_agg_i64(name):
    c = G.case(name)
    valid = c["valid"] if not c["valid"].all() else None
    s, cnt = orc.agg(orc.AGG_SUM, c["v"], valid)
    m, _ = orc.agg(orc.AGG_MEAN, c["v"], valid)
    lo, _ = orc.agg(orc.AGG_MIN, c["v"], valid)
    hi, _ = orc.agg(orc.AGG_MAX, c["v"], valid)
    assert cnt == int(c["count"])
    if c["isnull"][0]:
        assert s is None and m is None and lo is None and hi is None
    else:
        assert (s, lo, hi) == tuple(int(x) for x in c["exp_i"])
        assert_f64_bits([m], [c["exp_mean"]])


@pytest.mark.parametrize("name", [c for c in G.cases("aggregate") if "synth" in c])
def test_agg_f64_synth(name):
    c = G.case(name)
    v = orc.synth_vals(0, int(c["n"]), int(c["seed_off"]))
    got = [orc.agg(k, v)[0] for k in AGG_KINDS]
    assert_f64_bits(got, c["exp"], what=name)


def test_agg_kat(kat):
    for k in kat["aggregate"]:
        v = np.array(k["v"], np.int64)
        valid = np.array(k["valid"], bool)
        if "min" in k:
            assert orc.agg(orc.AGG_MIN, v, valid)[0] == k["min"]
            assert orc.agg(orc.AGG_MAX, v, valid)[0] == k["max"]
        if "mean" in k:
            assert orc.agg(orc.AGG_MEAN, v, valid)[0] == k["mean"]
        if "count" in k:
            assert orc.agg(orc.AGG_COUNT, v, valid)[0] == k["count"]


# ------------------------------------------------------------------ element-wise
OPS = {"add": orc.ADD, "sub": orc.SUB, "mul": orc.MUL, "div": orc.DIV}
CMPS = {"eq": orc.EQ, "ne": orc.NE, "lt": orc.LT, "le": orc.LE, "gt": orc.GT, "ge": orc.GE}


@pytest.mark.parametrize("name", [c for c in G.cases("elementwise") if c.startswith("ew_")])
@pytest.mark.parametrize("offset", [0, 5])
def test_elementwise(name, offset):
    c = G.case(name)
    scalar = c["b"].ndim == 0
    va = None if c["va"].all() else c["va"]
    vb = None if (scalar or c["vb"].all()) else c["vb"]
    b = c["b"].item() if scalar else c["b"]
    for k, op in OPS.items():
        vals, valid = orc.binary(op, c["a"], b, va, vb, offset)
        ev = c[f"{k}_valid"]
        if valid is not None:
            assert np.array_equal(valid, ev), f"{name} {k} validity"
        else:
            assert ev.all()
        if vals.dtype == np.float64:
            assert_f64_bits(vals, c[k], valid=ev, what=f"{name} {k}")
        else:
            assert np.array_equal(vals[ev], c[k][ev]), f"{name} {k}"
    for k, op in CMPS.items():
        vals, valid = orc.compare(op, c["a"], b, va, vb, offset)
        ev = c[f"{k}_valid"]
        assert np.array_equal(vals[ev], c[k][ev]), f"{name} {k}"
        if valid is not None:
            assert np.array_equal(valid, ev)


def test_int_divide_by_zero():
    with pytest.raises(orc.OracleError) as e:
        orc.binary(orc.DIV, np.array([7, 1]), np.array([2, 0]))
    assert str(e.value) == G.manifest["div_by_zero_message"]
    vals, valid = orc.binary(orc.DIV, np.array([7, 1]), np.array([2, 0]), None, np.array([True, False]))
    assert vals[0] == 3 and list(valid) == [True, False]


@pytest.mark.parametrize("name", [c for c in G.cases("elementwise") if c.startswith("logic_")])
def test_logical(name):
    c = G.case(name)
    for k, op in (("and_", orc.AND), ("or_", orc.OR)):
        vals, valid = orc.logical(op, c["a"], c["b"], c["va"], c["vb"], offset=3)
        ev = c[f"{k}valid"]
        assert np.array_equal(valid, ev)
        assert np.array_equal(vals[ev], c[k][ev])
    assert np.array_equal(orc.invert(c["a"], 2)[c["inv_valid"]], c["inv"][c["inv_valid"]])


def test_binary_kat(kat):
    for k in kat["binary"]:
        if k["dtype"] == "int64":
            a = np.array(k["a"], np.int64)
            b = k["b_scalar"] if "b_scalar" in k else np.array(k["b"], np.int64)
            for name, op in OPS.items():
                vals, _ = orc.binary(op, a, b)
                assert vals.dtype == np.int64 and list(vals) == k[name], (k["src"], name)
        else:
            vals, _ = orc.binary(orc.SUB, np.array(k["a"]), np.array(k["b"], np.int64))
            assert vals.dtype == np.float64 and np.allclose(vals, k["sub_approx"])


# ------------------------------------------------------------------ filter / take
@pytest.mark.parametrize("name", [c for c in G.cases("filter_take") if c.startswith("filter_")])
def test_filter(name):
    c = G.case(name)
    valid = None if c["valid"].all() else c["valid"]
    mvalid = None if c["mvalid"].all() else c["mvalid"]
    for emit, key in ((True, "emit"), (False, "drop")):
        vals, ok = orc.filter(c["v"], c["mask"], valid, mvalid, emit_null=emit, offset=0 if emit else 5)
        ev = c[f"{key}_valid"]
        assert len(vals) == len(c[key])
        if ok is not None:
            assert np.array_equal(ok, ev)
        assert_f64_bits(vals, c[key], valid=ev, what=name)


@pytest.mark.parametrize("name", [c for c in G.cases("filter_take") if c.startswith("take_")])
def test_take(name):
    c = G.case(name)
    valid = None if c["valid"].all() else c["valid"]
    ivalid = None if c["ivalid"].all() else c["ivalid"]
    vals, ok = orc.take(c["v"], c["idx"], valid, ivalid, offset=2)
    if ok is not None:
        assert np.array_equal(ok, c["out_valid"])
    assert np.array_equal(vals[c["out_valid"]], c["out"][c["out_valid"]])


def test_take_errors_and_kat(kat):
    with pytest.raises(orc.OracleError) as e:
        orc.take(np.array([1, 2, 3]), np.array([0, 5]))
    assert str(e.value) == G.manifest["take_oob_message"]
    with pytest.raises(orc.OracleError):
        orc.take(np.array([1, 2, 3]), np.array([-1]))
    for k in kat["take"]:
        vals, _ = orc.take(np.array(k["v"], np.int64), np.array(k["idx"]))
        assert list(vals) == k["out"]


# ------------------------------------------------------------------ group-by
GB_KINDS = {"sum": orc.AGG_SUM, "mean": orc.AGG_MEAN, "min": orc.AGG_MIN, "max": orc.AGG_MAX, "count": orc.AGG_COUNT}
# SURVEY 8(f)-3 ("next" aggregations on the same grouped layout)
GB_NEXT = {"variance": orc.AGG_VARIANCE, "stddev": orc.AGG_STDDEV, "product": orc.AGG_PRODUCT, "first": orc.AGG_FIRST, "last": orc.AGG_LAST}


def gb_expected_valid(c, col, k, G):
    if k == "count":
        return np.ones(G, bool)
    if k in ("first", "last"):
        return c[f"{col}_ok_{k}"]
    return c[f"{col}_ok"]


@pytest.mark.parametrize("name", [c for c in G.cases("groupby") if "synth" not in c])
def test_groupby(name):
    c = G.case(name)
    kvalid = None if ("kvalid" not in c or c["kvalid"].all()) else c["kvalid"]
    ids, uniq, isnull, first = orc.group_ids(c["keys"], kvalid, offset=1)
    assert np.array_equal(ids, c["ids"])
    if "uniq_valid" in c:
        assert np.array_equal(~isnull, c["uniq_valid"])
        assert np.array_equal(uniq[~isnull], c["uniq"][c["uniq_valid"]])
    else:
        assert np.array_equal(uniq, c["uniq"])
    # first_row really is the first occurrence
    assert all(ids[first[g]] == g and not (ids[: first[g]] == g).any() for g in range(0, len(uniq), max(1, len(uniq) // 50)))
    vvalid = None if ("vvalid" not in c or c["vvalid"].all()) else c["vvalid"]
    for col, key in (("f", "vf"), ("i", "vi")):
        if key not in c:
            continue
        for k, kind in {**GB_KINDS, **GB_NEXT}.items():
            vals, ok = orc.groupby_agg(kind, ids, len(uniq), c[key], vvalid, offset=2, nthreads=2)
            exp = c[f"{col}_{k}"]
            eok = gb_expected_valid(c, col, k, len(uniq))
            assert np.array_equal(ok, eok), f"{name} {col} {k} validity"
            if vals.dtype == np.float64:
                assert_f64_bits(vals, exp, valid=eok, what=f"{name} {col} {k}")
            else:
                assert np.array_equal(vals[eok], exp[eok]), f"{name} {col} {k}"


def test_groupby_synth_and_baseline_entry():
    c = G.case("gb_synth_300000_1000")
    n, nk = int(c["n"]), int(c["num_keys"])
    keys, vals = orc.synth_keys(0, n, nk), orc.synth_vals(0, n, 0)
    uk, s, m, cnt = orc.groupby_sum_mean_count(keys, vals, nthreads=2)
    assert np.array_equal(uk, c["uniq"]) and np.array_equal(cnt, c["f_count"])
    assert_f64_bits(s, c["f_sum"], what="sum")
    assert_f64_bits(m, c["f_mean"], what="mean")
    ids, uniq, _, _ = orc.group_ids(keys)
    for k in ("min", "max"):
        vals_k, _ = orc.groupby_agg(GB_KINDS[k], ids, len(uniq), vals)
        assert_f64_bits(vals_k, c[f"f_{k}"], what=k)


def test_groupby_kat(kat):
    for k in kat["groupby"]:
        keys = np.array(k["keys"], np.int64)
        ids, uniq, _, _ = orc.group_ids(keys)
        assert list(uniq) == k["uniques"], k["src"]
        offsets, rows = orc.groupings(ids, len(uniq))
        for cname, col in k["cols"].items():
            col = np.array(col, np.int64)
            for g, exp_rows in k.get("group_rows", {}).items():
                if cname == "age":
                    assert list(col[rows[offsets[int(g)]:offsets[int(g) + 1]]]) == exp_rows
            for agg_name, kind in GB_KINDS.items():
                if agg_name in k and cname in k[agg_name]:
                    vals, _ = orc.groupby_agg(kind, ids, len(uniq), col)
                    assert list(vals) == k[agg_name][cname], (k["src"], agg_name, cname)
        if "frame_sum" in k:
            tot = sum(orc.groupby_agg(orc.AGG_SUM, ids, len(uniq), np.array(c, np.int64))[0] for c in k["cols"].values())
            assert list(tot) == k["frame_sum"]


# ------------------------------------------------------------------ resample
@pytest.mark.parametrize("name", G.cases("resample"))
def test_resample(name):
    c = G.case(name)
    kw = dict(closed_right=bool(c["closed_right"]), label_right=bool(c["label_right"]))
    bins, _ = orc.resample_group_info(c["ts"], int(c["freq"]), **kw)
    assert len(bins) == int(c["nbins_total"])  # same bin edges as pandas (which the reference imitates)
    if bool(c["upsampling"]):  # reference quirk: fewer rows than bins -> "upSampling is not implemented."
        with pytest.raises(orc.OracleError, match="upSampling"):
            orc.resample_agg(orc.AGG_MEAN, c["ts"], c["v"], int(c["freq"]), **kw)
        return
    labels, means, ok = orc.resample_agg(orc.AGG_MEAN, c["ts"], c["v"], int(c["freq"]), **kw)
    assert np.array_equal(labels, c["labels"]), name
    assert ok.all()
    assert_f64_bits(means, c["mean"], what=name)
    _, sums, _ = orc.resample_agg(orc.AGG_SUM, c["ts"], c["v"], int(c["freq"]), **kw)
    assert_f64_bits(sums, c["sum"], what=name)
    _, counts, _ = orc.resample_agg(orc.AGG_COUNT, c["ts"], c["v"], int(c["freq"]), **kw)
    assert np.array_equal(counts, c["counts"])


def test_resample_kat(kat):
    for k in kat["resample"]:
        ts = orc.synth_ts(0, k["n"], k["t0_ns"], k["step_ns"])
        labels, sums, _ = orc.resample_agg(orc.AGG_SUM, ts, np.array(k["values"], np.int64), k["freq_ns"],
                                           closed_right=k["closed_right"], label_right=k["label_right"])
        assert list(labels) == k["labels"], k["src"]
        assert list(sums) == k["sum"], k["src"]
    for k in kat["resample_expand"]:
        out = np.zeros(len(k["row_labels"]), np.int64)
        import ctypes as C
        orc.lib().orc_resample_expand(orc._p(np.array(k["bins"], np.int64)), orc._p(np.array(k["labels"], np.int64)),
                                      C.c_int64(len(k["bins"])), orc._p(out))
        assert list(out) == k["row_labels"]


def test_concat_kat(kat):
    for k in kat["concat"]:
        vals, _ = orc.concat([np.array(p, np.int64) for p in k["parts"]])
        assert list(vals) == k["out"]
        idx, _ = orc.concat([np.arange(len(p), dtype=np.uint64) for p in k["parts"]])
        assert list(idx) == k["index"]
        w, ok = orc.concat([np.zeros(2), np.array(k["weight_parts"][1])], [np.zeros(2, bool), None])
        assert list(ok) == [False, False, True, True] and list(w[2:]) == k["weight_out"][2:]


# ------------------------------------------------------------------ sort: oracle vs Arrow's array_sort_indices
def _sort_cases():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sort_golden.npz"))
    return g, sorted({k[:-2] for k in g.files if k.endswith("_v")})


@pytest.mark.parametrize("name", _sort_cases()[1])
def test_argsort_oracle_vs_golden(name):
    g, _ = _sort_cases()
    v, valid = g[name + "_v"], g[name + "_valid"]
    assert np.array_equal(orc.argsort(v, valid, True), g[name + "_asc"])
    assert np.array_equal(orc.argsort(v, valid, False), g[name + "_desc"])
