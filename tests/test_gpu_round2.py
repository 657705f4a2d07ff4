"""GPU parity tests for the rows added in round 2 (run with `-m gpu` on an MI355X): scalar-lhs arithmetic (a1), temporal
rounding + DataFrame::downsample (a12), the remaining group-by aggregations (8(f)-3), frame-level aggregates, and the ABI's
threading contract.  Everything goes HIP kernels -> C ABI -> ctypes and is compared bit-for-bit with the golden vectors
(Arrow 25.0.0) and the CPU oracle."""
import numpy as np
import pytest

import oracle as orc
from conftest import assert_f64_bits, golden2

pytestmark = pytest.mark.gpu

G2 = golden2()
OPS = {"add": 0, "sub": 1, "mul": 2, "div": 3}
CMPS = {"eq": 0, "ne": 1, "lt": 2, "le": 3, "gt": 4, "ge": 5}


@pytest.fixture(scope="module")
def px():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from pandasarrow_amd import _lib as L
    from pandasarrow_amd import api, column

    L.check(L.load().pdx_init(0))

    class NS:
        pass

    ns = NS()
    ns.L, ns.K, ns.api, ns.Column, ns.torch = L, column, api, column.Column, torch
    return ns


def _valid_or_none(v):
    return None if v is None or np.all(v) else v


# ------------------------------------------------------------------ a1: Scalar op Series (src/scalar.cpp:24-56)
@pytest.mark.parametrize("name", G2.cases("scalar_lhs"))
@pytest.mark.parametrize("offset", [0, 5])
def test_scalar_lhs_golden(px, name, offset):
    c = G2.case(name)
    s = c["s"].item() if bool(c["s_valid"]) else None
    if s is None and c["s"].dtype == np.float64:
        S = px.Column.from_numpy(np.zeros(1), valid=np.zeros(1, bool))  # a null float64 scalar
    else:
        S = s
    B = px.Column.from_numpy(c["b"], _valid_or_none(c["vb"]), offset=offset)
    nan_exact = name.startswith("ewl_nanbits")
    side = px.L.SCALAR_LHS if isinstance(S, px.Column) else False  # a python scalar on the left is recognised by itself
    for k, op in OPS.items():
        vals, valid = px.K.binary(op, S, B, side).to_numpy()
        ev = c[f"{k}_valid"]
        if valid is not None:
            assert np.array_equal(valid, ev), f"{name} {k} validity"
        else:
            assert ev.all()
        if vals.dtype == np.float64:
            if nan_exact:  # which operand's NaN payload survives is part of the pinned behaviour
                assert np.array_equal(vals.view(np.uint64), c[k].view(np.uint64)), f"{name} {k}: NaN bits"
            assert_f64_bits(vals, c[k], valid=ev, what=f"{name} {k}", nan_bits=True)
        else:
            assert c[k].dtype == np.int64 and np.array_equal(vals[ev], c[k][ev]), f"{name} {k}"
    if "eq" not in c:
        return
    for k, op in CMPS.items():
        vals, valid = px.K.compare(op, S, B, side).to_numpy()
        ev = c[f"{k}_valid"]
        assert np.array_equal(vals[ev], c[k][ev]), f"{name} {k}"
        if valid is not None:
            assert np.array_equal(valid, ev)


def test_scalar_lhs_api_and_errors(px):
    S, Sc = px.api.Series, px.api.Scalar
    s = S(np.array([1, 2, 3, 4, 5]))
    assert list((10 - s).values()) == [9, 8, 7, 6, 5]
    assert list((Sc(10) / s).values()) == [10, 5, 3, 2, 2]
    assert list((Sc(2) * s).values()) == [2, 4, 6, 8, 10] and (Sc(2) * s).name == ""
    assert list((Sc(3) < s).values()) == [False, False, False, True, True]
    assert list((Sc(3) == s).values()) == [False, False, True, False, False]
    f = S(np.array([1.0, 4.0]))
    assert list((1 / f).values()) == [1.0, 0.25] and (1 / f).dtype() == px.L.FLOAT64
    with pytest.raises(RuntimeError, match="divide by zero"):
        7 / S(np.array([2, 0]))
    r = 7 / S(np.array([2, 0]), valid=np.array([True, False]))
    vals, valid = r.to_numpy()
    assert vals[0] == 3 and list(valid) == [True, False]
    df = px.api.DataFrame({"x": np.array([1, 2, 4]), "y": np.array([1.0, 2.0, 4.0])})
    q = 8 / df
    assert list(q["x"].values()) == [8, 4, 2] and list(q["y"].values()) == [8.0, 4.0, 2.0]
    # scalar must have length 1 at the ABI
    a2, b3 = px.Column.from_numpy(np.arange(2)), px.Column.from_numpy(np.arange(3))
    with pytest.raises(RuntimeError, match="scalar operand must have length 1"):
        px.K.binary(0, a2, b3, scalar=px.L.SCALAR_LHS)


def test_scalar_lhs_large_vs_oracle(px):
    n = 2_000_003
    b = orc.synth_vals(0, n, 2) - 0.5
    B = px.Column.from_numpy(b)
    for op in OPS.values():
        assert_f64_bits(px.K.binary(op, 0.75, B).to_numpy()[0], orc.binary(op, 0.75, b)[0], what=f"op{op}")
    for op in CMPS.values():
        assert np.array_equal(px.K.compare(op, 0.1, B).to_numpy()[0], orc.compare(op, 0.1, b)[0])


# ------------------------------------------------------------------ a12: floor/ceil_temporal + DataFrame::downsample (src/dataframe.cpp:1265-1290)
@pytest.mark.parametrize("name", G2.cases("round_temporal"))
def test_round_temporal_golden(px, name):
    c, inp = G2.case(name), G2.case("rt_input")
    for ceil, key, off in ((False, "floor", 0), (True, "ceil", 3)):
        T = px.Column.from_numpy(inp["ts"], dtype=px.L.TIMESTAMP_NS, offset=off)
        vals, valid = px.K.round_temporal(T, int(c["multiple"]), int(c["unit"]), ceil, bool(c["wsm"]), bool(c["cbo"])).to_numpy()
        assert valid is None and np.array_equal(vals, c[key]), f"{name} {key}"


def test_round_temporal_nulls_errors_large(px):
    inp, c = G2.case("rt_input"), G2.case("rt_nulls_minute_5")
    T = px.Column.from_numpy(inp["ts"], inp["valid"], dtype=px.L.TIMESTAMP_NS, offset=5)
    out = px.K.round_temporal(T, 5, px.L.UNIT_MINUTE)
    vals, valid = out.to_numpy()
    assert np.array_equal(valid, c["floor_valid"]) and np.array_equal(vals[valid], c["floor"][valid])
    for bad in (dict(multiple=0, unit=px.L.UNIT_MINUTE), dict(multiple=1, unit=11), dict(multiple=1, unit=-1)):
        with pytest.raises(RuntimeError):
            px.K.round_temporal(T, bad["multiple"], bad["unit"])
    with pytest.raises(RuntimeError, match="PDX_TIMESTAMP_NS"):
        px.K.round_temporal(px.Column.from_numpy(np.arange(4)), 1, px.L.UNIT_MINUTE)
    e = px.K.round_temporal(px.Column.from_numpy(np.zeros(0, np.int64), dtype=px.L.TIMESTAMP_NS), 3, px.L.UNIT_HOUR)
    assert e.length == 0
    # every unit / mode at 1e6 rows against the oracle (ragged tail of the 4-way unrolled loop included)
    n = 1_000_003
    rng = np.random.default_rng(11)
    ts = rng.integers(-2 * 10**18, 2 * 10**18, n)
    T = px.Column.from_numpy(ts, dtype=px.L.TIMESTAMP_NS)
    for unit in range(10):
        for mult, cbo in ((1, False), (7, False), (7, True)):
            for ceil in (False, True):
                got = px.K.round_temporal(T, mult, unit, ceil, unit % 2 == 0, cbo).to_numpy()[0]
                exp, _ = orc.round_temporal(ts, mult, unit, ceil, unit % 2 == 0, cbo)
                assert np.array_equal(got, exp), (unit, mult, cbo, ceil)
    # timestamps near the ends of int64 (the reciprocal-multiply division corrects its estimate with the exact remainder)
    far = np.concatenate([rng.integers(-91 * 10**17, -89 * 10**17, 50_000), rng.integers(89 * 10**17, 91 * 10**17, 50_000),
                          np.array([0, -1, 1, -10**9, 10**9 - 1, 59_999_999_999, 60_000_000_000, -60_000_000_001])])
    F = px.Column.from_numpy(far, dtype=px.L.TIMESTAMP_NS)
    for unit in range(0, 8):
        for mult, cbo in ((1, False), (13, False), (13, True)):
            for ceil in (False, True):
                got = px.K.round_temporal(F, mult, unit, ceil, True, cbo).to_numpy()[0]
                exp, _ = orc.round_temporal(far, mult, unit, ceil, True, cbo)
                assert np.array_equal(got, exp), ("far", unit, mult, cbo, ceil)


@pytest.mark.parametrize("name", G2.cases("downsample"))
def test_downsample_golden(px, name):
    c = G2.case(name)
    rule = G2.manifest["downsample_rules"][name]
    vf, vi = c["vf"].astype(np.float64), c["vi"].astype(np.int64)
    df = px.api.DataFrame({"f": px.api.Series(vf, valid=c["vvalid"]), "i": vi}, index=px.Column.from_numpy(c["ts"], dtype=px.L.TIMESTAMP_NS))
    r = df.downsample(rule, closed_label_right=bool(c["clr"]))
    assert np.array_equal(r.index().to_numpy()[0], c["labels"])
    if "binned" in c:
        assert np.array_equal(r.df.index.to_numpy()[0], c["binned"])
    for key, fn in (("sum", r.sum), ("mean", r.mean), ("min", r.min), ("max", r.max), ("count", r.count)):
        out = fn()
        fv, fok = out["f"].to_numpy()
        iv, _ = out["i"].to_numpy()
        if key == "count":
            assert np.array_equal(fv, c["f_count"]) and np.array_equal(iv, c["i_count"])
            continue
        assert np.array_equal(fok, c["f_ok"])
        assert_f64_bits(fv, c[f"f_{key}"], valid=c["f_ok"], what=f"{name} f {key}")
        if key == "mean":
            assert_f64_bits(iv, c["i_mean"], what=f"{name} i mean")
        else:
            assert np.array_equal(iv, c[f"i_{key}"]), f"{name} i {key}"


def test_downsample_kat(px, kat):
    """the reference's own downsample answers (tests/series_resample_test.cpp:87-165)"""
    for k in kat["downsample"]:
        cols = {nm: np.array(v, np.int64) for nm, v in k["columns"].items()}
        df = px.api.DataFrame(cols, index=px.Column.from_numpy(np.array(k["ts"], np.int64), dtype=px.L.TIMESTAMP_NS))
        r = df.downsample(k["rule"], k["closed_label_right"])
        assert list(r.index().to_numpy()[0]) == k["labels"], k["src"]
        for key in ("sum", "mean"):
            if key in k:
                out = getattr(r, key)()
                for nm, exp in k[key].items():
                    assert list(out[nm].values()) == exp, (k["src"], nm)
    df = px.api.DataFrame({"x": np.arange(3)}, index=px.Column.from_numpy(np.arange(3), dtype=px.L.TIMESTAMP_NS))
    with pytest.raises(RuntimeError, match="invalid unit got Y"):
        df.downsample("3Y")
    with pytest.raises(RuntimeError, match="timestamp"):
        px.api.DataFrame({"x": np.arange(3)}).downsample("3T")


# ------------------------------------------------------------------ 8(f)-3: all / any / count_distinct / min_max on the grouped layout
@pytest.mark.parametrize("name", G2.cases("groupby_extra"))
@pytest.mark.parametrize("offset", [0, 3])
def test_groupby_extra_golden(px, name, offset):
    c = G2.case(name)
    valid = _valid_or_none(c["vvalid"])
    L = px.L
    gb = px.K.GroupByHandle.create(px.Column.from_numpy(c["keys"]))
    G = gb.num_groups
    assert np.array_equal(gb.unique_keys().to_numpy()[0], c["uniq"])
    B = px.Column.from_numpy(c["vb"], valid, offset=offset)
    a, y = gb.agg(B, [L.AGG_ALL, L.AGG_ANY])
    (av, aok), (yv, yok) = a.to_numpy(), y.to_numpy()
    assert a.dtype == L.BOOL and a.length == G
    assert np.array_equal(aok, c["ok"]) and np.array_equal(yok, c["ok"])
    assert np.array_equal(av[c["ok"]], c["all"][c["ok"]]) and np.array_equal(yv[c["ok"]], c["any"][c["ok"]])
    assert a.null_count == int((~c["ok"]).sum())
    F = px.Column.from_numpy(c["vf"], valid, offset=offset)
    I = px.Column.from_numpy(c["vi"], valid, offset=offset)
    # count_distinct together with standard kinds of the same request (one call, the standard kinds share one grouped pass)
    cd, mn, mx = gb.agg(F, [L.AGG_COUNT_DISTINCT, L.AGG_MIN, L.AGG_MAX])
    assert cd.dtype == L.INT64 and np.array_equal(cd.to_numpy()[0], c["cd_f"])
    assert np.array_equal(gb.agg(I, [L.AGG_COUNT_DISTINCT])[0].to_numpy()[0], c["cd_i"])
    (mnv, mnok), (mxv, _) = mn.to_numpy(), mx.to_numpy()
    okm = c["ok_mm"]
    assert (mnok is None and okm.all()) or np.array_equal(mnok, okm)
    # bit-exact including the sign of tied zeros: a group WITH nulls keeps the LAST of tied maxima, one without the first
    assert np.array_equal(mnv.view(np.uint64)[okm & ~np.isnan(c["min_f"])], c["min_f"].view(np.uint64)[okm & ~np.isnan(c["min_f"])])
    assert np.array_equal(mxv.view(np.uint64)[okm & ~np.isnan(c["max_f"])], c["max_f"].view(np.uint64)[okm & ~np.isnan(c["max_f"])])
    assert np.array_equal(np.isnan(mnv[okm]), np.isnan(c["min_f"][okm]))


def test_groupby_extra_api_errors_large(px):
    L = px.L
    df = px.api.DataFrame({"k": np.array([1, 1, 3, 1, 3, 8]), "b": np.array([True, True, False, True, True, True]),
                           "v": np.array([1.0, 1.0, 2.0, 3.0, 2.0, 7.0])})
    g = df.group_by("k")
    assert list(g.all("b").values()) == [True, False, True] and list(g.any("b").values()) == [True, True, True]
    assert list(g.count_distinct("v").values()) == [2, 1, 1]
    mm = g.min_max("v")
    assert mm.names == ["min", "max"] and list(mm["min"].values()) == [1.0, 2.0, 7.0] and list(mm["max"].values()) == [3.0, 2.0, 7.0]
    mm2 = g.min_max(["v"])
    assert mm2.names == ["v_min", "v_max"]
    with pytest.raises(RuntimeError, match="all / any need PDX_BOOL"):
        g.all("v")
    with pytest.raises(RuntimeError, match="boolean values support all / any only"):
        g.sum("b")
    # large: hash-path keys, 2e6 rows, values with few distinct levels and nulls
    n = 2_000_003
    keys = orc.synth_keys(0, n, 40_000) * 7919 - 5
    rng = np.random.default_rng(9)
    vf = np.round(orc.synth_vals(0, n, 4) * 6) / 3.0
    vb = orc.synth_vals(0, n, 5) > 0.0005
    valid = rng.random(n) > 0.1
    ids, uniq, _, _ = orc.group_ids(keys)
    gb = px.K.GroupByHandle.create(px.Column.from_numpy(keys))
    cd, = gb.agg(px.Column.from_numpy(vf, valid), [L.AGG_COUNT_DISTINCT])
    assert np.array_equal(cd.to_numpy()[0], orc.groupby_count_distinct(ids, len(uniq), vf, valid))
    a, y = gb.agg(px.Column.from_numpy(vb, valid), [L.AGG_ALL, L.AGG_ANY])
    ea, ey, eok = orc.groupby_all_any(ids, len(uniq), vb, valid)
    (av, aok), (yv, _) = a.to_numpy(), y.to_numpy()
    assert np.array_equal(aok, eok) and np.array_equal(av[eok], ea[eok]) and np.array_equal(yv[eok], ey[eok])


def test_nullable_max_zero_tie_rule(px):
    """max of tied zeros: FIRST without nulls, LAST with at least one null (Arrow 25.0.0; minmax.hpp) -- whole array, grouped
    (classic nullable reducer and the fused last-digit replay), and a group without nulls inside a nullable column."""
    L = px.L
    for vals, valid, emax, emin in (([-0.0, 0.0, -5.0], None, -0.0, -5.0), ([-0.0, 0.0, -5.0], [True, True, False], 0.0, -0.0),
                                    ([0.0, -0.0, -5.0], [True, True, False], -0.0, 0.0), ([0.0, -5.0, -0.0], [True, False, True], -0.0, 0.0)):
        col = px.Column.from_numpy(np.array(vals), None if valid is None else np.array(valid))
        mx, _ = px.K.aggregate(L.AGG_MAX, col)
        mn, _ = px.K.aggregate(L.AGG_MIN, col)
        assert np.signbit(mx) == np.signbit(emax) and mx == emax and np.signbit(mn) == np.signbit(emin), (vals, valid, mx, mn)
    # grouped, sizes that take the classic path and (>= 2^22 rows, >= 2^10 slots) the fused last digit
    for n, nk in ((50_000, 300), (4_500_000, 3000)):
        rng = np.random.default_rng(n)
        keys = orc.synth_keys(0, n, nk)
        v = np.where(rng.random(n) < 0.5, -0.0, 0.0) - (rng.random(n) < 0.3) * 2.5
        valid = rng.random(n) > 0.0005  # most groups have no null at all, some have one or two
        ids, uniq, _, _ = orc.group_ids(keys)
        emx, eok = orc.groupby_agg(orc.AGG_MAX, ids, len(uniq), v, valid)
        emn, _ = orc.groupby_agg(orc.AGG_MIN, ids, len(uniq), v, valid)
        gb = px.K.GroupByHandle.create(px.Column.from_numpy(keys))
        mx, mn = gb.agg(px.Column.from_numpy(v, valid), [L.AGG_MAX, L.AGG_MIN])
        assert np.array_equal(mx.to_numpy()[0].view(np.uint64)[eok], emx.view(np.uint64)[eok]), n
        assert np.array_equal(mn.to_numpy()[0].view(np.uint64)[eok], emn.view(np.uint64)[eok]), n


# ------------------------------------------------------------------ frame-level aggregates (src/ndframe.h:329-335, src/ndframe.cpp:119-220)
@pytest.mark.parametrize("name", G2.cases("frame_aggs"))
def test_frame_aggs_golden(px, name):
    c = G2.case(name)
    k = int(c["ncols"])
    df = px.api.DataFrame({f"c{j}": px.Column.from_numpy(c[f"c{j}"], _valid_or_none(c[f"v{j}"])) for j in range(k)})
    assert df.count().value == int(c["count"])
    for j, key in enumerate(("sum", "mean", "min", "max")):
        got = getattr(df, key)().value
        if c["isnull"][j]:
            assert got is None, (name, key)
        elif c[key].dtype == np.float64:
            assert np.float64(got).view(np.uint64) == np.float64(c[key]).view(np.uint64), (name, key, got, c[key])
        else:
            assert got == int(c[key]), (name, key)


# ------------------------------------------------------------------ functions of one column: negate / abs / sign / sqrt / exp / bit_wise_not / power
def _unary_golden():
    import json
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "unary_golden.npz"))
    return z, json.loads(str(z["manifest"]))


def _ulp_distance(a_bits, b_bits):
    """distance in units in the last place between float64 bit patterns (both finite or both the same special)"""
    def key(u):
        s = u.view(np.int64)
        return np.where(s < 0, np.int64(-(2**63)) - s, s)  # monotone integer image of the floats
    with np.errstate(over="ignore"):
        return np.abs(key(np.ascontiguousarray(a_bits)) - key(np.ascontiguousarray(b_bits)))  # int64: exact for nearby values


# exp / power go through the device's libm (ocml), the reference through its host's glibc: both claim < 1 ULP, so results may differ
# in the last place or two.  Every other function is bit-exact.
LIBM_TOL_ULP = 2


@pytest.mark.parametrize("case", [c["case"] for c in _unary_golden()[1]["cases"]])
def test_unary_golden(px, case):
    z, m = _unary_golden()
    L, K = px.L, px.K
    ops = next(c["ops"] for c in m["cases"] if c["case"] == case)
    dt = str(z[case + "/dtype"])
    v, valid = z[case + "/in"].view(dt), z[case + "/valid"]
    col = px.Column.from_numpy(v, None if valid.all() else valid, dtype=L.UINT64 if dt == "uint64" else None)
    opcode = {"negate": L.NEGATE, "abs": L.ABS, "sign": L.SIGN, "sqrt": L.SQRT, "exp": L.EXP, "bit_wise_not": L.BIT_NOT}
    for op in ops:
        out = K.power(col, m["exponents"][int(op[6:])]) if op.startswith("power_") else K.unary(opcode[op], col)
        got, gok = out.to_numpy()
        assert (gok is None and valid.all()) or np.array_equal(gok, valid), (case, op)
        g, e = np.ascontiguousarray(got).view(np.uint64)[valid], z[f"{case}/{op}"][valid]
        if op == "exp" or op.startswith("power_"):
            gf, ef = g.view(np.float64), e.view(np.float64)
            assert np.array_equal(np.isnan(gf), np.isnan(ef)), (case, op)
            fin = ~np.isnan(ef)
            assert np.array_equal(np.isinf(gf[fin]), np.isinf(ef[fin])) and np.array_equal(np.sign(gf[fin]), np.sign(ef[fin])), (case, op)
            both = fin & ~np.isinf(ef)
            assert (_ulp_distance(g[both], e[both]) <= LIBM_TOL_ULP).all(), (case, op, _ulp_distance(g[both], e[both]).max())
        else:
            assert np.array_equal(g, e), (case, op)
    for err in m["errors"]:
        if err["case"] != case:
            continue
        with pytest.raises(RuntimeError) as ei:
            K.power(col, 2.0) if err["op"] == "power" else K.unary(opcode[err["op"]], col)
        if "not in range" in err["message"]:
            assert "not in range: " + err["message"].split("not in range: ")[1] in str(ei.value)


def test_unary_large_api_and_errors(px):
    L, K, api = px.L, px.K, px.api
    rng = np.random.default_rng(31)
    n = 1_000_003
    f = rng.standard_normal(n) * 10.0 ** rng.integers(-6, 7, n)
    f[rng.random(n) < 0.01] = np.nan
    fvalid = rng.random(n) > 0.1
    i = rng.integers(-(2**53), 2**53, n)
    for v, valid in ((f, None), (f, fvalid), (i, None), (i, fvalid)):
        col = px.Column.from_numpy(v, valid)
        for op, code in ((orc.UNARY_NEGATE, L.NEGATE), (orc.UNARY_ABS, L.ABS), (orc.UNARY_SIGN, L.SIGN), (orc.UNARY_SQRT, L.SQRT)):
            got, gok = K.unary(code, col).to_numpy()
            exp = orc.unary(op, v, valid)
            ok = np.ones(n, bool) if valid is None else valid
            assert np.array_equal(np.ascontiguousarray(got).view(np.uint64)[ok], np.ascontiguousarray(exp).view(np.uint64)[ok]), (v.dtype, op)
        if v.dtype == np.int64:
            got = K.unary(L.BIT_NOT, col).to_numpy()[0]
            assert np.array_equal(got, ~v)
    # Series / DataFrame mirrors
    s = api.Series(np.array([1.0, -4.0, np.nan, 9.0]), valid=np.array([1, 1, 1, 0], bool))
    assert np.array_equal((-s).to_numpy()[0][:2], [-1.0, 4.0]) and np.signbit((-s).to_numpy()[0][2])
    assert np.array_equal(s.abs().to_numpy()[0][:2], [1.0, 4.0]) and np.array_equal(s.sqrt().to_numpy()[0][:1], [1.0])
    assert np.isnan(s.sqrt().to_numpy()[0][1]) and not np.signbit(s.sqrt().to_numpy()[0][1])
    assert list(s.sign().to_numpy()[1]) == [True, True, True, False]
    assert np.allclose(s.pow(2.0).to_numpy()[0][:2], [1.0, 16.0]) and np.allclose(s.exp().to_numpy()[0][:1], [np.e])
    df = api.DataFrame({"a": np.array([1, -2, 3]), "b": np.array([4, 5, -6])})
    assert list((-df)["a"].to_numpy()[0]) == [-1, 2, -3] and list((~df)["b"].to_numpy()[0]) == [-5, -6, 5]
    assert list(df.abs()["b"].to_numpy()[0]) == [4, 5, 6] and list(df.sign()["a"].to_numpy()[0]) == [1, -1, 1]
    assert np.array_equal(df.sqrt()["a"].to_numpy()[0][[0, 2]], np.sqrt([1.0, 3.0])) and np.isnan(df.sqrt()["a"].to_numpy()[0][1])
    assert np.array_equal(df.pow(2.0)["b"].to_numpy()[0], [16.0, 25.0, 36.0])
    with pytest.raises(RuntimeError, match="bit_wise_not"):
        K.unary(L.BIT_NOT, px.Column.from_numpy(f))
    with pytest.raises(RuntimeError):
        K.unary(17, px.Column.from_numpy(f))
    with pytest.raises(RuntimeError):
        K.unary(L.ABS, px.Column.from_numpy(np.array([True, False])))
    assert K.unary(L.SQRT, px.Column.from_numpy(np.zeros(0))).length == 0


@pytest.mark.parametrize("fn", ["bit_wise_or", "bit_wise_and", "bit_wise_xor", "shift_left", "shift_right"])
def test_bitwise_and_shift_golden(px, fn):
    """BINARY_OPERATOR(| & ^ << >>) (src/series.cpp:237-245): array-array, array-scalar and scalar-array against Arrow, bit for bit"""
    z, m = _unary_golden()
    L, K, C = px.L, px.K, px.Column
    op = {"bit_wise_or": L.BIT_OR, "bit_wise_and": L.BIT_AND, "bit_wise_xor": L.BIT_XOR, "shift_left": L.SHIFT_LEFT, "shift_right": L.SHIFT_RIGHT}[fn]
    a, b, va, vb = z["bw/a"], z["bw/b"], z["bw/va"], z["bw/vb"]
    A, B = C.from_numpy(a, va, offset=3), C.from_numpy(b, vb, offset=5)
    got, ok = K.binary(op, A, B).to_numpy()
    assert np.array_equal(ok, z[f"bw/{fn}_valid"]) and np.array_equal(got.view(np.uint64)[ok], z[f"bw/{fn}"][ok])
    for j, sc in enumerate(m["bitwise_scalars"]):
        got, ok = K.binary(op, A, sc, True).to_numpy()
        assert np.array_equal(ok, va) and np.array_equal(got.view(np.uint64)[va], z[f"bw/{fn}_rhs{j}"][va]), (fn, "rhs", sc)
        got, ok = K.binary(op, sc, B).to_numpy()
        assert np.array_equal(ok, vb) and np.array_equal(got.view(np.uint64)[vb], z[f"bw/{fn}_lhs{j}"][vb]), (fn, "lhs", sc)
    # 1e6 rows against the oracle; Series operators; float operands have no kernel
    rng = np.random.default_rng(17)
    n = 1_000_001
    x, y = rng.integers(-(2**63), 2**63 - 1, n), rng.integers(-2, 66, n)
    exp, _ = orc.binary(getattr(orc, {"bit_wise_or": "BIT_OR", "bit_wise_and": "BIT_AND", "bit_wise_xor": "BIT_XOR", "shift_left": "SHIFT_LEFT",
                                      "shift_right": "SHIFT_RIGHT"}[fn]), x, y)
    assert np.array_equal(K.binary(op, C.from_numpy(x), C.from_numpy(y)).to_numpy()[0], exp)
    S = px.api.Series
    s, t = S(np.array([6, -8, 1])), S(np.array([3, 1, 62]))
    pyop = {"bit_wise_or": lambda p, q: p | q, "bit_wise_and": lambda p, q: p & q, "bit_wise_xor": lambda p, q: p ^ q,
            "shift_left": lambda p, q: p << q, "shift_right": lambda p, q: p >> q}[fn]
    assert np.array_equal(pyop(s, t).to_numpy()[0], pyop(np.array([6, -8, 1]), np.array([3, 1, 62])))
    with pytest.raises(RuntimeError, match="no kernel"):
        K.binary(op, C.from_numpy(np.array([1.5])), C.from_numpy(np.array([2])))


# ------------------------------------------------------------------ if_else + the reference's own frame tests on the device
def test_if_else_golden_and_kat(px, kat):
    z, m = _unary_golden()
    K, C, L = px.K, px.Column, px.L
    cond, cv, va, vb = z["ie/cond"], z["ie/cv"], z["ie/va"], z["ie/vb"]
    Cc = C.from_numpy(cond, cv, offset=5)
    col = lambda v, valid, off: C.from_numpy(v, valid, offset=off)
    ops = {"ii": (col(z["ie/ai"], va, 1), col(z["ie/bi"], vb, 9)), "ff": (col(z["ie/af"], va, 0), col(z["ie/bf"], vb, 3)),
           "if": (col(z["ie/ai"], va, 2), col(z["ie/bf"], vb, 0)), "fi": (col(z["ie/af"], va, 7), col(z["ie/bi"], vb, 1)),
           "i_s7": (col(z["ie/ai"], va, 0), 7), "f_snull": (col(z["ie/af"], va, 0), None), "s2.5_i": (2.5, col(z["ie/bi"], vb, 4)),
           "i_s1.5": (col(z["ie/ai"], va, 0), 1.5)}
    for name, (a, b) in ops.items():
        got, ok = K.if_else(Cc, a, b).to_numpy()
        assert np.array_equal(ok, z[f"ie/{name}_valid"]), name
        assert np.array_equal(np.ascontiguousarray(got).view(np.uint64)[ok], z[f"ie/{name}"][ok]), name
    # no nulls anywhere, 1e6 rows
    rng = np.random.default_rng(3)
    n = 1_000_003
    c, x, y = rng.random(n) > 0.3, rng.standard_normal(n), rng.integers(-9, 9, n)
    got, ok = K.if_else(C.from_numpy(c), C.from_numpy(x), C.from_numpy(y)).to_numpy()
    assert ok is None and np.array_equal(got, np.where(c, x, y.astype(np.float64)))
    S, Scalar = px.api.Series, px.api.Scalar
    for k in kat["if_else"]:
        s, mk = S(np.array(k["v"])), S(np.array(k["mask"], bool))
        assert list(s.where(mk, Scalar(k["other_scalar"])).to_numpy()[0]) == k["out"]
        assert list(s.if_else(mk, S(np.full(len(k["v"]), k["other_scalar"]))).to_numpy()[0]) == k["out"]
    with pytest.raises(RuntimeError, match="PDX_BOOL"):
        K.if_else(C.from_numpy(np.array([1, 0])), C.from_numpy(np.array([1, 2])), 3)
    with pytest.raises(RuntimeError, match="same length"):
        K.if_else(C.from_numpy(np.array([True, False])), C.from_numpy(np.array([1, 2, 3])), 3)
    assert K.if_else(C.from_numpy(np.zeros(0, bool)), C.from_numpy(np.zeros(0)), 1.0).length == 0


def test_frame_kat_device(px, kat):
    """tests/dataframe_arithmetric_test.cpp:22-310 and tests/dataframe_indexing_test.cpp:97-154 through the DataFrame mirror"""
    api = px.api
    for k in kat["frame_binary"]:
        a = api.DataFrame({c: np.array(v, np.int64 if k["a_dtype"] == "int64" else np.float64) for c, v in k["a"].items()})
        if "b" in k:
            b = api.DataFrame({c: np.array(v, np.int64 if k["b_dtype"] == "int64" else np.float64) for c, v in k["b"].items()})
            for key, res in (("add", a + b), ("sub", a - b), ("mul", a * b), ("div", a / b)):
                for c in k["a"]:
                    got = res[c].to_numpy()[0]
                    if key in k:
                        assert np.array_equal(got, np.array(k[key][c], got.dtype)), (k["src"], key, c)
                    else:
                        assert np.allclose(got, k["div_approx"][c], rtol=1e-12)
            if "mismatch_rows" in k:
                short = api.DataFrame({"colX": np.arange(k["mismatch_rows"]), "colY": np.arange(k["mismatch_rows"])})
                with pytest.raises(RuntimeError):
                    a + short
        if "series" in k:
            res = a + api.Series(np.array(k["series"]))
            for c in k["a"]:
                assert np.array_equal(res[c].to_numpy()[0], np.array(k["add"][c], np.float64))
            with pytest.raises(RuntimeError):
                a + api.Series(np.array(k["short_series"]))
        if "scalar" in k:
            res = a + api.Scalar(k["scalar"])
            for c in k["a"]:
                got = res[c].to_numpy()[0]
                assert got.dtype == np.int64 and list(got) == k["add"][c]
    for k in kat["frame_unary"]:
        df = api.DataFrame({c: np.array(v) for c, v in k["cols"].items()})
        for c in k["cols"]:
            assert list(df.abs()[c].to_numpy()[0]) == k["abs"][c] and list(df.sign()[c].to_numpy()[0]) == k["sign"][c]
            assert np.allclose(df.pow(2.0)[c].to_numpy()[0], k["pow2"][c])
            for got, exp in zip(df.sqrt()[c].to_numpy()[0], k["sqrt"][c]):
                assert np.isnan(got) if exp == "nan" else abs(got - exp) < 1e-3
            exp_ref = k["exp_approx"].get(c) or k["exp_approx"][c + "_first2"]
            assert np.allclose(df.exp()[c].to_numpy()[0][: len(exp_ref)], exp_ref, rtol=1e-4)


def test_concat_rows_kat(px, kat):
    """Concatenator::concatenateRows (tests/concat_test.cpp:591-749): schema union, int64 / double promotion, null fill, inner join"""
    api = px.api
    for k in kat["concat_rows"]:
        frames = []
        for j, cols in enumerate(k["frames"]):
            fl = set(k.get("float_cols", [[]] * len(k["frames"]))[j])
            frames.append(api.DataFrame({c: np.array(v, np.float64 if c in fl else np.int64) for c, v in cols.items()}))
        res = api.concat(frames, join=k["join"], ignore_index=bool(k.get("ignore_index")))
        assert res.names == list(k["out"]), k["src"]
        for c, exp in k["out"].items():
            vals, valid = res[c].to_numpy()
            ev = np.array([x is not None for x in exp])
            assert (valid is None and ev.all()) or np.array_equal(valid, ev), (k["src"], c)
            want = np.array([0 if x is None else x for x in exp], np.float64 if c in k.get("out_float", []) else np.int64)
            assert vals.dtype == want.dtype and np.array_equal(vals[ev], want[ev]), (k["src"], c)
        if k["index"] is None:
            assert res.index is None
        else:
            assert list(res.index.to_numpy()[0]) == k["index"]


def test_series_misc_kat(px, kat):
    """all / any / count / count_na / nunique / unique as the reference's tests pin them (tests/series_aggregation_test.cpp:12-120)"""
    S = px.api.Series
    for k in kat["series_misc"]:
        if "bool" in k:
            s = S(np.array(k["bool"], bool))
            if "all" in k:
                assert s.all() == k["all"], k["src"]
            if "any" in k:
                assert s.any() == k["any"], k["src"]
            if "non_bool_throws" in k:
                with pytest.raises(RuntimeError):
                    S(np.array(k["non_bool_throws"])).any()
            continue
        valid = np.array(k["valid"], bool)
        s = S(np.array(k["v"], np.int64), valid=None if valid.all() else valid)
        if "count" in k:
            assert s.count() == k["count"] and s.count_na() == k["count_na"], k["src"]
        if "nunique" in k:
            assert s.nunique() == k["nunique"], k["src"]
        if "unique" in k:
            assert list(s.unique().to_numpy()[0]) == k["unique"], k["src"]
    # nulls are skipped by all / any; a Series without a valid value is null (min_count = 1): the mirror raises
    assert S(np.array([True, False, True]), valid=np.array([1, 0, 1], bool)).all() is True
    assert S(np.array([False, True, False]), valid=np.array([1, 0, 1], bool)).any() is False
    with pytest.raises(RuntimeError):
        S(np.array([True, True]), valid=np.zeros(2, bool)).all()
    big = np.random.default_rng(1).integers(0, 1000, 300_000)
    assert S(big).nunique() == len(np.unique(big)) and np.array_equal(S(big).unique().to_numpy()[0], big[np.sort(np.unique(big, return_index=True)[1])])
