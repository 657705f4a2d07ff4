"""Round-4 GPU parity tests: the order-free group aggregates (count / min / max / int64 sum) WITHOUT the value sort
(pandasarrow_amd/csrc/gb_acc.hpp; reference GROUPBY_AGG(min | max | sum) src/pd_core_macros.h:80-147, GROUPBY_NUMERIC_AGG(count) 5-78,
src/dataframe.cpp:1526-1534).  Every case asserts through pdx_groupby_last_plan that the accumulate path (reducer=lds_acc) ran and compares
with the oracle bit for bit (NaN bits excepted, as everywhere for reductions): every key -> slot path the accumulators support (no
partition / dense slots through pass 0 / hash-partitioned slots), values with nulls, NaN, tied zeros of both signs (Arrow keeps the first
minimum, the first maximum -- the last when the group holds a null), hot keys, null keys, bound columns, and the requests that must
keep the sorted layout."""
import zlib

import numpy as np
import pytest

import oracle as orc
from conftest import assert_f64_bits

pytestmark = pytest.mark.gpu

SUM, MEAN, MIN, MAX, COUNT, VAR, STD, PROD, FIRST, LAST = range(10)


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


def _check(kinds, outs, ids, G, vals, vvalid, what):
    for kind, out in zip(kinds, outs):
        got, ok = out.to_numpy()
        exp, eok = orc.groupby_agg(kind, ids, G, vals, vvalid, nthreads=8)
        assert ok is None or np.array_equal(ok, eok), (what, kind)
        if exp.dtype == np.float64:
            assert_f64_bits(got, exp, valid=eok, what=f"{what} kind={kind}")
        else:
            assert np.array_equal(got[eok], exp[eok]), (what, kind)


def _values(rng, n, dtype, special):
    if dtype == "i64":
        v = rng.integers(-2**62, 2**62, n).astype(np.int64)
        if special:
            v[rng.integers(0, n, n // 50)] = np.iinfo(np.int64).min
            v[rng.integers(0, n, n // 50)] = np.iinfo(np.int64).max
        return v
    v = rng.standard_normal(n)
    if special:
        m = rng.integers(0, n, n // 20)
        v[m] = rng.choice(np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 5e-324, -5e-324]), m.size)
    return v


GEOMS = {
    # name: (rows, keys, PDX_GROUPBY_DENSE, expected slots, expected layout prefix)
    "direct_small_domain": (300_000, 700, "1", "dense", "buckets:1x"),
    "dense_pass0": (700_000, 120_000, "1", "dense", "buckets:"),
    "hash_lds": (600_000, 90_000, "0", "hash_lds", "buckets:256x"),
}


@pytest.mark.parametrize("geom", list(GEOMS))
@pytest.mark.parametrize("dtype,nulls", [("f64", False), ("f64", True), ("i64", False), ("i64", True)])
def test_order_free_kinds_skip_the_sort(px, monkeypatch, geom, dtype, nulls):
    n, nk, dense, slots, layout = GEOMS[geom]
    if geom == "hash_lds" and nulls:
        pytest.skip("hash-partitioned slots + nullable values keep the sorted layout (asserted in test_requests_that_keep_the_sort)")
    monkeypatch.setenv("PDX_ACC_MIN_ROWS", "1")
    monkeypatch.setenv("PDX_GROUPBY_DENSE", dense)
    rng = np.random.default_rng(zlib.crc32(repr((geom, dtype, nulls)).encode()))
    keys = rng.integers(1000, 1000 + nk, n).astype(np.int64)
    ids, uniq, _, first = orc.group_ids(keys)
    vals = _values(rng, n, dtype, special=True)
    vvalid = (rng.random(n) > 0.07) if nulls else None
    if nulls:  # one group without any valid value, one group of NaNs only
        vvalid[ids == 3] = False
    if dtype == "f64":
        vals[ids == 5] = np.nan
    kcol, vcol = px.Column.from_numpy(keys), px.Column.from_numpy(vals, vvalid)
    gb = px.K.GroupByHandle.create(kcol)
    assert np.array_equal(gb.unique_keys().to_numpy()[0], uniq)
    for kinds in ([MIN, MAX], [COUNT], [MIN], [MAX, COUNT], [MIN, MAX, COUNT] + ([SUM] if dtype == "i64" else [])):
        outs = gb.agg(vcol, kinds)
        plan = gb.last_plan()
        assert plan["slots"] == slots and plan["layout"].startswith(layout), (kinds, plan)
        assert plan["reducer"] in ("lds_acc", "sizes_cache"), (kinds, plan)
        _check(kinds, outs, ids, len(uniq), vals, vvalid, f"{geom} {dtype} nulls={nulls} kinds={kinds}")


@pytest.mark.parametrize("geom", ["direct_small_domain", "dense_pass0", "hash_lds"])
@pytest.mark.parametrize("nulls", [False, True])
def test_tied_zeros_follow_row_order(px, monkeypatch, geom, nulls):
    """every group holds +0.0 and -0.0 in random order and nothing below / above them on one side: min and max are ties that Arrow
    settles by row order (first; the last maximum when the group has a null) -- the zero_ties pass must run and agree with the oracle"""
    n, nk, dense, slots, _ = GEOMS[geom]
    if geom == "hash_lds" and nulls:
        pytest.skip("hash-partitioned slots + nullable values keep the sorted layout")
    monkeypatch.setenv("PDX_ACC_MIN_ROWS", "1")
    monkeypatch.setenv("PDX_GROUPBY_DENSE", dense)
    rng = np.random.default_rng(77 + len(geom) + nulls)
    keys = rng.integers(0, nk, n).astype(np.int64)
    ids, uniq, _, _ = orc.group_ids(keys)
    G = len(uniq)
    zeros = rng.choice(np.array([0.0, -0.0]), n)
    side = rng.integers(0, 3, G)[ids]  # 0: zeros are the minimum (others positive), 1: the maximum (others negative), 2: only zeros
    mag = rng.random(n) + 0.5
    vals = np.where(rng.random(n) < 0.4, zeros, np.where(side == 0, mag, np.where(side == 1, -mag, zeros)))
    vvalid = None
    if nulls:
        vvalid = rng.random(n) > 0.1
        vvalid[(ids % 2) == 0] = True  # half of the groups without a null: first maximum; the others: last maximum
    kcol, vcol = px.Column.from_numpy(keys), px.Column.from_numpy(vals, vvalid)
    gb = px.K.GroupByHandle.create(kcol)
    kinds = [MIN, MAX]
    outs = gb.agg(vcol, kinds)
    plan = gb.last_plan()
    assert plan["slots"] == slots and plan["reducer"] == "lds_acc" and int(plan.get("zero_ties", "0")) > 0, plan
    _check(kinds, outs, ids, G, vals, vvalid, f"tied zeros {geom} nulls={nulls}")
    outs = gb.agg(vcol, [MAX])
    _check([MAX], outs, ids, G, vals, vvalid, f"tied zeros {geom} nulls={nulls} max alone")


def test_hot_key_and_null_keys(px, monkeypatch):
    """one key holds 40 % of the rows (its bucket spans many workgroups: the partial blocks of several segments fold), and 3 % of
    the KEYS are null (their own group)"""
    monkeypatch.setenv("PDX_ACC_MIN_ROWS", "1")
    n, nk = 1_500_000, 50_000
    rng = np.random.default_rng(5)
    keys = rng.integers(0, nk, n).astype(np.int64)
    keys[rng.random(n) < 0.4] = 4242
    kvalid = rng.random(n) > 0.03
    ids, uniq, uok, _ = orc.group_ids(keys, kvalid)
    vals = rng.standard_normal(n)
    vvalid = rng.random(n) > 0.05
    gb = px.K.GroupByHandle.create(px.Column.from_numpy(keys, kvalid))
    for vv in (None, vvalid):
        vcol = px.Column.from_numpy(vals, vv)
        kinds = [MIN, MAX, COUNT]
        outs = gb.agg(vcol, kinds)
        plan = gb.last_plan()
        assert plan["reducer"] in ("lds_acc",), plan
        _check(kinds, outs, ids, len(uniq), vals, vv, f"hot key nulls={vv is not None}")


def test_default_threshold_headline_geometry(px):
    """nothing forced: 1e6 dense keys, 2.1e7 rows (20 slot bits -> 128 buckets of 8192 accumulators: min + max share one pass), count served
    from the group sizes the second time"""
    n, nk = 21_000_000, 1_000_000
    keys = orc.synth_keys(0, n, nk)
    vals = orc.synth_vals(0, n) - 0.5
    ids, uniq, _, _ = orc.group_ids(keys)
    kcol, vcol = px.K.synth_keys(0, n, nk), px.Column.from_numpy(vals)
    gb = px.K.GroupByHandle.create(kcol)
    outs = gb.agg(vcol, [MIN, MAX])
    plan = gb.last_plan()
    assert plan == {**plan, "slots": "dense", "sort": "part:7", "layout": "buckets:128x8192", "reducer": "lds_acc", "passes": "1"}, plan
    _check([MIN, MAX], outs, ids, len(uniq), vals, None, "headline geometry min max")
    outs = gb.agg(vcol, [COUNT])
    plan = gb.last_plan()
    assert plan["sort"] == "part_keys:7" and plan["reducer"] == "lds_acc", plan
    _check([COUNT], outs, ids, len(uniq), vals, None, "headline geometry count")
    outs = gb.agg(vcol, [COUNT])
    assert gb.last_plan()["reducer"] == "sizes_cache", gb.last_plan()
    _check([COUNT], outs, ids, len(uniq), vals, None, "headline geometry count (cached sizes)")
    ivals = (vals * 2**40).astype(np.int64)
    icol = px.Column.from_numpy(ivals)
    kinds = [SUM, MIN, MAX, COUNT]
    outs = gb.agg(icol, kinds)
    plan = gb.last_plan()
    assert plan["reducer"] == "lds_acc" and plan["passes"] == "2", plan
    _check(kinds, outs, ids, len(uniq), ivals, None, "headline geometry int64 sum min max count")


def test_bound_column_keeps_order_free_results(px, monkeypatch):
    """gb.min(c); gb.max(c); gb.count(c) on a bound column: one accumulate pass, later calls are copies; a later sum() builds the sorted
    layout and everything stays consistent"""
    monkeypatch.setenv("PDX_ACC_MIN_ROWS", "1")
    n, nk = 900_000, 30_000
    rng = np.random.default_rng(11)
    keys = rng.integers(0, nk, n).astype(np.int64)
    ids, uniq, _, _ = orc.group_ids(keys)
    vals = rng.standard_normal(n)
    vvalid = rng.random(n) > 0.1
    kcol, vcol = px.Column.from_numpy(keys), px.Column.from_numpy(vals, vvalid)
    gb = px.K.GroupByHandle.create(kcol)
    gb.bind(vcol)
    o = gb.agg(vcol, [MIN])
    p = gb.last_plan()
    assert p["reducer"] == "lds_acc" and p["bound"] == "1" and p["cache"] == "fill", p
    _check([MIN], o, ids, len(uniq), vals, vvalid, "bound min")
    o = gb.agg(vcol, [MAX])
    p = gb.last_plan()
    assert p["cache"] == "hit", p
    _check([MAX], o, ids, len(uniq), vals, vvalid, "bound max (cached)")
    o = gb.agg(vcol, [COUNT])
    p = gb.last_plan()
    assert p["reducer"] == "lds_acc" and p["cache"] == "fill", p
    _check([COUNT], o, ids, len(uniq), vals, vvalid, "bound count")
    o = gb.agg(vcol, [SUM, MEAN, COUNT, MIN, MAX])
    p = gb.last_plan()
    assert p["reducer"] != "lds_acc", p
    _check([SUM, MEAN, COUNT, MIN, MAX], o, ids, len(uniq), vals, vvalid, "bound five kinds after the order-free calls")
    o = gb.agg(vcol, [MIN, COUNT])
    assert gb.last_plan()["cache"] == "hit", gb.last_plan()
    _check([MIN, COUNT], o, ids, len(uniq), vals, vvalid, "bound min count (cached)")


def test_requests_that_keep_the_sort(px, monkeypatch):
    """fp64 sum / mean, int64 mean (Arrow sums the doubles pairwise), variance, first / last are order dependent; nullable values on
    hash-partitioned slots and PDX_GROUPBY_ACC=0 keep the sorted layout too"""
    monkeypatch.setenv("PDX_ACC_MIN_ROWS", "1")
    n, nk = 400_000, 5_000
    rng = np.random.default_rng(3)
    keys = rng.integers(0, nk, n).astype(np.int64)
    ids, uniq, _, _ = orc.group_ids(keys)
    vals = rng.standard_normal(n)
    ivals = rng.integers(-1000, 1000, n).astype(np.int64)
    kcol = px.Column.from_numpy(keys)
    gb = px.K.GroupByHandle.create(kcol)
    for col, v, kinds in ((px.Column.from_numpy(vals), vals, [SUM, MIN]), (px.Column.from_numpy(vals), vals, [MEAN]),
                          (px.Column.from_numpy(ivals), ivals, [MEAN, MAX]), (px.Column.from_numpy(vals), vals, [VAR, MIN]),
                          (px.Column.from_numpy(vals), vals, [FIRST, COUNT])):
        outs = gb.agg(col, kinds)
        assert gb.last_plan()["reducer"] != "lds_acc", (kinds, gb.last_plan())
        _check(kinds, outs, ids, len(uniq), v, None, f"sorted layout kinds={kinds}")
    monkeypatch.setenv("PDX_GROUPBY_ACC", "0")
    outs = gb.agg(px.Column.from_numpy(vals), [MIN, MAX])
    assert gb.last_plan()["reducer"] != "lds_acc", gb.last_plan()
    _check([MIN, MAX], outs, ids, len(uniq), vals, None, "switch off")
    monkeypatch.delenv("PDX_GROUPBY_ACC")
    monkeypatch.setenv("PDX_GROUPBY_DENSE", "0")
    vvalid = rng.random(n) > 0.2
    gb2 = px.K.GroupByHandle.create(kcol)
    outs = gb2.agg(px.Column.from_numpy(vals, vvalid), [MIN, MAX, COUNT])
    assert gb2.last_plan()["reducer"] != "lds_acc", gb2.last_plan()
    _check([MIN, MAX, COUNT], outs, ids, len(uniq), vals, vvalid, "hash slots + nullable values")


@pytest.mark.parametrize("seed", range(24))
def test_order_free_fuzz(px, monkeypatch, seed):
    """seeded shapes through the accumulate path: sizes, cardinalities, key shapes, null densities, special values, slice offsets"""
    monkeypatch.setenv("PDX_ACC_MIN_ROWS", "1")
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([1, 17, 4095, 4097, 70_001, 300_000, 1_200_007]))
    nk = int(rng.choice([1, 3, 64, 5000, 200_000]))
    dense = str(int(rng.random() < 0.65))
    monkeypatch.setenv("PDX_GROUPBY_DENSE", dense)
    off = int(rng.choice([0, 1, 5]))
    keys = rng.integers(-nk // 2, nk - nk // 2, n + off).astype(np.int64) * (1 if dense == "1" else 1_000_003)
    dtype = "f64" if rng.random() < 0.6 else "i64"
    vals = _values(rng, n + off, dtype, special=rng.random() < 0.7)
    vvalid = (rng.random(n + off) > rng.choice([0.0, 0.02, 0.5])) if rng.random() < 0.5 else None
    kvalid = (rng.random(n + off) > 0.05) if rng.random() < 0.3 else None
    ids, uniq, _, _ = orc.group_ids(keys[off:], None if kvalid is None else kvalid[off:])
    kcol = px.Column.from_numpy(keys, kvalid).slice(off, n)
    vcol = px.Column.from_numpy(vals, vvalid).slice(off, n)
    gb = px.K.GroupByHandle.create(kcol)
    pool = [MIN, MAX, COUNT] + ([SUM] if dtype == "i64" else [])
    kinds = [k for k in pool if rng.random() < 0.6] or [MIN]
    outs = gb.agg(vcol, kinds)
    _check(kinds, outs, ids, len(uniq), vals[off:], None if vvalid is None else vvalid[off:], f"fuzz seed={seed} plan={gb.last_plan()}")


def test_mixed_type_promotion_is_arrows_checked_cast(px):
    """int64 (op) float64 promotes through Arrow's CHECKED cast (DispatchBest; pinned live by tests/cpp/arrow_bridge_test.cpp and
    tests/test_oracle_golden_r4.py): a valid int64 outside +-2^53 fails add / compare / if_else with Arrow's message; a null slot is not
    looked at; +-2^53 pass."""
    L, K = px.L, px.K
    i = np.array([1, 2**53, -2**53, 7, -5], dtype=np.int64)
    f = np.array([0.5, 1.5, 2.5, np.nan, -0.0])
    a, b = px.Column.from_numpy(i), px.Column.from_numpy(f)
    got, ok = K.binary(L.ADD, a, b).to_numpy()
    exp, _ = orc.binary(L.ADD, i, f)
    assert_f64_bits(got, exp, what="int64 + float64", nan_bits=True)
    bad = i.copy()
    bad[3] = -(2**53) - 1
    for call in (lambda c: K.binary(L.MUL, c, b), lambda c: K.binary(L.SUB, b, c), lambda c: K.compare(L.LT, c, b),
                 lambda c: K.if_else(K.compare(L.GT, b, 1.0, True), c, b), lambda c: K.binary(L.ADD, c, 2.5, True)):
        with pytest.raises(L.PdxError, match=r"Integer value -9007199254740993 not in range: -9007199254740992 to 9007199254740992") as e:
            call(px.Column.from_numpy(bad))
        assert e.value.status == L.INVALID
    valid = np.array([True, True, True, False, True])
    got, ok = K.binary(L.ADD, px.Column.from_numpy(bad, valid), b).to_numpy()
    exp, eok = orc.binary(L.ADD, bad, f, va=valid)
    assert np.array_equal(ok, eok) and np.array_equal(got[eok].view(np.uint64), exp[eok].view(np.uint64))
    # same dtypes never pay the check; int64 (op) int64 wraps as before
    got, _ = K.binary(L.ADD, px.Column.from_numpy(bad), px.Column.from_numpy(bad)).to_numpy()
    assert np.array_equal(got, bad + bad)


def test_sort_index_makes_group_results_order_independent(px):
    """DataFrame::sort_index (src/dataframe.cpp:1062-1071) in the python facade: a group-by result sorted by key is the frame a consumer
    compares with the reference's (whose Grouper numbers groups in a bounded permutation of first-occurrence order)"""
    api = px.api
    rng = np.random.default_rng(2)
    n = 50_000
    k = rng.integers(-300, 300, n).astype(np.int64)
    v = rng.standard_normal(n)
    df = api.DataFrame({"k": px.Column.from_numpy(k), "v": px.Column.from_numpy(v)})
    res = df.group_by("k").sum("v")
    by_key = api.DataFrame({"v": res.col}, index=res.index).sort_index()
    ids, uniq, _, _ = orc.group_ids(k)
    exp, _ = orc.groupby_agg(SUM, ids, len(uniq), v, None, nthreads=4)
    order = np.argsort(uniq, kind="stable")
    assert np.array_equal(by_key.index.to_numpy()[0], uniq[order])
    assert np.array_equal(by_key["v"].col.to_numpy()[0].view(np.uint64), exp[order].view(np.uint64))
    sk = res.sort_index(ascending=False)
    assert np.array_equal(sk.index.to_numpy()[0], uniq[order][::-1]) and np.array_equal(sk.col.to_numpy()[0].view(np.uint64), exp[order][::-1].view(np.uint64))
    noidx = api.DataFrame({"v": res.col}, index=res.index).sort_index(ignore_index=True)
    assert noidx.index is None


# ---------------------------------------------------------------- NaN sign / payload of the fp64 sum trees (Arrow's x86 bits, section 9e)
def _nan_golden():
    import json
    import os

    from conftest import GOLDEN_DIR

    z = np.load(os.path.join(GOLDEN_DIR, "nan_bits_golden.npz"))
    return z


def test_sum_tree_nan_bits_whole_column(px):
    """pdx_aggregate(sum / mean): dense and nullable kernels, 130 seeded arrays with quiet / signalling NaNs of both signs, +-inf and
    nulls, against Arrow C++ 25's results BIT for bit -- NaN sign and payload included"""
    import _nanbits_inputs as inp

    z = _nan_golden()
    for name, v, valid in inp.whole_cases():
        es, em, eok = z[name]
        col = px.Column.from_numpy(v, valid)
        s, _ = px.K.aggregate(SUM, col)
        m, _ = px.K.aggregate(MEAN, col)
        if not eok:
            assert s is None and m is None, name
            continue
        for got, exp, what in ((s, es, "sum"), (m, em, "mean")):
            g, e = np.float64(got).view(np.uint64), np.float64(exp).view(np.uint64)
            assert g == e, (name, what, hex(int(g)), hex(int(e)))


@pytest.mark.parametrize("path", ["default", "hash", "bound_full_layout", "no_fused"])
def test_sum_tree_nan_bits_per_group(px, monkeypatch, path):
    """per-group sum / mean through every reducer family -- fused last digit (dense + nullable replay), classic segment reducers
    (short / mid / long groups), hash slots -- with NaN bits held to Arrow's"""
    import _nanbits_inputs as inp

    if path == "hash":
        monkeypatch.setenv("PDX_GROUPBY_DENSE", "0")
    if path == "no_fused":
        monkeypatch.setenv("PDX_FUSED_LAST_DIGIT", "0")
    if path == "default":
        monkeypatch.setenv("PDX_FUSED_LAST_DIGIT_MIN_ROWS", "1000")
        monkeypatch.setenv("PDX_FUSED_LAST_DIGIT_MIN_RUN", "0")
        monkeypatch.setenv("PDX_FUSED_LAST_DIGIT_MIN_LOW_BITS", "1")
    z = _nan_golden()
    for name, keys, v, valid in inp.group_cases():
        exp = z[name]
        eok = exp[2] != 0
        kcol, vcol = px.Column.from_numpy(keys), px.Column.from_numpy(v, valid)
        gb = px.K.GroupByHandle.create(kcol)
        assert np.array_equal(gb.unique_keys().to_numpy()[0], z[name + "/uniq"]), name
        if path == "bound_full_layout":
            gb.bind(vcol)
            gb.agg(vcol, [FIRST])   # builds the fully sorted layout: the classic reducers serve the sums
        outs = gb.agg(vcol, [SUM, MEAN])
        for out, row, what in ((outs[0], 0, "sum"), (outs[1], 1, "mean")):
            got, ok = out.to_numpy()
            assert ok is None or np.array_equal(ok, eok), (name, what, path)
            g, e = got.view(np.uint64)[eok], exp[row].view(np.uint64)[eok]
            assert np.array_equal(g, e), (name, what, path, gb.last_plan(), int((g != e).sum()), [hex(int(x)) for x in g[g != e][:3]], [hex(int(x)) for x in e[g != e][:3]])


@pytest.mark.parametrize("form", ["keys_only", "rows_in_partition", "skewed_tail"])
def test_hash_build_without_row_ids(px, monkeypatch, form):
    """No null keys: the hash build's partition moves the keys alone, the groups' first rows come from positions inside the buckets
    (k_first_rows_from_pos) and the row ids are replayed only for the callers that want them (group ids per row, validity flags of
    nullable values, the skewed-bucket tail).  Arbitrary int64 keys incl. INT64_MIN (its own slot); ids / uniques / first rows and
    the aggregates against the oracle, and the old form (PDX_HASH_ROWS=1) as a cross-check of the switch."""
    monkeypatch.setenv("PDX_GROUPBY_DENSE", "0")
    if form == "rows_in_partition":
        monkeypatch.setenv("PDX_HASH_ROWS", "1")
    if form == "skewed_tail":
        monkeypatch.setenv("PDX_HASH_HEAD_ROWS", "4096")
    n, nk = 700_001, 30_000
    rng = np.random.default_rng(11)
    pool = rng.integers(-(2 ** 62), 2 ** 62, nk).astype(np.int64)
    pool[7] = np.iinfo(np.int64).min
    pool[8] = np.iinfo(np.int64).max
    keys = pool[rng.integers(0, nk, n)]
    keys[n - 5:] = np.array([3, 1, 4, 1, 5], dtype=np.int64)  # keys whose first row sits in the ragged last tile
    if form == "skewed_tail":
        keys[rng.random(n) < 0.3] = pool[99]
    ids, uniq, uok, first = orc.group_ids(keys, None)
    G = len(uniq)
    gb = px.K.GroupByHandle.create(px.Column.from_numpy(keys))
    assert np.array_equal(gb.first_rows().cpu().numpy(), first)
    ukeys, _ = gb.unique_keys().to_numpy()
    assert np.array_equal(ukeys, uniq)
    vals = rng.standard_normal(n)
    vvalid = rng.random(n) > 0.1
    for vv in (None, vvalid):  # (nullable values: the flags are looked up by original row -> the replayed row ids)
        kinds = [SUM, MEAN, COUNT, MIN, MAX]
        outs = gb.agg(px.Column.from_numpy(vals, vv), kinds)
        assert gb.last_plan()["slots"] == "hash_lds", gb.last_plan()
        _check(kinds, outs, ids, G, vals, vv, f"{form} nulls={vv is not None}")
    assert np.array_equal(gb.group_ids().cpu().numpy(), ids)
