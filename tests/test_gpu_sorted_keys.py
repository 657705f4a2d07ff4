"""GPU: keys that arrive grouped (non-decreasing) take the segments path of pdx_groupby_create -- same groups, same order, same bits
as the dictionary path (checked against the oracle and against the library itself with PDX_GROUPBY_SORTED=0)."""
import os

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def px():
    from pandasarrow_amd import _lib as L
    from pandasarrow_amd import api, column

    L.check(L.load().pdx_init(0))

    class NS:
        pass

    ns = NS()
    ns.L, ns.K, ns.api, ns.Column = L, column, api, column.Column
    return ns


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64) if a.dtype.itemsize == 8 else a


KINDS = ["SUM", "MEAN", "MIN", "MAX", "COUNT", "VARIANCE", "STDDEV", "PRODUCT", "FIRST", "LAST"]


def _check_against_oracle(px, keys, vals, valid=None, dtype=None):
    K, L, C = px.K, px.L, px.Column
    kc = C.from_numpy(keys, dtype=dtype) if dtype is not None else C.from_numpy(keys)
    gb = K.GroupByHandle.create(kc)
    ids, uniq, _, first = orc.group_ids(keys.view(np.int64) if keys.dtype == np.uint64 else keys)
    G = len(uniq)
    assert gb.num_groups == G
    assert np.array_equal(_bits(gb.unique_keys().to_numpy()[0]), _bits(uniq))
    assert np.array_equal(gb.group_ids().cpu().numpy().astype(np.uint32), ids)
    assert np.array_equal(gb.first_rows().cpu().numpy(), first)
    vc = C.from_numpy(vals, valid)
    outs = gb.agg(vc, [getattr(L, "AGG_" + k) for k in KINDS])
    for k, o in zip(KINDS, outs):
        want, wok = orc.groupby_agg(getattr(orc, "AGG_" + k), ids, G, vals, valid)
        got, gok = o.to_numpy()
        if gok is None:
            gok = np.ones(G, bool)
        if k == "COUNT":
            wok = np.ones(G, bool)
        assert np.array_equal(gok, wok), k
        assert np.array_equal(_bits(got)[wok], _bits(want)[wok]), k
    return gb


def _sorted_keys(rng, n, ngroups, lo=-(10**12), hi=10**12):
    pool = np.sort(rng.choice(np.arange(lo, hi, max(1, (hi - lo) // (4 * ngroups + 1)), dtype=np.int64), size=ngroups, replace=False))
    return np.sort(pool[rng.integers(0, ngroups, n)])


@pytest.mark.parametrize("n,ngroups", [(2, 1), (2, 2), (65, 3), (4096, 4096), (4097, 17), (100_003, 1000), (1_000_003, 250_000), (3_000_001, 5)])
@pytest.mark.parametrize("isf", [True, False])
def test_sorted_int64_keys_vs_oracle(px, n, ngroups, isf):
    rng = np.random.default_rng(n * 7 + ngroups)
    keys = _sorted_keys(rng, n, min(ngroups, n))
    vals = (orc.synth_vals(3, n) - 0.5) * 1e3 if isf else rng.integers(-(10**9), 10**9, n)
    _check_against_oracle(px, keys, vals)
    valid = rng.random(n) > 0.2
    _check_against_oracle(px, keys, vals, valid)


def test_sorted_uint64_keys_crossing_the_sign_bit(px):
    rng = np.random.default_rng(5)
    n = 50_000
    pool = np.sort(np.concatenate([rng.integers(0, 2**62, 40, dtype=np.uint64), rng.integers(2**63, 2**64 - 1, 40, dtype=np.uint64)]))
    keys = np.sort(pool[rng.integers(0, 80, n)])  # unsigned order: descends once when read as int64
    assert (keys.view(np.int64)[1:] < keys.view(np.int64)[:-1]).sum() == 1
    _check_against_oracle(px, keys, orc.synth_vals(1, n), dtype=px.L.UINT64)


def test_sample_passes_but_not_sorted(px):
    """one swapped pair between the sampled positions: the counting pass finds it and the dictionary build takes over"""
    rng = np.random.default_rng(9)
    n = 1_500_000  # 65536 sampled pairs of 1.5e6: most pairs are never sampled
    keys = _sorted_keys(rng, n, 3000)
    cuts = np.flatnonzero(keys[1:] != keys[:-1])
    hit = 0
    for c in cuts:  # swap across a run boundary at an unsampled pair
        sampled = set(int(j * (n - 1) // 65536) for j in (int(c * 65536 // (n - 1)) + d for d in (-1, 0, 1, 2)) if 0 <= j < 65536)
        if int(c) not in sampled:
            keys[c], keys[c + 1] = keys[c + 1], keys[c]
            hit += 1
            if hit == 3:
                break
    assert hit == 3 and (keys[1:] < keys[:-1]).sum() == 3
    gb = _check_against_oracle(px, keys, orc.synth_vals(2, n))
    assert gb.num_groups == 3000


def test_same_bits_with_the_path_disabled(px):
    K, L, C = px.K, px.L, px.Column
    rng = np.random.default_rng(11)
    n = 700_001
    keys = _sorted_keys(rng, n, 40_000)
    vals = orc.synth_vals(4, n) * 1e6 - 5e5
    valid = rng.random(n) > 0.05
    kinds = [getattr(L, "AGG_" + k) for k in KINDS]

    def run():
        gb = K.GroupByHandle.create(C.from_numpy(keys))
        return [gb.unique_keys().to_numpy()[0], gb.group_ids().cpu().numpy(), gb.first_rows().cpu().numpy()] + [
            x for o in gb.agg(C.from_numpy(vals, valid), kinds) for x in o.to_numpy()]

    a = run()
    os.environ["PDX_GROUPBY_SORTED"] = "0"
    try:
        b = run()
    finally:
        os.environ.pop("PDX_GROUPBY_SORTED", None)
    assert len(a) == len(b)
    for j in range(0, len(a)):
        x, y = a[j], b[j]
        if x is None or y is None:
            assert x is None and y is None
            continue
        if j >= 3 and (j - 3) % 2 == 0 and a[j + 1] is not None:  # values of a nullable output: compare where valid
            ok = a[j + 1]
            assert np.array_equal(_bits(x)[ok], _bits(y)[ok]), j
        else:
            assert np.array_equal(_bits(x), _bits(y)), j


def test_sorted_keys_with_nulls_take_the_dictionary(px):
    K, L, C = px.K, px.L, px.Column
    keys = np.array([1, 1, 2, 2, 3, 3, 3], np.int64)
    kvalid = np.array([1, 0, 1, 1, 0, 1, 1], bool)
    gb = K.GroupByHandle.create(C.from_numpy(keys, kvalid))
    assert gb.num_groups == 4  # 1, null, 2, 3
    u, uok = gb.unique_keys().to_numpy()
    assert list(uok) == [True, False, True, True] and list(u[uok]) == [1, 2, 3]
    s = gb.agg(C.from_numpy(np.arange(7.0)), [L.AGG_SUM])[0].to_numpy()[0]
    assert list(s) == [0.0, 1.0 + 4.0, 2.0 + 3.0, 5.0 + 6.0]


@pytest.mark.parametrize("rule,clr,epoch", [("1T", True, True), ("5T", False, True), ("7H", True, False), ("5H", True, True), ("D", True, True),
                                            ("3D", False, False), ("3D", True, True), ("W", True, True), ("2W", False, False), ("2W", True, True),
                                            ("M", True, True), ("2M", False, True), ("Q", True, True), ("90S", False, True)])
def test_downsample_create_runs_vs_dictionary(px, rule, clr, epoch):
    """pdx_downsample_create: one pass over a sorted axis (runs of equal rounded labels) against the same call with the run path
    disabled (round_temporal + dictionary), and against the oracle's labels.  "5H" / "3D" / "2W" with a calendar origin and ceil
    step back at an origin: the labels descend, the run path gives up and the dictionary takes over inside the same call."""
    K, L, C, api = px.K, px.L, px.Column, px.api
    rng = np.random.default_rng(len(rule) * 31 + clr * 7 + epoch)
    n = 200_003
    span = {"T": 3, "S": 1, "H": 60, "D": 400, "W": 2000, "M": 4000, "Q": 9000}[rule[-1]] * 86400
    ts = np.sort(rng.integers(1_500_000_000, 1_500_000_000 + span, n)) * 10**9 + rng.integers(0, 10**9, n)
    ts.sort()
    vals = orc.synth_vals(8, n) - 0.5
    valid = rng.random(n) > 0.1
    df = api.DataFrame({"v": api.Series(vals, valid=valid), "w": rng.integers(-5, 50, n)}, index=C.from_numpy(ts, dtype=L.TIMESTAMP_NS))

    wsm = len(rule) % 2 == 0

    def run():
        r = df.downsample(rule, clr, wsm, epoch)
        out = [r.index().to_numpy()[0], None]
        for fn in (r.sum, r.mean, r.min, r.max, r.count):
            res = fn()
            out += [res["v"].to_numpy()[0], res["v"].to_numpy()[1], res["w"].to_numpy()[0]]
        return out, r.df.index.to_numpy()[0]

    (a, abinned) = run()
    os.environ["PDX_GROUPBY_SORTED"] = "0"
    try:
        (b, bbinned) = run()
    finally:
        os.environ.pop("PDX_GROUPBY_SORTED", None)
    assert np.array_equal(abinned, bbinned)
    assert len(a[0]) >= 2
    for j, (x, y) in enumerate(zip(a, b)):
        if x is None:
            continue
        if j >= 2 and (j - 2) % 3 == 0:  # float values of a nullable column: compare where valid
            ok = a[j + 1]
            assert np.array_equal(_bits(x)[ok], _bits(y)[ok]), (rule, j)
        else:
            assert np.array_equal(_bits(x), _bits(y)), (rule, j)
    # the rounded labels of the groups are the distinct values of the rounded index, in first-occurrence order
    _, first = np.unique(abinned, return_index=True)
    assert np.array_equal(a[0], abinned[np.sort(first)])


def test_downsample_create_nulls_unsorted_and_errors(px):
    K, L, C = px.K, px.L, px.Column
    rng = np.random.default_rng(3)
    n = 50_000
    ts = rng.integers(1_600_000_000, 1_600_000_000 + 30 * 86400, n) * 10**9  # unsorted
    tvalid = rng.random(n) > 0.05
    vals = orc.synth_vals(9, n)
    for valid in (None, tvalid):
        T = C.from_numpy(ts, valid, dtype=L.TIMESTAMP_NS)
        h = K.GroupByHandle.downsample(T, 6, L.UNIT_HOUR, True, True, True, -86400 * 10**9)
        binned = K.round_temporal(T, 6, L.UNIT_HOUR, True, True, True)
        bv, bok = binned.to_numpy()
        ref = K.GroupByHandle.create(binned)
        u, uok = h.unique_keys().to_numpy()
        ru, ruok = ref.unique_keys().to_numpy()
        assert np.array_equal(uok, ruok) and np.array_equal(u[uok], ru[ruok] - 86400 * 10**9)
        assert np.array_equal(h.group_ids().cpu().numpy(), ref.group_ids().cpu().numpy())
        got = h.agg(C.from_numpy(vals), [L.AGG_SUM])[0].to_numpy()[0]
        exp = ref.agg(C.from_numpy(vals), [L.AGG_SUM])[0].to_numpy()[0]
        assert np.array_equal(_bits(got), _bits(exp))
    with pytest.raises(RuntimeError):
        K.GroupByHandle.downsample(C.from_numpy(ts, dtype=L.TIMESTAMP_NS), 0, L.UNIT_HOUR)
    with pytest.raises(RuntimeError, match="PDX_TIMESTAMP_NS"):
        K.GroupByHandle.downsample(C.from_numpy(ts), 1, L.UNIT_HOUR)
    e = K.GroupByHandle.downsample(C.from_numpy(np.zeros(0, np.int64), dtype=L.TIMESTAMP_NS), 1, L.UNIT_DAY)
    assert e.num_groups == 0
    one = K.GroupByHandle.downsample(C.from_numpy(np.array([86400 * 10**9 + 5]), dtype=L.TIMESTAMP_NS), 1, L.UNIT_DAY, True)
    assert list(one.unique_keys().to_numpy()[0]) == [2 * 86400 * 10**9]


@pytest.mark.parametrize("case", ["small_range", "one_value", "uint64_top", "all_null", "all_nan", "nan_and_null", "high_bits_only", "n1", "ts_sorted"])
def test_argsort_digit_skipping_and_classes(px, case):
    """pdx_argsort's 64-bit sort skips digits that are equal in every key and places NaNs / nulls behind the numbers in row order"""
    K, C = px.K, px.Column
    rng = np.random.default_rng(len(case))
    n = 300_007
    valid = None
    if case == "small_range":
        v = rng.integers(-300, 300, n).astype(np.int64)  # two varying digits after the sign flip
    elif case == "one_value":
        v = np.full(n, -17, np.int64)
    elif case == "uint64_top":
        v = rng.integers(2**63 - 1000, 2**63 + 1000, n, dtype=np.uint64)
    elif case == "all_null":
        v, valid = rng.standard_normal(n), np.zeros(n, bool)
    elif case == "all_nan":
        v = np.full(n, np.nan)
    elif case == "nan_and_null":
        v = np.round(rng.standard_normal(n), 2)
        v[rng.random(n) < 0.3] = np.nan
        valid = rng.random(n) > 0.3
    elif case == "high_bits_only":
        v = (rng.integers(0, 50, n).astype(np.int64) << 40) - (1 << 44)
    elif case == "n1":
        v = np.array([3.5])
    else:
        v = np.sort(rng.integers(0, 10**15, n)).astype(np.int64)
        valid = rng.random(n) > 0.01
    col = C.from_numpy(v, valid) if v.dtype != np.uint64 else C.from_numpy(v, valid, dtype=px.L.UINT64)
    vi = v.view(np.int64) if v.dtype == np.uint64 else v
    for asc in (True, False):
        got, _ = K.argsort(col, asc).to_numpy()
        if v.dtype == np.uint64:  # the oracle restates Arrow on signed / float input: order the unsigned values directly
            key = v if asc else ~v
            exp = np.argsort(key, kind="stable").astype(np.uint64)
        else:
            exp = orc.argsort(vi, valid, asc)
        assert np.array_equal(got.astype(np.uint64), exp), (case, asc)


def test_extra_aggs_and_downsample_on_a_sorted_index(px):
    K, L, C, api = px.K, px.L, px.Column, px.api
    rng = np.random.default_rng(21)
    n = 300_000
    ts = np.sort(rng.integers(1_600_000_000, 1_600_000_000 + 40 * 86400, n)) * 10**9
    vals = orc.synth_vals(6, n) - 0.5
    df = api.DataFrame({"v": vals, "w": rng.integers(0, 50, n)}, index=C.from_numpy(ts, dtype=L.TIMESTAMP_NS))

    def run():
        r = df.downsample("6H", True, False, False)
        return [r.index().to_numpy()[0]] + [fn()[c].to_numpy()[0] for fn in (r.sum, r.mean, r.min, r.max, r.count) for c in ("v", "w")]

    a = run()
    os.environ["PDX_GROUPBY_SORTED"] = "0"
    try:
        b = run()
    finally:
        os.environ.pop("PDX_GROUPBY_SORTED", None)
    assert len(a[0]) > 100
    for x, y in zip(a, b):
        assert np.array_equal(_bits(x), _bits(y))
    # count_distinct / all / any run their own dictionaries on top of the handle's group ids
    keys = _sorted_keys(rng, n, 500)
    gb = K.GroupByHandle.create(C.from_numpy(keys))
    ids, uniq, _, _ = orc.group_ids(keys)
    w = rng.integers(0, 20, n)
    got = gb.agg(C.from_numpy(w), [L.AGG_COUNT_DISTINCT])[0].to_numpy()[0]
    assert np.array_equal(got, orc.groupby_count_distinct(ids, len(uniq), w))
