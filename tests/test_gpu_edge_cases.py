"""GPU: empty / degenerate inputs through every C-ABI entry point (the reference's tests cover empty and ragged inputs only
implicitly; Arrow semantics for them were checked with pyarrow when the expectations below were written)."""
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


def test_empty_columns_everywhere(px):
    K, L, C = px.K, px.L, px.Column
    e_f, e_i, e_b = C.from_numpy(np.zeros(0)), C.from_numpy(np.zeros(0, np.int64)), C.from_numpy(np.zeros(0, bool))
    assert K.binary(L.ADD, e_f, e_f).length == 0 and K.binary(L.DIV, e_i, 3).length == 0
    assert K.compare(L.LT, e_f, 1.0).length == 0 and K.logical(L.AND, e_b, e_b).length == 0 and K.invert(e_b).length == 0
    for kind in (L.AGG_SUM, L.AGG_MEAN, L.AGG_MIN, L.AGG_MAX):
        assert K.aggregate(kind, e_f) == (None, 0) and K.aggregate(kind, e_i) == (None, 0)  # min_count = 1 -> null
    assert K.aggregate(L.AGG_COUNT, e_f) == (0, 0)
    assert K.filter_count(e_b) == 0 and K.filter([e_f, e_i], e_b)[0].length == 0
    assert K.take([e_f], e_i)[0].length == 0
    assert K.take([C.from_numpy(np.arange(3.0))], e_i)[0].length == 0
    assert K.concat([e_f, e_f]).length == 0
    got = K.concat([e_f, C.from_numpy(np.array([1.5, 2.5])), e_f]).to_numpy()[0]
    assert list(got) == [1.5, 2.5]
    gb = K.GroupByHandle.create(e_i)
    assert gb.num_groups == 0 and gb.unique_keys().length == 0
    assert [o.length for o in gb.agg(e_f, [L.AGG_SUM, L.AGG_COUNT])] == [0, 0]
    r = K.GroupByHandle.resample(C.from_numpy(np.zeros(0, np.int64), dtype=L.TIMESTAMP_NS), 60 * 10**9)
    assert r.num_groups == 0


def test_all_null_and_single_row(px):
    K, L, C = px.K, px.L, px.Column
    v = np.array([1.0, 2.0, 3.0])
    alln = C.from_numpy(v, np.zeros(3, bool))
    assert K.aggregate(L.AGG_SUM, alln) == (None, 0) and K.aggregate(L.AGG_MAX, alln) == (None, 0) and K.aggregate(L.AGG_COUNT, alln)[0] == 0
    # group-by whose value column is entirely null: every aggregate but count is null
    gb = K.GroupByHandle.create(C.from_numpy(np.array([7, 7, 9])))
    s, c = gb.agg(alln, [L.AGG_SUM, L.AGG_COUNT])
    assert list(s.to_numpy()[1]) == [False, False] and list(c.to_numpy()[0]) == [0, 0]
    # all keys null -> one group
    gbn = K.GroupByHandle.create(C.from_numpy(np.array([5, 6, 7]), np.zeros(3, bool)))
    assert gbn.num_groups == 1 and list(gbn.unique_keys().to_numpy()[1]) == [False]
    assert gbn.agg(C.from_numpy(v), [L.AGG_SUM])[0].to_numpy()[0][0] == 6.0
    one = C.from_numpy(np.array([-0.0]))
    assert K.aggregate(L.AGG_SUM, one)[0] == 0.0 and not np.signbit(K.aggregate(L.AGG_SUM, one)[0])  # 0.0 + -0.0
    assert np.signbit(K.aggregate(L.AGG_MIN, one)[0])


@pytest.mark.parametrize("n", [63, 64, 65, 4095, 4096, 4097, 8191])
def test_tile_boundaries_groupby_and_filter(px, n):
    """row counts around the 64-row wave word and the 4096-row sort tile"""
    K, L, C = px.K, px.L, px.Column
    keys = orc.synth_keys(0, n, 37) * 1_000_003  # sparse -> hash table path
    vals = orc.synth_vals(0, n) - 0.5
    for env in ("0", "2"):
        import os
        os.environ["PDX_HASH_PARTITION"] = env
        try:
            gb = K.GroupByHandle.create(C.from_numpy(keys))
            ids, uniq, _, _ = orc.group_ids(keys)
            assert np.array_equal(gb.unique_keys().to_numpy()[0], uniq)
            got = gb.agg(C.from_numpy(vals), [L.AGG_SUM])[0].to_numpy()[0]
            assert np.array_equal(got.view(np.uint64), orc.groupby_agg(orc.AGG_SUM, ids, len(uniq), vals)[0].view(np.uint64))
        finally:
            os.environ.pop("PDX_HASH_PARTITION", None)
    mask = vals > 0
    out = K.filter([C.from_numpy(vals)], C.from_numpy(mask))[0].to_numpy()[0]
    assert np.array_equal(out, vals[mask])


# ---------------------------------------------------------------- index alignment: Series::broadcast / reindex
@pytest.mark.parametrize("dtype", ["int64", "uint64", "timestamp"])
@pytest.mark.parametrize("na,nb", [(0, 5), (1, 1), (1000, 700), (200_003, 150_001)])
def test_index_union_and_reindex_vs_oracle(px, na, nb, dtype):
    rng = np.random.default_rng(na * 7 + nb)
    lo, hi = (-3 * max(na, nb, 1), 3 * max(na, nb, 1)) if dtype != "uint64" else (0, 6 * max(na, nb, 1))
    a = rng.integers(lo, hi, na).astype(np.int64) * (10**9 if dtype == "timestamp" else 1)
    b = rng.integers(lo, hi, nb).astype(np.int64) * (10**9 if dtype == "timestamp" else 1)
    if dtype == "int64" and na > 10:
        a[:3] = [np.iinfo(np.int64).min, np.iinfo(np.int64).max, -1]  # sign handling of the 64-bit label sort
    dt = {"int64": px.L.INT64, "uint64": px.L.UINT64, "timestamp": px.L.TIMESTAMP_NS}[dtype]
    ca, cb = px.Column.from_numpy(a, dtype=dt), px.Column.from_numpy(b, dtype=dt)
    got = px.K.index_union(ca, cb).to_numpy()[0].astype(np.int64)
    exp = orc.index_union(a.view(np.uint64) if dtype == "uint64" else a, b.view(np.uint64) if dtype == "uint64" else b).astype(np.int64)
    assert np.array_equal(got, exp)
    idx, ok = px.K.reindex_indices(ca, cb).to_numpy()   # duplicates in `a`: the LAST position wins
    eidx, eok = orc.reindex_indices(a, b)
    assert np.array_equal(ok, eok) and np.array_equal(idx[eok], eidx[eok])


def test_series_binary_ops_align_unequal_indexes(px):
    """Series + Series with different indexes: union of the labels, values aligned by label, labels missing on one side -> null"""
    api = px.api
    rng = np.random.default_rng(4)
    ia = rng.permutation(5000)[:3000].astype(np.int64)
    ib = rng.permutation(5000)[:2500].astype(np.int64)
    va, vb = rng.standard_normal(3000), rng.standard_normal(2500)
    sa = api.Series(va, index=px.Column.from_numpy(ia))
    sb = api.Series(vb, index=px.Column.from_numpy(ib))
    out = sa + sb
    labels = out.index.to_numpy()[0]
    vals, ok = out.to_numpy()
    exp_labels = orc.index_union(ia, ib)
    assert np.array_equal(labels, exp_labels)
    pa_, oka = orc.reindex_indices(ia, exp_labels)
    pb_, okb = orc.reindex_indices(ib, exp_labels)
    eok = oka & okb
    assert np.array_equal(ok, eok)
    assert_bits = (va[pa_] + vb[pb_])[eok]
    assert np.array_equal(vals[eok].view(np.uint64), assert_bits.view(np.uint64))
    # equal indexes keep the fast path and the index object
    same = sa + api.Series(va * 2, index=px.Column.from_numpy(ia))
    assert same.index is sa.index and np.array_equal(same.values(), va + va * 2)
    with pytest.raises(px.L.PdxError):
        _ = sa + api.Series(vb, index=px.Column.from_numpy(ib.astype(np.uint64), dtype=px.L.UINT64))  # type(NewIndex) != type(CurrentIndex)


@pytest.mark.parametrize("na,nb", [(0, 5), (5, 0), (1000, 700), (120_003, 90_001)])
def test_index_union_unsorted_and_intersection_vs_oracle(px, na, nb):
    rng = np.random.default_rng(na + 3 * nb)
    a = rng.integers(-2 * max(na, nb, 1), 2 * max(na, nb, 1), na).astype(np.int64)   # duplicates on both sides
    b = rng.integers(-2 * max(na, nb, 1), 2 * max(na, nb, 1), nb).astype(np.int64)
    ca, cb = px.Column.from_numpy(a), px.Column.from_numpy(b)
    assert np.array_equal(px.K.index_union(ca, cb, sort=False).to_numpy()[0], orc.index_union(a, b, sort=False))
    assert np.array_equal(px.K.index_intersection(ca, cb).to_numpy()[0], orc.index_intersection(a, b))


def test_concat_columns(px):
    """pd::concat(..., AxisType::Columns): reference vectors (tests/concat_test.cpp:115-270) + frames with different indexes"""
    api = px.api
    df1 = api.DataFrame({"a": np.array([1, 2, 3]), "b": np.array([4, 5, 6])})
    df2 = api.DataFrame({"a": np.array([7.0, 8.0, 9.0]), "c": np.array([10, 11, 12])})
    r = api.concat([df1, df2], axis="columns")
    assert r.names == ["a", "b", "a", "c"] and r.index is None
    assert [list(c.to_numpy()[0]) for c in r.cols] == [[1, 2, 3], [4, 5, 6], [7.0, 8.0, 9.0], [10, 11, 12]]
    assert api.concat([df1, df2], axis="columns", ignore_index=True).names == ["0", "1", "2", "3"]
    # different indexes: outer = union in first-occurrence order (sorted on request), inner = intersection; missing labels -> null
    rng = np.random.default_rng(8)
    ia, ib = rng.permutation(4000)[:2500].astype(np.int64), rng.permutation(4000)[:3000].astype(np.int64)
    va, vb = rng.standard_normal(2500), rng.integers(0, 100, 3000).astype(np.int64)
    fa = api.DataFrame({"x": va}, index=ia)
    fb = api.DataFrame({"y": vb}, index=ib)
    for join, sort in (("outer", False), ("outer", True), ("inner", False)):
        r = api.concat([fa, fb], axis="columns", join=join, sort=sort)
        exp_idx = orc.index_intersection(ia, ib) if join == "inner" else orc.index_union(ia, ib, sort=sort)
        assert np.array_equal(r.index.to_numpy()[0], exp_idx), (join, sort)
        for col, (src_idx, src_vals) in zip(r.cols, ((ia, va), (ib, vb))):
            pos, present = orc.reindex_indices(src_idx, exp_idx)
            vals, ok = col.to_numpy()
            assert np.array_equal(np.ones(len(exp_idx), bool) if ok is None else ok, present)
            assert np.array_equal(vals[present], src_vals[pos][present])
