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
