"""CPU tests (no GPU): the oracle against the round-3 golden vectors (Arrow 25.0.0 through pyarrow, oracle/gen_golden_r3.py) and the
reference's own known answers for DataFrame::reindex (tests/dataframe_indexing_test.cpp:203-247, transcribed as data)."""
import numpy as np
import pytest

import oracle as orc
from conftest import golden3

G3 = golden3()
CMP = {"eq": 0, "ne": 1, "lt": 2, "le": 3, "gt": 4, "ge": 5}


def _v(valid):
    return None if valid.all() else valid


def _expect(z, key, ncols):
    v, ok = z[key], z[key + "_valid"]
    return np.split(v, ncols), np.split(ok, ncols)


@pytest.mark.parametrize("name", G3.cases("frame_compare"))
def test_frame_compare_oracle(name):
    """BINARY_OPERATOR_DF(> >= < <= == !=) (src/dataframe.cpp:563-573): one compare kernel over the frame's ChunkedArray == column by column"""
    z = G3.case(name)
    ncols = 3
    for short, op in CMP.items():
        for tag in ("frame", "series"):
            ev, eok = _expect(z, f"{short}_{tag}", ncols)
            for c in range(ncols):
                b, bv = (z[f"b{c}"], z[f"b{c}_valid"]) if tag == "frame" else (z["s"], z["s_valid"])
                got, ok = orc.compare(op, z[f"a{c}"], b, _v(z[f"a{c}_valid"]), _v(bv))
                ok = np.ones(len(got), bool) if ok is None else ok
                assert np.array_equal(ok, eok[c]), (name, short, tag, c)
                assert np.array_equal(got[ok], ev[c][ok]), (name, short, tag, c)
        for si in range(len(z["scalars"])):
            ev, eok = _expect(z, f"{short}_scalar{si}", ncols)
            is_f = z["a0"].dtype == np.float64
            sc = float(z["scalars"][si]) if is_f else int(z["scalars"][si])
            for c in range(ncols):
                got, ok = orc.compare(op, z[f"a{c}"], sc, _v(z[f"a{c}_valid"]), None if z["scalars_valid"][si] else np.array([False]))
                ok = np.ones(len(got), bool) if ok is None else ok
                assert np.array_equal(ok, eok[c]), (name, short, si, c)
                assert np.array_equal(got[ok], ev[c][ok]), (name, short, si, c)


@pytest.mark.parametrize("name", G3.cases("frame_logical"))
def test_frame_logical_oracle(name):
    """BINARY_OPERATOR_DF(&&, and) / (||, or) (src/dataframe.cpp:575-577): Arrow's non-Kleene kernels; a scalar is broadcast"""
    z = G3.case(name)
    ncols = 2
    n = len(z["a0"])
    for short, op in (("and", 0), ("or", 1)):
        for tag in ("frame", "series"):
            ev, eok = _expect(z, f"{short}_{tag}", ncols)
            for c in range(ncols):
                b, bv = (z[f"b{c}"], z[f"b{c}_valid"]) if tag == "frame" else (z["s"], z["s_valid"])
                got, ok = orc.logical(op, z[f"a{c}"], b, _v(z[f"a{c}_valid"]), _v(bv))
                ok = np.ones(n, bool) if ok is None else ok
                assert np.array_equal(ok, eok[c]) and np.array_equal(got[ok], ev[c][ok]), (name, short, tag, c)
        for si, (sv, sok) in enumerate(((True, True), (False, True), (False, False))):
            ev, eok = _expect(z, f"{short}_scalar{si}", ncols)
            for c in range(ncols):
                got, ok = orc.logical(op, z[f"a{c}"], np.full(n, sv), _v(z[f"a{c}_valid"]), None if sok else np.zeros(n, bool))
                ok = np.ones(n, bool) if ok is None else ok
                assert np.array_equal(ok, eok[c]) and np.array_equal(got[ok], ev[c][ok]), (name, short, si, c)


@pytest.mark.parametrize("name", G3.cases("reindex_fill"))
def test_reindex_fill_oracle(name):
    """Series::reindex(newIndex, fillValue) (src/series.cpp:1255-1309) = take at the LAST position, absent labels -> fill | null"""
    z = G3.case(name)
    for tag, fill in (("null", None), ("fill", z["fill"][0])):
        got, ok = orc.reindex(z["values"], z["values_valid"], z["old_index"], z["new_index"], fill)
        assert np.array_equal(ok, z[f"out_{tag}_valid"]), (name, tag)
        assert np.array_equal(got[ok].view(np.uint64), z[f"out_{tag}"][ok].view(np.uint64)), (name, tag)


def test_frame_reindex_kat_oracle(kat):
    for k in kat["frame_reindex"]:
        for c, v in k["cols"].items():
            valid = np.array([x is not None for x in v])
            vals = np.array([0 if x is None else x for x in v], np.int64 if k["col_dtype"] == "int64" else np.float64)
            got, ok = orc.reindex(vals, valid, np.array(k["index"], np.int64), np.array(k["new_index"], np.int64), None)
            assert list(ok.astype(int)) == k["out_valid"][c], k["src"]
            assert [x for x, o in zip(got, ok) if o] == [x for x, o in zip(k["out"][c], k["out_valid"][c]) if o], k["src"]


# ------------------------------------------------------------------ the documented deviation: result ROW ORDER on large inputs
def _order_cases():
    import json
    import os

    from conftest import GOLDEN_DIR

    z = np.load(os.path.join(GOLDEN_DIR, "group_order_arrow25.npz"))
    return z, json.loads(str(z["manifest"]))["cases"]


def order_case_keys(info):
    return np.random.default_rng(info["seed"]).integers(0, info["keys"], info["rows"]).astype(np.int64)


@pytest.mark.parametrize("name", ["rows1e5_keys5e4", "rows2e6_keys1e4"])
def test_group_order_deviation_is_exactly_as_documented(name):
    """Arrow 25's Grouper (reference: src/dataframe.cpp:1580-1591) orders groups only approximately by first occurrence when many new
    keys meet in one mini-batch of its swiss table; this backend defines FIRST-OCCURRENCE order (include/pdx/abi.h at
    pdx_groupby_create).  Frozen from Arrow 25.0.0 (oracle/gen_golden_order.py): the key SETS are equal, the number of positions that
    differ is the recorded one -- if Arrow's order ever became reproducible here, this test is the place that would notice."""
    z, cases = _order_cases()
    info = cases[name]
    keys = order_case_keys(info)
    ids, uniq, isnull, first = orc.group_ids(keys)
    assert len(uniq) == info["groups"] and not isnull.any()
    assert np.array_equal(first, np.sort(first)) and np.array_equal(uniq, keys[first])      # ours: first occurrence, exactly
    arrow_order = z[f"{name}/arrow_order"]
    assert np.array_equal(np.sort(arrow_order), np.sort(uniq))                               # same groups
    assert int((arrow_order != uniq).sum()) == info["positions_that_differ"] > 0             # different row order, as documented
