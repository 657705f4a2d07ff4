"""GPU: seeded differential fuzz of the group-by path against the oracle -- random sizes, cardinalities, key distributions
(uniform / zipf-like / sorted / runs), key and value nulls, value dtype and aggregate sets, through every key->slot path."""
import os

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu

ALL_KINDS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]  # sum mean min max count variance stddev product first last


@pytest.fixture(scope="module")
def px():
    from pandasarrow_amd import _lib as L
    from pandasarrow_amd import column

    L.check(L.load().pdx_init(0))

    class NS:
        pass

    ns = NS()
    ns.L, ns.K, ns.Column = L, column, column.Column
    return ns


def _bits_equal(a, b, ok):
    if a.dtype == np.float64:
        return np.array_equal(a.view(np.uint64)[ok], b.view(np.uint64)[ok])
    return np.array_equal(a[ok], b[ok])


def _make_case(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 63, 64, 65, 1000, 4095, 4097, 20_000, 70_000, 150_000, 300_000, 1_000_003, 2_500_000]))
    card = int(rng.choice([1, 2, 7, 100, 5000, 10**6]))
    shape = rng.choice(["uniform", "zipf", "sorted", "runs", "spread"])
    if shape == "uniform":
        keys = rng.integers(0, card, n)
    elif shape == "zipf":
        keys = np.minimum(rng.zipf(1.3, n), card) - 1
    elif shape == "sorted":
        keys = np.sort(rng.integers(0, card, n))
    elif shape == "runs":
        keys = np.repeat(rng.integers(0, card, n // 17 + 1), 17)[:n]
    else:  # far-apart keys: never a dense domain
        keys = rng.integers(0, card, n) * 1_000_003_019 - (1 << 61)
    keys = keys.astype(np.int64)
    kvalid = (rng.random(n) > 0.03) if rng.random() < 0.3 else None
    if rng.random() < 0.5:
        vals = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 6, n)
        if rng.random() < 0.3:
            # NaNs of both signs with distinct payloads (quiet and signalling) and a few infinities: the sum trees must hand on the bits
            # Arrow's x86 adds do (leaf: the earlier NaN, merges: the later operand's, inf + -inf: the negative default NaN)
            m = rng.random(n) < 0.01
            k = int(m.sum())
            vals[m] = (rng.integers(1, 2**51, k).astype(np.uint64) | np.uint64(0x7FF0000000000000) | (rng.integers(0, 2, k).astype(np.uint64) << np.uint64(63))
                       | (rng.integers(0, 2, k).astype(np.uint64) << np.uint64(51))).view(np.float64)
            mi = rng.random(n) < 0.002
            vals[mi] = rng.choice(np.array([np.inf, -np.inf]), int(mi.sum()))
    else:
        vals = rng.integers(-50, 50, n).astype(np.int64)
    vvalid = (rng.random(n) > rng.choice([0.02, 0.5])) if rng.random() < 0.4 else None
    kinds = [int(k) for k in rng.permutation(ALL_KINDS)[: int(rng.integers(1, 6))]]
    return keys, kvalid, vals, vvalid, kinds


@pytest.mark.parametrize("mode", ["default", "hash", "hash_global", "fused_dense", "fused_hash", "fused_side"])
@pytest.mark.parametrize("seed", range(160))
def test_groupby_fuzz(px, monkeypatch, seed, mode):
    if mode in ("hash", "hash_global", "fused_hash"):
        monkeypatch.setenv("PDX_GROUPBY_DENSE", "0")
        monkeypatch.setenv("PDX_HASH_PARTITION", "0" if mode == "hash_global" else "2")
    if mode == "fused_side":  # runs over 1500 rows count as long: a few of them -> side form, many -> classic path; either way the oracle's bits
        monkeypatch.setenv("PDX_FLR_MAX_RUN", "1500")
    if mode.startswith("fused"):  # the fused last-digit reduce at any size (product default: >= 2^22 rows and >= 2^10 runs)
        monkeypatch.setenv("PDX_FUSED_LAST_DIGIT_MIN_ROWS", "0")
        monkeypatch.setenv("PDX_FUSED_LAST_DIGIT_MIN_LOW_BITS", "4")
        monkeypatch.setenv("PDX_FUSED_LAST_DIGIT_MIN_RUN", "0")
        monkeypatch.setenv("PDX_FUSED_LAST_DIGIT_HASH", "1")
    keys, kvalid, vals, vvalid, kinds = _make_case(seed * 7919 + 13)
    gb = px.K.GroupByHandle.create(px.Column.from_numpy(keys, kvalid, offset=int(seed % 3)))
    ids, uniq, isnull, first = orc.group_ids(keys, kvalid)
    G = len(uniq)
    assert gb.num_groups == G
    assert np.array_equal(gb.group_ids().cpu().numpy().astype(np.uint32), ids)
    assert np.array_equal(gb.first_rows().cpu().numpy(), first)
    uk, uok = gb.unique_keys().to_numpy()
    assert np.array_equal(np.ones(G, bool) if uok is None else uok, ~isnull) and np.array_equal(uk[~isnull], uniq[~isnull])
    outs = gb.agg(px.Column.from_numpy(vals, vvalid, offset=int(seed % 5)), kinds)
    for kind, out in zip(kinds, outs):
        got, ok = out.to_numpy()
        exp, eok = orc.groupby_agg(kind, ids, G, vals, vvalid, nthreads=4)
        if kind in (2, 3) and vals.dtype == np.float64:  # all-NaN groups: value is NaN on both sides, compare validity only there
            pass
        assert (ok is None and eok.all()) or np.array_equal(ok, eok), (seed, mode, kind)
        assert _bits_equal(got, exp, eok), (seed, mode, kind)


@pytest.mark.parametrize("mode", ["bound", "bound_fused", "bound_hash_fused"])
@pytest.mark.parametrize("seed", range(60))
def test_groupby_fuzz_bound_columns(px, monkeypatch, seed, mode):
    """the same generator through a BOUND column (pdx_groupby_bind): the kinds one call at a time in the case's random order, then all of
    them in one call, then each once more -- layouts (fused, full, both) and cached per-group results must serve every request
    bit-identically to the oracle, whatever was asked before"""
    if mode == "bound_hash_fused":
        monkeypatch.setenv("PDX_GROUPBY_DENSE", "0")
        monkeypatch.setenv("PDX_HASH_PARTITION", "2")
    if mode.endswith("fused"):
        monkeypatch.setenv("PDX_FUSED_LAST_DIGIT_MIN_ROWS", "0")
        monkeypatch.setenv("PDX_FUSED_LAST_DIGIT_MIN_LOW_BITS", "4")
        monkeypatch.setenv("PDX_FUSED_LAST_DIGIT_MIN_RUN", "0")
        monkeypatch.setenv("PDX_FUSED_LAST_DIGIT_HASH", "1")
    keys, kvalid, vals, vvalid, kinds = _make_case(seed * 104_729 + 7)
    gb = px.K.GroupByHandle.create(px.Column.from_numpy(keys, kvalid, offset=int(seed % 3)))
    ids, uniq, isnull, first = orc.group_ids(keys, kvalid)
    G = len(uniq)
    assert gb.num_groups == G
    vcol = px.Column.from_numpy(vals, vvalid, offset=int(seed % 5))
    gb.bind(vcol)
    expected = {k: orc.groupby_agg(k, ids, G, vals, vvalid, nthreads=4) for k in set(kinds)}

    def check(kind, out, what):
        got, ok = out.to_numpy()
        exp, eok = expected[kind]
        assert (ok is None and eok.all()) or np.array_equal(ok, eok), (seed, mode, kind, what)
        assert _bits_equal(got, exp, eok), (seed, mode, kind, what)

    for k in kinds:
        check(k, gb.agg(vcol, [k])[0], "single")
        assert gb.last_plan()["bound"] == "1"
    for k, out in zip(kinds, gb.agg(vcol, kinds)):
        check(k, out, "together")
    for k in reversed(kinds):
        check(k, gb.agg(vcol, [k])[0], "again")
    other = px.Column.from_numpy(np.arange(len(keys), dtype=np.float64))   # an unbound column on the same handle is never served from the cache
    got, _ = gb.agg(other, [0])[0].to_numpy()
    exp, eok = orc.groupby_agg(0, ids, G, np.arange(len(keys), dtype=np.float64), None, nthreads=4)
    assert _bits_equal(got, exp, eok) and gb.last_plan()["bound"] == "0"


@pytest.mark.parametrize("seed", range(48))
def test_groupby_fuzz_side_form(px, monkeypatch, seed):
    """hot keys: 1-4 keys hold 1-6 % of the rows each, so their runs are far longer than the limit (lowered to 20 000 rows here) while the
    others stay below it -- the fused kernels skip those runs and the side form reduces them.  Random value dtype, nulls, kinds (variance
    included), dense / hash slots, narrow / 4-byte sort keys, bound or not; the plan must say `side`, the bits must be the oracle's."""
    rng = np.random.default_rng(seed * 1009 + 5)
    monkeypatch.setenv("PDX_FLR_MAX_RUN", "20000")
    monkeypatch.setenv("PDX_FUSED_LAST_DIGIT_MIN_ROWS", "0")
    monkeypatch.setenv("PDX_FUSED_LAST_DIGIT_MIN_RUN", "0")
    monkeypatch.setenv("PDX_FUSED_LAST_DIGIT_MIN_LOW_BITS", "4")
    hashed = seed % 4 == 3
    if hashed:
        monkeypatch.setenv("PDX_GROUPBY_DENSE", "0")
    if seed % 5 == 4:
        monkeypatch.setenv("PDX_SORT_NARROW", "0")
    n = int(rng.integers(600_000, 3_000_000))
    card = int(rng.choice([20_000, 60_000, 150_000]))
    keys = rng.integers(0, card, n).astype(np.int64)
    u = rng.random(n)
    lo = 0.0
    for h in range(int(rng.integers(1, 5))):
        share = float(rng.uniform(0.01, 0.06))
        keys[(u >= lo) & (u < lo + share)] = int(rng.integers(0, card))
        lo += share
    kvalid = (rng.random(n) > 0.02) if seed % 7 == 6 and not hashed else None
    if rng.random() < 0.6:
        vals = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 6, n)
    else:
        vals = rng.integers(-10**12, 10**12, n).astype(np.int64)
    vvalid = (rng.random(n) > rng.choice([0.03, 0.4])) if rng.random() < 0.4 else None
    kinds = [int(k) for k in rng.permutation([0, 1, 2, 3, 4, 5, 6])[: int(rng.integers(1, 5))]]
    ids, uniq, isnull, first = orc.group_ids(keys, kvalid)
    G = len(uniq)
    gb = px.K.GroupByHandle.create(px.Column.from_numpy(keys, kvalid))
    assert gb.num_groups == G
    vcol = px.Column.from_numpy(vals, vvalid, offset=int(seed % 3))
    if seed % 3 == 0:
        gb.bind(vcol)
    outs = gb.agg(vcol, kinds)
    plan = gb.last_plan()
    if os.environ.get("PDX_TEST_PRINT_PLAN"):
        print("PLAN", plan["layout"], "side=" + plan.get("side", "0"))
    if plan["layout"] == "fused":   # (a hash table with the null / INT64_MIN slots, or too few slot bits, keeps the classic path)
        assert "side" in plan, (seed, plan)
    for kind, out in zip(kinds, outs):
        got, ok = out.to_numpy()
        exp, eok = orc.groupby_agg(kind, ids, G, vals, vvalid, nthreads=4)
        assert (ok is None and eok.all()) or np.array_equal(ok, eok), (seed, kind, plan)
        assert _bits_equal(got, exp, eok), (seed, kind, plan)
