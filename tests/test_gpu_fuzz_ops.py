"""GPU: seeded differential fuzz of the other rows of the hot path against the oracle -- element-wise arithmetic / compare /
logical, whole-column aggregates, filter, take, resample and concat on random lengths, null patterns, NaN / inf / extreme
integers and Arrow slice offsets."""
import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu
MIN64, MAX64 = np.iinfo(np.int64).min, np.iinfo(np.int64).max


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


def _eq(got, exp, ok=None, nan_bits=True):
    got, exp = np.asarray(got), np.asarray(exp)
    if ok is None:
        ok = np.ones(len(exp), bool)
    if exp.dtype == np.float64:
        if not nan_bits:  # reductions: NaN-ness only (see test_aggregate_filter_take_concat_fuzz)
            both = np.isnan(got) & np.isnan(exp)
            ok = ok & ~both
        return np.array_equal(got.view(np.uint64)[ok], exp.view(np.uint64)[ok])
    return np.array_equal(got[ok], exp[ok])


def _same_valid(ok, eok, n):
    a = np.ones(n, bool) if ok is None else np.asarray(ok, bool)
    b = np.ones(n, bool) if eok is None else np.asarray(eok, bool)
    return np.array_equal(a, b)


def _rand_len(rng):
    return int(rng.choice([0, 1, 7, 63, 64, 65, 511, 4096, 4097, 33_333, 262_145, 1_000_003]))


def _rand_f64(rng, n):
    v = rng.standard_normal(n) * 10.0 ** rng.integers(-8, 9, n)
    for special, p in ((np.nan, 0.01), (np.inf, 0.005), (-np.inf, 0.005), (0.0, 0.02), (-0.0, 0.02)):
        v[rng.random(n) < p] = special
    return v


def _rand_i64(rng, n):
    v = rng.integers(-10**6, 10**6, n).astype(np.int64)
    for special, p in ((MIN64, 0.005), (MAX64, 0.005), (0, 0.02), (-1, 0.02)):
        v[rng.random(n) < p] = special
    return v


def _rand_valid(rng, n):
    r = rng.random()
    if r < 0.4:
        return None
    if r < 0.5:
        return np.zeros(n, bool)
    return rng.random(n) > rng.choice([0.02, 0.3, 0.9])


@pytest.mark.parametrize("seed", range(40))
def test_elementwise_fuzz(px, seed):
    rng = np.random.default_rng(1000 + seed)
    n = _rand_len(rng)
    a = _rand_f64(rng, n) if rng.random() < 0.5 else _rand_i64(rng, n)
    b = _rand_f64(rng, n) if rng.random() < 0.5 else _rand_i64(rng, n)
    va, vb = _rand_valid(rng, n), _rand_valid(rng, n)
    off = int(rng.integers(0, 9))
    A, B = px.Column.from_numpy(a, va, offset=off), px.Column.from_numpy(b, vb, offset=int(rng.integers(0, 9)))
    scalar = rng.choice([None, 3, -2.5, 0])
    for op in range(4):  # add sub mul div
        for rhs_col, rhs_np, rhs_valid, is_scalar in ((B, b, vb, False),) + (((scalar, scalar, None, True),) if scalar is not None else ()):
            try:
                exp, eok = orc.binary(op, a, rhs_np, va, rhs_valid)
            except orc.OracleError:
                with pytest.raises(px.L.PdxError):
                    px.K.binary(op, A, rhs_col, is_scalar)
                continue
            got, ok = px.K.binary(op, A, rhs_col, is_scalar).to_numpy()
            assert _same_valid(ok, eok, n) and _eq(got, exp, None if eok is None else eok), (seed, op, is_scalar)
    for op in range(6):  # eq ne lt le gt ge
        try:
            exp, eok = orc.compare(op, a, b, va, vb)
        except orc.OracleError:  # int64 against float64 with a valid value outside +-2^53: Arrow's checked promotion fails the call
            with pytest.raises(px.L.PdxError, match="not in range"):
                px.K.compare(op, A, B)
            continue
        got, ok = px.K.compare(op, A, B).to_numpy()
        assert _same_valid(ok, eok, n) and _eq(got, exp, None if eok is None else eok), (seed, "cmp", op)
    m1, m2 = rng.random(n) < 0.5, rng.random(n) < 0.5
    M1, M2 = px.Column.from_numpy(m1, va), px.Column.from_numpy(m2, vb)
    for op in range(2):
        exp, eok = orc.logical(op, m1, m2, va, vb)
        got, ok = px.K.logical(op, M1, M2).to_numpy()
        assert _same_valid(ok, eok, n) and _eq(got, exp, None if eok is None else eok), (seed, "logical", op)
    assert _eq(px.K.invert(M1).to_numpy()[0], orc.invert(m1), None if va is None else va)


@pytest.mark.parametrize("seed", range(40))
def test_aggregate_filter_take_concat_fuzz(px, seed):
    rng = np.random.default_rng(2000 + seed)
    n = _rand_len(rng)
    v = _rand_f64(rng, n) if rng.random() < 0.6 else _rand_i64(rng, n)
    if v.dtype == np.float64 and rng.random() < 0.5:
        v = np.where(np.isfinite(v), v, 1.0)  # sums of +-inf are NaN either way; keep half of the cases finite
    valid = _rand_valid(rng, n)
    off = int(rng.integers(0, 70))
    col = px.Column.from_numpy(v, valid, offset=off)
    for kind in range(5):
        ev, ecnt = orc.agg(kind, v, valid)
        gv, gcnt = px.K.aggregate(kind, col)
        if kind == 4:
            assert gv == ev, (seed, kind)
            continue
        assert gcnt == ecnt, (seed, kind)
        if ev is None:
            assert gv is None
        elif isinstance(ev, float):
            # bit for bit, NaN sign and payload included: the sum trees spell out the x86 operand rules (round 4)
            assert np.float64(gv).view(np.uint64) == np.float64(ev).view(np.uint64), (seed, kind, gv, ev)
        else:
            assert gv == ev, (seed, kind)
    # filter
    mask, mvalid = rng.random(n) < rng.choice([0.01, 0.5, 0.99]), _rand_valid(rng, n)
    M = px.Column.from_numpy(mask, mvalid, offset=int(rng.integers(0, 9)))
    for emit_null in (True, False):
        exp, eok = orc.filter(v, mask, valid, mvalid, emit_null=emit_null)
        assert px.K.filter_count(M, emit_null) == len(exp)
        got, ok = px.K.filter([col], M, emit_null=emit_null)[0].to_numpy()
        assert _same_valid(ok, eok, len(exp)) and _eq(got, exp, None if eok is None else eok), (seed, "filter", emit_null)
    # take (incl. null indices); one out-of-range index must raise like Arrow does
    if n:
        m = int(rng.integers(0, 2 * n + 2))
        idx = rng.integers(0, n, m).astype(np.int64)
        ivalid = _rand_valid(rng, m)
        exp, eok = orc.take(v, idx, valid, ivalid)
        got, ok = px.K.take([col], px.Column.from_numpy(idx, ivalid))[0].to_numpy()
        assert _same_valid(ok, eok, m) and _eq(got, exp, None if eok is None else eok), (seed, "take")
        if m:
            bad = idx.copy()
            bad[int(rng.integers(0, m))] = n + int(rng.integers(0, 5))
            with pytest.raises(px.L.PdxError, match="out of bounds"):
                px.K.take([col], px.Column.from_numpy(bad))
    # concat of ragged parts (some empty), mixed validity
    k = int(rng.integers(1, 6))
    cuts = np.sort(rng.integers(0, n + 1, k - 1)) if k > 1 else np.array([], int)
    bounds = [0, *cuts.tolist(), n]
    parts = [px.Column.from_numpy(v[a:b], None if (valid is None or rng.random() < 0.3 and valid[a:b].all()) else valid[a:b], offset=int(rng.integers(0, 5)))
             for a, b in zip(bounds[:-1], bounds[1:])]
    got, ok = px.K.concat(parts).to_numpy()
    assert _eq(got, v, valid) and _same_valid(ok, valid, n)


@pytest.mark.parametrize("seed", range(30))
def test_resample_fuzz(px, seed):
    rng = np.random.default_rng(3000 + seed)
    n = int(rng.choice([1, 2, 50, 1000, 65_537, 400_000]))
    sec = 10**9
    freq = int(rng.choice([1, 7, 60, 3600, 86400])) * sec
    span = int(freq * rng.choice([0.5, 3, 40, 1000]))
    base = 1_500_000_000 * sec + int(rng.integers(0, 86400)) * sec
    ts = base + np.sort(rng.integers(0, max(span, 1), n)).astype(np.int64)
    if rng.random() < 0.3:
        ts = base + (np.arange(n, dtype=np.int64) * (freq // 4))  # exactly on the grid every 4th row
    v = _rand_f64(rng, n) if rng.random() < 0.7 else _rand_i64(rng, n)
    if v.dtype == np.float64:
        v = np.where(np.isfinite(v), v, 2.0)
    valid = _rand_valid(rng, n)
    kw = dict(closed_right=bool(rng.random() < 0.5), label_right=bool(rng.random() < 0.5), origin=int(rng.integers(0, 5)),
              offset_ns=int(rng.choice([0, 0, 13 * sec, -5 * sec])))
    kinds = [0, 1, 2, 3, 4, 5, 8, 9]
    try:
        exp = [orc.resample_agg(k, ts, v, freq, valid=valid, **kw) for k in kinds]
    except orc.OracleError as e:
        with pytest.raises(px.L.PdxError):
            px.K.GroupByHandle.resample(px.Column.from_numpy(ts, dtype=px.L.TIMESTAMP_NS), freq, kw["closed_right"], kw["label_right"], kw["origin"], 0,
                                        kw["offset_ns"])
        return
    gb = px.K.GroupByHandle.resample(px.Column.from_numpy(ts, dtype=px.L.TIMESTAMP_NS), freq, kw["closed_right"], kw["label_right"], kw["origin"], 0,
                                     kw["offset_ns"])
    assert np.array_equal(gb.unique_keys().to_numpy()[0], exp[0][0]), (seed, kw)
    outs = gb.agg(px.Column.from_numpy(v, valid, offset=int(rng.integers(0, 9))), kinds)
    for kind, out, (_, ev, eok) in zip(kinds, outs, exp):
        got, ok = out.to_numpy()
        eok = np.asarray(eok, bool)
        assert _same_valid(ok, eok, len(ev)) and _eq(got, ev, eok, nan_bits=kind in (0, 1, 2, 3)), (seed, kind, kw)  # (sum / mean / min / max: NaN bits too)
