"""CPU: round-4 pinning of the oracle against Arrow C++ 25 (the pyarrow wheel; skipped where it is absent).

1. The SHAPE of the one standing deviation from the reference: Arrow's Grouper::Consume does not hand out group ids in exact
   first-occurrence order on inputs with many new keys per mini-batch.  What holds -- and is asserted here against live Arrow on inputs
   spanning many mini-batches -- is: Consume walks its batch in mini-batches of 128, 256, 512 and then 1024 rows; the keys FIRST SEEN in
   a mini-batch receive one contiguous block of ids, the same block first-occurrence numbering gives them, permuted inside the block.
   So (a) the set of groups, every per-key result and the id block of every mini-batch agree with this backend, (b) rows of a result
   can differ from the reference's only by such a bounded permutation; sort_by_key in both facades gives an order-independent frame.
2. The implicit int64 -> float64 promotion of mixed-type arithmetic / comparisons is Arrow's checked cast (valid values outside +-2^53
   fail the call): the oracle restates it, pyarrow confirms."""
import numpy as np
import pytest

import oracle as orc

pa = pytest.importorskip("pyarrow")


def _minibatch_of(rows):
    """index of the Grouper mini-batch a row falls into: sizes 128, 256, 512, then 1024 each"""
    rows = np.asarray(rows, np.int64)
    out = np.empty(len(rows), np.int64)
    small = rows < 896
    out[small] = np.searchsorted(np.array([128, 384, 896]), rows[small], side="right")
    out[~small] = 3 + (rows[~small] - 896) // 1024
    return out


@pytest.mark.parametrize("n,nk,seed", [(17, 16, 1), (100_000, 97, 2), (100_000, 50_000, 1), (2_000_000, 10_000, 1), (5_000_000, 1_000_000, 3)])
def test_arrow_group_order_is_first_occurrence_up_to_a_permutation_inside_each_minibatch(n, nk, seed):
    try:
        keys, aid = orc.arrow_order_run(n, nk, seed)
    except (ImportError, FileNotFoundError, StopIteration, IndexError) as e:
        pytest.skip(f"Arrow C++ (pyarrow wheel headers / libarrow) not available: {e}")
    ids, uniq, _, first = orc.group_ids(keys)
    G = len(uniq)
    assert int(aid.max()) + 1 == G
    # the same partition of the rows: one Arrow id per oracle id and vice versa
    pair = np.unique(np.stack([ids.astype(np.int64), aid.astype(np.int64)]), axis=1)
    assert pair.shape[1] == G
    to_arrow = np.empty(G, np.int64)
    to_arrow[pair[0]] = pair[1]          # first-occurrence id -> Arrow id
    mb = _minibatch_of(first)            # mini-batch in which each group (first-occurrence order) first appears: non-decreasing
    assert np.all(np.diff(mb) >= 0)
    # Arrow's id of a group lies in the id block of that group's mini-batch: [first id of the block, last id of the block]
    starts = np.flatnonzero(np.r_[True, mb[1:] != mb[:-1]])
    ends = np.r_[starts[1:], G] - 1
    blk = np.searchsorted(starts, np.arange(G), side="right") - 1
    assert np.all((to_arrow >= starts[blk]) & (to_arrow <= ends[blk])), "an Arrow id left its mini-batch's block"
    moved = int((to_arrow != np.arange(G)).sum())
    if nk <= 16:
        assert moved == 0     # few new keys per mini-batch (the reference's own tests): exactly first occurrence
    print(f"n={n} keys={nk}: {moved} of {G} group positions differ from first occurrence, all inside their mini-batch's block")


def test_mixed_type_promotion_is_a_checked_cast():
    import pyarrow.compute as pc

    big = np.array([1, 2**53, -2**53, 7], dtype=np.int64)
    f = np.array([0.5, 1.5, 2.5, 3.5])
    v, ok = orc.binary(0, big, f)
    assert ok is None and np.array_equal(v, pc.add(pa.array(big), pa.array(f)).to_numpy())
    bad = big.copy()
    bad[2] = 2**53 + 1
    with pytest.raises(orc.OracleError, match="not in range: -9007199254740992 to 9007199254740992"):
        orc.binary(0, bad, f)
    with pytest.raises(pa.ArrowInvalid, match="not in range"):
        pc.add(pa.array(bad), pa.array(f))
    with pytest.raises(orc.OracleError):
        orc.compare(4, f, bad)
    with pytest.raises(pa.ArrowInvalid):
        pc.greater(pa.array(f), pa.array(bad))
    # a null slot is not looked at
    valid = np.array([True, True, False, True])
    v, ok = orc.binary(0, bad, f, va=valid)
    want = pc.add(pa.array(bad, mask=~valid), pa.array(f))
    assert np.array_equal(ok, valid) and np.array_equal(v[valid], want.to_numpy(zero_copy_only=False)[valid])


# ---------------------------------------------------------------- NaN sign / payload of the fp64 sum trees
def _nan_golden():
    import json
    import os

    from conftest import GOLDEN_DIR

    z = np.load(os.path.join(GOLDEN_DIR, "nan_bits_golden.npz"))
    return z, json.loads(str(z["manifest"]))["cases"]


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_sum_tree_nan_bits_whole_column():
    """Arrow's pairwise sum hands NaNs on the way its compiled SSE adds do (leaf: the earlier NaN, merges: the later operand's, inf + -inf:
    the negative default NaN); the oracle spells that rule out.  130 seeded arrays with quiet / signalling NaNs of both signs, +-inf,
    nulls: sum and mean BIT for bit, NaN bits included."""
    import _nanbits_inputs as inp

    z, cases = _nan_golden()
    for name, v, valid in inp.whole_cases():
        es, em, eok = z[name]
        s, cnt = orc.agg(0, v, valid)
        m, _ = orc.agg(1, v, valid)
        if not eok:
            assert s is None and m is None, name
            continue
        assert _bits([s])[0] == _bits([es])[0], (name, hex(_bits([s])[0]), hex(_bits([es])[0]))
        assert _bits([m])[0] == _bits([em])[0], (name, hex(_bits([m])[0]), hex(_bits([em])[0]))


def test_sum_tree_nan_bits_per_group():
    import _nanbits_inputs as inp

    z, cases = _nan_golden()
    for name, keys, v, valid in inp.group_cases():
        exp = z[name]
        ids, uniq, _, _ = orc.group_ids(keys)
        assert np.array_equal(uniq, z[name + "/uniq"]), name
        eok = exp[2] != 0
        for kind, row in ((0, 0), (1, 1)):
            got, ok = orc.groupby_agg(kind, ids, len(uniq), v, valid, nthreads=8)
            assert np.array_equal(np.asarray(ok, bool), eok), (name, kind)
            assert np.array_equal(_bits(got)[eok], _bits(exp[row])[eok]), (name, kind, int((_bits(got)[eok] != _bits(exp[row])[eok]).sum()))
