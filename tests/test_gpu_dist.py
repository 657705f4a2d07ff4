"""GPU: the sharded group-by (pandasarrow_amd/dist.py) with the product engine (HipEngine -> C ABI).  One process exercises
the W=1 degenerate exchange; two processes share the single GPU of the test box over `gloo` (RCCL refuses two ranks on one
device; the 8-GPU RCCL run is the driver's) -- the result must be bit-identical to the single-process oracle."""
import os
import socket

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu
KINDS = [0, 1, 4, 2, 3]


def _expected(keys, vals):
    ids, uniq, isnull, first = orc.group_ids(keys)
    return uniq, first, [orc.groupby_agg(k, ids, len(uniq), vals, nthreads=4)[0] for k in KINDS]


def _compare(res, keys, vals):
    uniq, first, outs = _expected(keys, vals)
    assert res["G"] == len(uniq)
    assert np.array_equal(res["keys"], uniq) and np.array_equal(res["first"], first)
    for got, exp in zip(res["outs"], outs):
        if exp.dtype == np.float64:
            assert np.array_equal(got.view(np.uint64), exp.view(np.uint64))
        else:
            assert np.array_equal(got, exp)


def _compare_fast(fast, keys, vals):
    uniq, first, outs = _expected(keys, vals)
    assert fast["G"] == len(uniq) and np.array_equal(fast["keys"].cpu().numpy(), uniq) and np.array_equal(fast["first_rows"].cpu().numpy(), first)
    for j, kind in enumerate((0, 1, 4)):  # sum, mean, count
        got = fast["outs"][j][0].cpu().numpy()
        exp = outs[KINDS.index(kind)]
        assert (np.array_equal(got.view(np.uint64), exp.view(np.uint64)) if exp.dtype == np.float64 else np.array_equal(got, exp)), kind


def _data(n, nk):
    return orc.synth_keys(0, n, nk) * 7919 - 12345, orc.synth_vals(0, n) - 0.5


@pytest.mark.parametrize("wave_fill", ["default", "0"])
def test_sharded_single_rank(monkeypatch, wave_fill):
    """(wave_fill = 0: the partial records of every group come from the thread-per-group kernel; default: groups with at most 64
    interior leaves take the wave-per-group form)"""
    import torch

    if wave_fill != "default":
        monkeypatch.setenv("PDX_PARTIAL_FILL_WAVE", wave_fill)
    from pandasarrow_amd import _lib as L
    from pandasarrow_amd import dist as pdist
    from pandasarrow_amd.column import Column

    L.check(L.load().pdx_init(0))
    keys, vals = _data(400_003, 5000)
    res = pdist.groupby_agg_sharded(pdist.HipEngine(), Column.from_numpy(keys), Column.from_numpy(vals), KINDS)
    out = {"G": res["G"], "keys": res["keys"].cpu().numpy(), "first": res["first_rows"].cpu().numpy(),
           "outs": [v.cpu().numpy() for v, _ in res["outs"]]}
    _compare(out, keys, vals)
    assert all(v for v in pdist.check_result(res, len(keys)).values() if isinstance(v, bool))
    fast = pdist.groupby_sum_mean_count_sharded(pdist.HipEngine(), Column.from_numpy(keys), Column.from_numpy(vals))
    _compare_fast(fast, keys, vals)
    torch.cuda.synchronize()


@pytest.mark.parametrize("narrow", ["default", "0"])
def test_sharded_single_rank_large_dense(monkeypatch, narrow):
    """>= 2^22 rows of dense keys with a three-digit sort plan: the grouped values of the partial-tree exchange come from the full
    narrowing sort (4 -> 2 -> 1 byte keys, group offsets from the scatter offsets) unless PDX_SORT_NARROW=0; a hot key makes one
    group much longer than a tile.  Bit-identical to the oracle either way."""
    import torch
    from pandasarrow_amd import _lib as L
    from pandasarrow_amd import dist as pdist
    from pandasarrow_amd.column import Column

    if narrow != "default":
        monkeypatch.setenv("PDX_SORT_NARROW", narrow)
    L.check(L.load().pdx_init(0))
    n = 4_700_021
    keys = orc.synth_keys(0, n, 300_000) + 1_000_000
    keys[np.random.default_rng(3).random(n) < 0.05] = 1_000_777
    vals = orc.synth_vals(0, n) - 0.5
    fast = pdist.groupby_sum_mean_count_sharded(pdist.HipEngine(), Column.from_numpy(keys), Column.from_numpy(vals))
    _compare_fast(fast, keys, vals)
    torch.cuda.synchronize()


def _worker(rank, world, port, n, nk, q):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pandasarrow_amd import _lib as L
        from pandasarrow_amd import dist as pdist
        from pandasarrow_amd.column import Column

        torch.cuda.set_device(0)
        L.check(L.load().pdx_init(0))
        keys, vals = _data(n, nk)
        lo, hi = n * rank // world, n * (rank + 1) // world
        res = pdist.groupby_agg_sharded(pdist.HipEngine(), Column.from_numpy(keys[lo:hi]), Column.from_numpy(vals[lo:hi]), KINDS, row_offset=lo)
        fast = pdist.groupby_sum_mean_count_sharded(pdist.HipEngine(), Column.from_numpy(keys[lo:hi]), Column.from_numpy(vals[lo:hi]), row_offset=lo)
        if rank == 0:
            q.put({"G": res["G"], "keys": res["keys"].cpu().numpy(), "first": res["first_rows"].cpu().numpy(),
                   "outs": [v.cpu().numpy() for v, _ in res["outs"]],
                   "fast": {"G": fast["G"], "keys": fast["keys"].cpu(), "first_rows": fast["first_rows"].cpu(), "outs": [(v.cpu(), None) for v, _ in fast["outs"]]}})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_three_ranks_one_gpu():
    import torch.multiprocessing as mp

    n, nk, world = 300_007, 3000, 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, nk, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    _compare(got, *_data(n, nk))
    _compare_fast(got["fast"], *_data(n, nk))


# ---------------------------------------------------------------- sharded whole-column aggregates / resample / concat (HipEngine)
def _ops_data():
    rng = np.random.default_rng(3)
    n = 200_003
    minute = 60 * 10**9
    ts = 1_600_000_000 * 10**9 + np.sort(rng.integers(0, 900 * minute, n)).astype(np.int64)
    v = rng.standard_normal(n) * 10.0 ** rng.integers(-4, 7, n)
    ok = rng.random(n) > 0.07
    iv = rng.integers(-2**62, 2**62, n).astype(np.int64)
    return ts, v, ok, iv, minute


def _worker_ops(rank, world, port, q):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pandasarrow_amd import _lib as L
        from pandasarrow_amd import dist as pdist
        from pandasarrow_amd.column import Column

        torch.cuda.set_device(0)
        L.check(L.load().pdx_init(0))
        eng = pdist.HipEngine()
        ts, v, ok, iv, minute = _ops_data()
        n = len(ts)
        lo, hi = n * rank // world, n * (rank + 1) // world
        out = {}
        for name, col in (("f64", Column.from_numpy(v[lo:hi])), ("f64_nulls", Column.from_numpy(v[lo:hi], ok[lo:hi])), ("i64", Column.from_numpy(iv[lo:hi]))):
            out[name] = [pdist.aggregate_sharded(eng, col, k) for k in (0, 1, 2, 3, 4)]
        cat = pdist.concat_sharded(eng, Column.from_numpy(v[lo:hi], ok[lo:hi]))
        out["concat"] = cat.to_numpy()
        for name, vcol, kw in (("rs_plain", Column.from_numpy(v[lo:hi]), dict(closed_right=False, label_right=False, origin=L.ORIGIN_START_DAY)),
                               ("rs_nulls_right", Column.from_numpy(v[lo:hi], ok[lo:hi]), dict(closed_right=True, label_right=True, origin=L.ORIGIN_START, offset_ns=7 * 10**9))):
            res = pdist.resample_agg_sharded(eng, Column.from_numpy(ts[lo:hi], dtype=L.TIMESTAMP_NS), vcol, [0, 1, 2, 3, 4], 5 * minute, **kw)
            out[name] = {"labels": res["labels"].cpu().numpy(), "outs": [(a.cpu().numpy(), None if b is None else b.cpu().numpy()) for a, b in res["outs"]]}
        if rank == 0:
            q.put(out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_ops_three_ranks_one_gpu():
    import torch.multiprocessing as mp

    world = 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_ops, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ts, v, ok, iv, minute = _ops_data()
    for name, (vals, valid) in (("f64", (v, None)), ("f64_nulls", (v, ok)), ("i64", (iv, None))):
        for kind in (0, 1, 2, 3, 4):
            ev, ecnt = orc.agg(kind, vals, valid)
            gv, gcnt = got[name][kind]
            if kind == 4:
                assert gv == ev
            elif isinstance(ev, float):
                assert gcnt == ecnt and np.float64(gv).view(np.uint64) == np.float64(ev).view(np.uint64), (name, kind, gv, ev)
            else:
                assert gcnt == ecnt and gv == ev, (name, kind, gv, ev)
    cv, cok = got["concat"]
    assert np.array_equal(cv.view(np.uint64), v.view(np.uint64)) and np.array_equal(cok, ok)
    for name, valid, kw in (("rs_plain", None, dict(closed_right=False, label_right=False, origin=1)),
                            ("rs_nulls_right", ok, dict(closed_right=True, label_right=True, origin=2, offset_ns=7 * 10**9))):
        exp = [orc.resample_agg(k, ts, v, 5 * minute, valid=valid, **kw) for k in (0, 1, 2, 3, 4)]
        assert np.array_equal(got[name]["labels"], exp[0][0])
        for (gv, gok), (_, ev, eok) in zip(got[name]["outs"], exp):
            eok = np.asarray(eok, bool)
            assert (gok is None and eok.all()) or np.array_equal(gok, eok)
            if ev.dtype == np.float64:
                assert np.array_equal(gv.view(np.uint64)[eok], ev.view(np.uint64)[eok]), name
            else:
                assert np.array_equal(gv[eok], ev[eok]), name
