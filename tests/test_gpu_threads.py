"""The ABI's threading contract (include/pdx/abi.h "Threading"; reference call pattern: tbb::parallel_for workers calling the
compute boundary concurrently, src/pd_core_macros.h:21,56,94,122).  Several host threads, each on its own HIP stream, push
different inputs through pdx_binary / pdx_aggregate / pdx_groupby_* / pdx_filter at the same time; every result is compared
bit-for-bit with the CPU oracle.  The scratch pool is shared between the threads, so a block handed to one thread while another
stream's kernels still use it shows up here as a wrong answer."""
import threading

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


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


def _expected(seed, n, nk):
    keys = orc.synth_keys(seed * 1000, n, nk)
    vals = orc.synth_vals(seed * 1000, n, seed)
    ek, es, em, ec = orc.groupby_sum_mean_count(keys, vals)
    total, _ = orc.agg(orc.AGG_SUM, vals)
    added, _ = orc.binary(orc.ADD, vals, vals)
    mask = vals > 0.5
    return dict(keys=ek, sum=es, mean=em, count=ec, total=total, added=added, filtered=vals[mask])


def _worker(px, seed, n, nk, rounds, out, errors):
    torch = px.torch
    try:
        torch.cuda.set_device(0)
        px.L.check(px.L.load().pdx_init(0))  # per-thread device selection, as a tbb worker would do
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            for r in range(rounds):
                keys = px.K.synth_keys(seed * 1000, n, nk)
                vals = px.K.synth_vals(seed * 1000, n, seed)
                gb = px.K.GroupByHandle.create(keys)
                s, m, c = gb.agg(vals, [px.L.AGG_SUM, px.L.AGG_MEAN, px.L.AGG_COUNT])
                uk = gb.unique_keys()
                added = px.K.binary(px.L.ADD, vals, vals)
                total, _ = px.K.aggregate(px.L.AGG_SUM, vals)
                filt = px.K.filter([vals], px.K.compare(px.L.GT, vals, 0.5))[0]
                res = dict(keys=uk.to_numpy()[0], sum=s.to_numpy()[0], mean=m.to_numpy()[0], count=c.to_numpy()[0], total=total,
                           added=added.to_numpy()[0], filtered=filt.to_numpy()[0])
                gb.close()
                out.append((seed, r, res))
            stream.synchronize()
    except Exception as e:  # noqa: BLE001
        errors.append((seed, repr(e)))


@pytest.mark.parametrize("nthreads", [2, 4])
def test_concurrent_threads_streams(px, nthreads):
    n, nk, rounds = 400_003, 3000, 6
    sizes = [n + 1111 * t for t in range(nthreads)]  # different sizes -> different scratch footprints per thread
    exp = {t: _expected(t + 1, sizes[t], nk + 7 * t) for t in range(nthreads)}
    out, errors = [], []
    threads = [threading.Thread(target=_worker, args=(px, t + 1, sizes[t], nk + 7 * t, rounds, out, errors)) for t in range(nthreads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(out) == nthreads * rounds
    for seed, r, res in out:
        e = exp[seed - 1]
        assert np.array_equal(res["keys"], e["keys"]), (seed, r, "keys")
        for k in ("sum", "mean", "added", "filtered"):
            assert np.array_equal(np.asarray(res[k]).view(np.uint64), np.asarray(e[k]).view(np.uint64)), (seed, r, k)
        assert np.array_equal(res["count"], e["count"]), (seed, r, "count")
        assert np.float64(res["total"]).view(np.uint64) == np.float64(e["total"]).view(np.uint64), (seed, r, "total")


def test_pool_reuse_is_stream_ordered(px):
    """Two streams of ONE thread: a block freed by a call on stream A while A is still busy must not be handed to a call on
    stream B (it would be overwritten under A's feet).  A long group-by keeps A busy; B's work runs against the same pool."""
    torch = px.torch
    n = 3_000_000
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    keys_np, vals_np = orc.synth_keys(0, n, 50_000), orc.synth_vals(0, n, 3)
    ek, es, em, ec = orc.groupby_sum_mean_count(keys_np, vals_np)
    small = orc.synth_vals(77, 200_000, 5)
    exp_small = orc.agg(orc.AGG_SUM, small)[0]
    with torch.cuda.stream(sa):
        keys, vals = px.K.synth_keys(0, n, 50_000), px.K.synth_vals(0, n, 3)
    with torch.cuda.stream(sb):
        sv = px.K.synth_vals(77, 200_000, 5)
    torch.cuda.synchronize()
    for _ in range(5):
        with torch.cuda.stream(sa):
            gb = px.K.GroupByHandle.create(keys)
            s, m, c = gb.agg(vals, [px.L.AGG_SUM, px.L.AGG_MEAN, px.L.AGG_COUNT])
            gb.close()  # the handle's blocks go back to the pool with A's work still queued behind them
        with torch.cuda.stream(sb):
            for _ in range(4):
                got, _ = px.K.aggregate(px.L.AGG_SUM, sv)
                assert np.float64(got).view(np.uint64) == np.float64(exp_small).view(np.uint64)
        with torch.cuda.stream(sa):
            assert np.array_equal(s.to_numpy()[0].view(np.uint64), es.view(np.uint64))
            assert np.array_equal(c.to_numpy()[0], ec)
