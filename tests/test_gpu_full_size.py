"""GPU: BASELINE.json's configurations at their FULL sizes.  The oracle cannot run 1e9 rows in a test, so every
check here is either a size-independent property or the bit-equality of two INDEPENDENT kernel chains of this library on the same
resident input (plus torch's own indexing for the pure data-movement ops):

  C1  (a + b).sum() at 1e9 rows           whole-column pairwise kernels  ==  group-by with ONE key (many-waves-per-group reducers)
  C2  filter + take, 1e8 rows x 8 + index  streaming filter / fused gather ==  torch boolean / integer indexing
  C3  group_by.agg(sum, mean, count), 1e9 / 1e6   narrowing sort + fused last digit (plan asserted)  ==  two 5e8-row chunks through the
                                           classic reducers merged by the partial-tree exchange (pdx_groupby_sum_mean_count_chunked)
  C5  resample('1min').mean(), 1e9 rows    arithmetic bins on the sorted axis  ==  downsample('1T') (calendar rounding, runs of labels)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SUM, MEAN, COUNT = 0, 1, 4


@pytest.fixture(scope="module")
def px():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from pandasarrow_amd import _lib as L
    from pandasarrow_amd import api, column, dist

    L.check(L.load().pdx_init(0))

    class NS:
        pass

    ns = NS()
    ns.L, ns.K, ns.api, ns.dist, ns.torch = L, column, api, dist, torch
    yield ns
    L.load().pdx_trim_pool()


def _bits(t):
    import torch

    return t.view(torch.int64) if t.dtype == torch.float64 else t


def test_c1_add_sum_full_size(px):
    K, L, torch = px.K, px.L, px.torch
    n = 1_000_000_000
    a, b = K.synth_vals(0, n, 1), K.synth_vals(0, n, 2)
    c = K.binary(L.ADD, a, b)
    assert torch.equal(c.values[:n], a.values[:n] + b.values[:n])          # one rounding per element: any correct add agrees
    total, cnt = K.aggregate(L.AGG_SUM, c)
    assert cnt == n
    one_key = K.Column(L.INT64, n, torch.zeros(1, dtype=torch.int64, device=c.values.device).expand(n).contiguous(), None)
    gb = K.GroupByHandle.create(one_key)
    s, m, k = gb.agg(c, [SUM, MEAN, COUNT])
    assert gb.num_groups == 1 and int(k.values[0].item()) == n
    assert np.float64(total).view(np.uint64) == s.to_numpy()[0].view(np.uint64)[0]   # same pairwise tree from two different kernel families
    assert np.float64(total / n).view(np.uint64) == m.to_numpy()[0].view(np.uint64)[0]
    del one_key, gb


def test_c2_filter_take_full_size(px):
    K, L, api, torch = px.K, px.L, px.api, px.torch
    n = 100_000_000
    cols = {f"c{j}": K.synth_vals(0, n, 20 + j) for j in range(8)}
    df = api.DataFrame(cols, index=K.synth_keys(0, n, 1 << 62))
    mask = df["c0"] > 0.5
    sel = cols["c0"].values[:n] > 0.5
    out = df.where(mask)
    m = int(sel.sum().item())
    assert out.num_rows() == m == K.filter_count(mask.col)
    for j in range(8):
        assert torch.equal(_bits(out.cols[j].values[:m]), _bits(cols[f"c{j}"].values[:n][sel])), j
    assert torch.equal(out.index.values[:m], df.index.values[:n][sel])
    mtake = n // 2
    idx = K.synth_keys(7, mtake, n)
    tk = df.take(api.Series(idx))
    for j in (0, 3, 7):
        assert torch.equal(_bits(tk.cols[j].values[:mtake]), _bits(cols[f"c{j}"].values[:n][idx.values[:mtake]])), j
    assert torch.equal(tk.index.values[:mtake], df.index.values[:n][idx.values[:mtake]])


def test_c3_groupby_full_size_two_kernel_chains(px):
    K, L, torch = px.K, px.L, px.torch
    n, nk = 1_000_000_000, 1_000_000
    keys, vals = K.synth_keys(0, n, nk), K.synth_vals(0, n)
    gb = K.GroupByHandle.create(keys)
    s, m, c = gb.agg(vals, [SUM, MEAN, COUNT])
    assert gb.last_plan() == {"slots": "dense", "sort": "narrow:7+7", "layout": "fused", "reducer": "flr_reduce_dense", "bound": "0"}
    G = gb.num_groups
    assert G == nk and int(c.values[:G].sum().item()) == n
    fr = gb.first_rows()
    assert bool((fr[1:] > fr[:-1]).all().item())
    uk = gb.unique_keys()
    assert torch.equal(keys.values[fr], uk.values[:G])                                   # the key AT each group's first row is the group's key
    assert torch.equal(m.values[:G], s.values[:G] / c.values[:G].to(torch.float64))
    res = px.dist.groupby_sum_mean_count_chunked(keys, vals, 500_000_000)                # 2 chunks: classic reducers + partial-tree merge
    assert res["G"] == G and torch.equal(res["keys"], uk.values[:G]) and torch.equal(res["first_rows"], fr)
    assert torch.equal(_bits(res["outs"][0][0]), _bits(s.values[:G])), "sum: fused chain != chunked chain"
    assert torch.equal(_bits(res["outs"][1][0]), _bits(m.values[:G]))
    assert torch.equal(res["outs"][2][0], c.values[:G])


def test_c5_resample_full_size_two_paths(px):
    K, L, api, torch = px.K, px.L, px.api, px.torch
    n = 1_000_000_000
    ts = K.synth_ts(0, n, 946_684_800 * 10**9, 100_000_000)      # 100 ms spacing: 600 rows per one-minute bin
    vals = K.synth_vals(0, n)
    ser = api.Series(vals, index=ts, name="v")
    r = ser.resample("1min").mean()
    nb = r.num_rows()
    assert nb == (n + 599) // 600
    d = api.DataFrame({"v": vals}, index=ts).downsample("1T", closed_label_right=False).mean()
    assert d.num_rows() == nb
    assert torch.equal(r.index.values[:nb], d.index.values[:nb])
    assert torch.equal(_bits(r["v"].col.values[:nb]), _bits(d["v"].col.values[:nb]))
    cnt = ser.resample("1min").count()
    assert int(cnt["v"].col.values[:nb].sum().item()) == n


@pytest.mark.parametrize("dense", ["1", "0"])
def test_c3_against_the_oracle_at_1e8_rows(px, monkeypatch, dense):
    """SURVEY 8(d)'s CPU-comparison size -- 1e8 rows / 1e6 keys -- through the HIP path and the C oracle (OpenMP over groups, ~5 s), bit for
    bit, for the default dense plan AND with every key through the hash table; the plan is asserted, so the kernels compared are the ones
    the bench times.  The order-free kinds (min / max / count without the value sort, gb_acc.hpp) ride on the same handle."""
    import os

    import oracle as orc

    monkeypatch.setenv("PDX_GROUPBY_DENSE", dense)
    K, L = px.K, px.L
    n, nk = 100_000_000, 1_000_000
    threads = min(64, os.cpu_count() or 1)
    keys, vals = K.synth_keys(0, n, nk), K.synth_vals(0, n)
    gb = K.GroupByHandle.create(keys)
    s, m, c = gb.agg(vals, [SUM, MEAN, COUNT])
    plan = gb.last_plan()
    hk, hv = orc.synth_keys(0, n, nk), orc.synth_vals(0, n)
    ek, es, em, ec = orc.groupby_sum_mean_count(hk, hv, nthreads=threads)
    # (100 rows per group = 6103-row runs: since round 4's lower run threshold the DEFAULT plan at this size is the narrowing sort + the
    #  fused last digit, i.e. the kernels the bench times at 1e9 rows)
    assert plan["slots"] == ("dense" if dense == "1" else "hash_lds") and plan["layout"] == "fused" and plan["reducer"].startswith("flr_reduce"), plan
    assert np.array_equal(gb.unique_keys().to_numpy()[0], ek)
    assert np.array_equal(s.to_numpy()[0].view(np.uint64), es.view(np.uint64))
    assert np.array_equal(m.to_numpy()[0].view(np.uint64), em.view(np.uint64))
    assert np.array_equal(c.to_numpy()[0], ec)
    # order-free kinds on the same handle: no value sort
    ids = orc.group_ids(hk)[0]
    mn, mx = gb.agg(vals, [2, 3])
    assert gb.last_plan()["reducer"] == "lds_acc", gb.last_plan()
    for kind, got in ((2, mn), (3, mx)):
        exp = orc.groupby_agg(kind, ids, len(ek), hv, None, nthreads=threads)[0]
        assert np.array_equal(got.to_numpy()[0].view(np.uint64), exp.view(np.uint64)), kind
    cnt = gb.agg(vals, [COUNT])[0]
    assert gb.last_plan()["reducer"] in ("lds_acc", "sizes_cache") and np.array_equal(cnt.to_numpy()[0], ec)
    del gb
