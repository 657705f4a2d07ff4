"""GPU: the sharded path behind the C ABI (pdx_dist_*, csrc/dist.hip) -- orchestration, glue kernels and collectives inside
libpdx_hip.so.

* one rank, RCCL transport, PDX_DIST_FORCE_COLLECTIVES=1: ncclCommInitRank, ncclAllGather and grouped ncclSend / ncclRecv made by the
  library itself, in a fresh child process (RCCL state is per process);
* three ranks sharing the box's one GPU over the custom transport (callbacks on torch.distributed / gloo: RCCL refuses duplicate
  devices): the same C orchestration with real cross-rank data -- global dictionary, count exchange, partial records to owners,
  replay, result gather -- bit-identical to the single-process oracle; concat with nulls and unequal shard sizes."""
import os
import socket

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _data(n, nk, null_keys=False):
    keys = orc.synth_keys(0, n, nk) * 7919 - 12345
    vals = orc.synth_vals(0, n) - 0.5
    kvalid = None
    if null_keys:
        kvalid = np.random.default_rng(4).random(n) > 0.02
    return keys, vals, kvalid


def _check(res, keys, vals, kvalid):
    ids, uniq, isnull, first = orc.group_ids(keys, kvalid)
    assert res["G"] == len(uniq)
    assert np.array_equal(res["keys_ok"], ~isnull) and np.array_equal(res["keys"][~isnull], uniq[~isnull])
    assert np.array_equal(res["first_rows"], first)
    for j, kind in enumerate((0, 1, 4)):
        exp = orc.groupby_agg(kind, ids, len(uniq), vals, nthreads=4)[0]
        got = res["outs"][j]
        assert np.array_equal(got.view(np.uint64), exp.view(np.uint64)), kind


def _to_host(res):
    return {"G": res["G"], "keys": res["keys"].cpu().numpy(), "keys_ok": res["keys_ok"].cpu().numpy(), "first_rows": res["first_rows"].cpu().numpy(),
            "outs": [v.cpu().numpy() for v, _ in res["outs"]], "records": res["records"]}


def _resample_data():
    rng = np.random.default_rng(3)
    n = 200_003
    minute = 60 * 10**9
    ts = 1_600_000_000 * 10**9 + np.sort(rng.integers(0, 900 * minute, n)).astype(np.int64)
    v = rng.standard_normal(n) * 10.0 ** rng.integers(-4, 7, n)
    ok = rng.random(n) > 0.07
    return ts, v, ok, minute


RESAMPLE_CASES = (("plain", False, dict(closed_right=False, label_right=False, origin=1)),
                  ("nulls_right", True, dict(closed_right=True, label_right=True, origin=2, offset_ns=7 * 10**9)))


def _resample_cases(cd, rank, world, L, Column):
    ts, v, ok, minute = _resample_data()
    n = len(ts)
    cuts = [n * q // world for q in range(world + 1)]
    if world == 3:
        cuts = [0, n * 30 // 100, n * 30 // 100 + 17, n]  # a 17-row shard: its rows all belong to a bin that opened on the rank before
    lo, hi = cuts[rank], cuts[rank + 1]
    out = {}
    for name, nulls, kw in RESAMPLE_CASES:
        res = cd.resample(Column.from_numpy(ts[lo:hi], dtype=L.TIMESTAMP_NS), Column.from_numpy(v[lo:hi], ok[lo:hi] if nulls else None), [0, 1, 2, 3, 4],
                          5 * minute, **kw)
        out[name] = {"labels": res["labels"].cpu().numpy(), "outs": [(a.cpu().numpy(), None if b is None else b.cpu().numpy()) for a, b in res["outs"]]}
    return out


def _check_resample(got):
    ts, v, ok, minute = _resample_data()
    for name, nulls, kw in RESAMPLE_CASES:
        exp = [orc.resample_agg(k, ts, v, 5 * minute, valid=ok if nulls else None, **kw) for k in (0, 1, 2, 3, 4)]
        assert np.array_equal(got[name]["labels"], exp[0][0]), name
        for (gv, gok), (_, ev, eok) in zip(got[name]["outs"], exp):
            eok = np.asarray(eok, bool)
            assert (gok is None and eok.all()) or np.array_equal(gok, eok), name
            if ev.dtype == np.float64:
                assert np.array_equal(gv.view(np.uint64)[eok], ev.view(np.uint64)[eok]), name
            else:
                assert np.array_equal(gv[eok], ev[eok]), name


def _rccl_worker(q):
    os.environ["PDX_DIST_FORCE_COLLECTIVES"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch

    from pandasarrow_amd import _lib as L
    from pandasarrow_amd import dist as pdist
    from pandasarrow_amd.column import Column

    torch.cuda.set_device(0)
    L.check(L.load().pdx_init(0))
    out = {}
    cd = pdist.CDist("rccl")   # no process group: world size 1, the library creates its own RCCL communicator
    for name, (n, nk, nullk) in {"small": (200_003, 3000, False), "null_keys": (150_001, 500, True), "large_dense": (4_700_021, 300_000, False)}.items():
        keys, vals, kvalid = _data(n, nk, nullk)
        if name == "large_dense":
            keys = orc.synth_keys(0, n, nk) + 1_000_000
        out[name] = _to_host(cd.groupby_sum_mean_count(Column.from_numpy(keys, kvalid), Column.from_numpy(vals)))
    v = np.arange(1000, dtype=np.float64) / 7
    ok = np.arange(1000) % 5 != 0
    c = cd.concat(Column.from_numpy(v, ok, offset=3))
    out["concat"] = c.to_numpy()
    out["resample"] = _resample_cases(cd, 0, 1, L, Column)
    cd.close()
    q.put(out)


def test_c_abi_sharded_one_rank_rccl_on_the_wire():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(q,))
    p.start()
    got = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    for name, (n, nk, nullk) in {"small": (200_003, 3000, False), "null_keys": (150_001, 500, True), "large_dense": (4_700_021, 300_000, False)}.items():
        keys, vals, kvalid = _data(n, nk, nullk)
        if name == "large_dense":
            keys = orc.synth_keys(0, n, nk) + 1_000_000
        _check(got[name], keys, vals, kvalid)
    cv, cok = got["concat"]
    assert np.array_equal(cv, np.arange(1000, dtype=np.float64) / 7) and np.array_equal(cok, np.arange(1000) % 5 != 0)
    _check_resample(got["resample"])


def _gloo_worker(rank, world, port, q):
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
        cd = pdist.CDist("torch")
        out = {}
        # unequal shards (the last rank holds 40 % of the rows); keys whose first occurrence lies on different ranks; a null key
        for name, (n, nk, nullk) in {"uniform": (300_007, 3000, False), "null_keys": (200_003, 700, True), "few_groups": (100_003, 5, False)}.items():
            keys, vals, kvalid = _data(n, nk, nullk)
            cuts = [0, n * 25 // 100, n * 60 // 100, n]
            lo, hi = cuts[rank], cuts[rank + 1]
            res = cd.groupby_sum_mean_count(Column.from_numpy(keys[lo:hi], None if kvalid is None else kvalid[lo:hi]), Column.from_numpy(vals[lo:hi]),
                                            row_offset=lo)
            out[name] = _to_host(res)
        rng = np.random.default_rng(11)
        v, ok = rng.standard_normal(10_000), rng.random(10_000) > 0.1
        cuts = [0, 1, 7003, 10_000]
        lo, hi = cuts[rank], cuts[rank + 1]
        out["concat"] = cd.concat(Column.from_numpy(v[lo:hi], ok[lo:hi] if rank != 1 else None)).to_numpy()   # rank 1's shard has no bitmap
        out["resample"] = _resample_cases(cd, rank, world, L, Column)
        cd.close()
        if rank == 0:
            q.put(out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_c_abi_sharded_three_ranks_one_gpu():
    import torch.multiprocessing as mp

    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for name, (n, nk, nullk) in {"uniform": (300_007, 3000, False), "null_keys": (200_003, 700, True), "few_groups": (100_003, 5, False)}.items():
        keys, vals, kvalid = _data(n, nk, nullk)
        _check(got[name], keys, vals, kvalid)
        assert got[name]["records"] > 0
    rng = np.random.default_rng(11)
    v, ok = rng.standard_normal(10_000), rng.random(10_000) > 0.1
    ok[1:7003] = True
    cv, cok = got["concat"]
    assert np.array_equal(cok, ok) and np.array_equal(cv[ok].view(np.uint64), v[ok].view(np.uint64))
    _check_resample(got["resample"])


@pytest.mark.parametrize("shape", ["four_chunks", "ragged_seven", "null_keys_bit_offsets", "one_chunk"])
def test_chunked_groupby_beyond_the_row_limit_merge(shape):
    """pdx_groupby_sum_mean_count_chunked: the path for inputs of more than 2^31 - 1 rows, exercised with small chunks -- every chunk
    is a virtual rank on a host thread of its own, merged through the partial-tree records: bit-identical to ONE tree over the column"""
    import torch

    from pandasarrow_amd import _lib as L
    from pandasarrow_amd import dist as pdist
    from pandasarrow_amd.column import Column

    L.check(L.load().pdx_init(0))
    n, nk, chunk, nullk = {"four_chunks": (400_003, 3000, 100_001, False), "ragged_seven": (333_337, 50, 50_000, False),
                           "null_keys_bit_offsets": (250_007, 700, 77_777, True), "one_chunk": (120_000, 900, 0, False)}[shape]
    keys, vals, kvalid = _data(n, nk, nullk)
    res = pdist.groupby_sum_mean_count_chunked(Column.from_numpy(keys, kvalid, offset=3 if nullk else 0), Column.from_numpy(vals), chunk)
    _check(_to_host(res), keys, vals, kvalid)
    torch.cuda.synchronize()


# ---------------------------------------------------------------- round 4: order-free kinds over shards; collective error gates
def _order_free_data(dtype):
    rng = np.random.default_rng(21)
    n, nk = 260_003, 4_000
    keys = rng.integers(0, nk, n).astype(np.int64) * 31 + 5
    if dtype == "f64":
        vals = rng.standard_normal(n)
        m = rng.integers(0, n, n // 25)
        vals[m] = rng.choice(np.array([0.0, -0.0, np.nan, np.inf, -np.inf]), m.size)
    else:
        vals = rng.integers(-2**62, 2**62, n).astype(np.int64)
        vals[rng.integers(0, n, 500)] = np.iinfo(np.int64).min
        vals[rng.integers(0, n, 500)] = np.iinfo(np.int64).max
    valid = rng.random(n) > 0.08
    valid[keys == 5] = False  # a group without any valid value
    return keys, vals, valid


ORDER_FREE_CASES = (("f64", False, (2, 3, 4)), ("f64", True, (3, 4, 2)), ("i64", False, (0, 2, 3, 4)), ("i64", True, (4, 0)))


def _order_free_cases(cd, rank, world, Column):
    out = {}
    for dtype, nulls, kinds in ORDER_FREE_CASES:
        keys, vals, valid = _order_free_data(dtype)
        n = len(keys)
        cuts = [n * q // world for q in range(world + 1)] if world != 3 else [0, n * 20 // 100, n * 55 // 100, n]
        lo, hi = cuts[rank], cuts[rank + 1]
        res = cd.groupby_order_free(Column.from_numpy(keys[lo:hi]), Column.from_numpy(vals[lo:hi], valid[lo:hi] if nulls else None), list(kinds), row_offset=lo)
        out[(dtype, nulls)] = {"keys": res["keys"].cpu().numpy(), "first_rows": res["first_rows"].cpu().numpy(),
                               "outs": [(a.cpu().numpy(), None if b is None else b.cpu().numpy()) for a, b in res["outs"]]}
    return out


def _check_order_free(got, world):
    for dtype, nulls, kinds in ORDER_FREE_CASES:
        keys, vals, valid = _order_free_data(dtype)
        ids, uniq, _, first = orc.group_ids(keys)
        g = got[(dtype, nulls)]
        assert np.array_equal(g["keys"], uniq) and np.array_equal(g["first_rows"], first)
        for kind, (gv, gok) in zip(kinds, g["outs"]):
            ev, eok = orc.groupby_agg(kind, ids, len(uniq), vals, valid if nulls else None, nthreads=4)
            eok = np.asarray(eok, bool)
            assert (gok is None and eok.all()) or np.array_equal(gok, eok), (dtype, nulls, kind)
            if ev.dtype == np.float64:
                a, b = gv[eok], ev[eok]
                same = (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
                if world > 1 and kind == 3 and nulls:
                    # the documented corner (include/pdx/abi.h at pdx_dist_groupby_order_free): a maximum that is a tie of zeros of both
                    # signs ACROSS ranks in a group with a null may keep another share's zero -- the value is a zero either way
                    same |= (a == 0.0) & (b == 0.0)
                assert same.all(), (dtype, nulls, kind, int((~same).sum()))
            else:
                assert np.array_equal(gv[eok], ev[eok]), (dtype, nulls, kind)


def _gloo_worker_r4(rank, world, port, q):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PDX_ACC_MIN_ROWS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pandasarrow_amd import _lib as L
        from pandasarrow_amd import dist as pdist
        from pandasarrow_amd.column import Column

        torch.cuda.set_device(0)
        L.check(L.load().pdx_init(0))
        cd = pdist.CDist("torch")
        out = {"order_free": _order_free_cases(cd, rank, world, Column)}
        # ---- one bad shard: EVERY rank must come back with an error, none may wait in a collective
        keys, vals, valid = _order_free_data("f64")
        n = len(keys)
        lo, hi = n * rank // world, n * (rank + 1) // world
        errs = {}
        try:  # rank 1's values carry nulls: the partial-tree exchange refuses them -- on all ranks
            cd.groupby_sum_mean_count(Column.from_numpy(keys[lo:hi]), Column.from_numpy(vals[lo:hi], valid[lo:hi] if rank == 1 else None), row_offset=lo)
            errs["nulls"] = None
        except L.PdxError as e:
            errs["nulls"] = (e.status, str(e))
        ts, v, ok, minute = _resample_data()
        m = len(ts)
        lo, hi = m * rank // world, m * (rank + 1) // world
        tloc = ts[lo:hi].copy()
        if rank == 2:  # unsorted INSIDE the shard (its first and last rows still fit the neighbours)
            tloc[1000], tloc[1001] = tloc[1001] + 5, tloc[1000]
        try:
            cd.resample(Column.from_numpy(tloc, dtype=L.TIMESTAMP_NS), Column.from_numpy(v[lo:hi]), [0, 4], 5 * minute)
            errs["unsorted"] = None
        except L.PdxError as e:
            errs["unsorted"] = (e.status, str(e))
        try:  # a kind the order-free entry refuses is refused before any exchange, and the communicator is still usable afterwards
            cd.groupby_order_free(Column.from_numpy(keys[:100]), Column.from_numpy(vals[:100]), [0])
            errs["kind"] = None
        except L.PdxError as e:
            errs["kind"] = (e.status, str(e))
        out["after_errors"] = _to_host(cd.groupby_sum_mean_count(Column.from_numpy(keys[n * rank // world:n * (rank + 1) // world]),
                                                                 Column.from_numpy(np.nan_to_num(vals[n * rank // world:n * (rank + 1) // world])),
                                                                 row_offset=n * rank // world))
        cd.close()
        q.put((rank, out if rank == 0 else None, errs))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_c_abi_order_free_and_error_gates_three_ranks_one_gpu():
    import torch.multiprocessing as mp

    from pandasarrow_amd import _lib as L

    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker_r4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    by_rank = {r: (out, errs) for r, out, errs in got}
    _check_order_free(by_rank[0][0]["order_free"], world)
    for r in range(world):
        errs = by_rank[r][1]
        assert errs["nulls"] is not None and errs["nulls"][0] == L.NOT_IMPLEMENTED, (r, errs)
        assert ("rank 1" in errs["nulls"][1]) == (r != 1), (r, errs)  # the failing rank keeps its own message, the others name it
        assert errs["unsorted"] is not None and errs["unsorted"][0] == L.INVALID, (r, errs)
        assert errs["kind"] is not None and errs["kind"][0] == L.NOT_IMPLEMENTED, (r, errs)
    keys, vals, _ = _order_free_data("f64")
    _check(by_rank[0][0]["after_errors"], keys, np.nan_to_num(vals), None)


def _rccl_worker_r4(q):
    os.environ["PDX_DIST_FORCE_COLLECTIVES"] = "1"
    os.environ["PDX_ACC_MIN_ROWS"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch

    from pandasarrow_amd import _lib as L
    from pandasarrow_amd import dist as pdist
    from pandasarrow_amd.column import Column

    torch.cuda.set_device(0)
    L.check(L.load().pdx_init(0))
    cd = pdist.CDist("rccl")
    out = _order_free_cases(cd, 0, 1, Column)
    cd.close()
    q.put(out)


def test_c_abi_order_free_one_rank_rccl_on_the_wire():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker_r4, args=(q,))
    p.start()
    got = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    _check_order_free(got, 1)


@pytest.mark.parametrize("shape", ["four_chunks_f64_nulls", "ragged_int64", "one_chunk"])
def test_chunked_order_free_beyond_the_row_limit(shape, monkeypatch):
    """pdx_groupby_order_free_chunked: min / max / count / int64 sum for inputs of more than 2^31 - 1 rows, exercised with small chunks:
    every chunk is a virtual rank on a host thread of its own, dense partials folded in chunk order, equal to the oracle"""
    import torch

    from pandasarrow_amd import _lib as L
    from pandasarrow_amd import dist as pdist
    from pandasarrow_amd.column import Column

    monkeypatch.setenv("PDX_ACC_MIN_ROWS", "1")
    L.check(L.load().pdx_init(0))
    dtype, nulls, kinds, chunk = {"four_chunks_f64_nulls": ("f64", True, [2, 3, 4], 70_001), "ragged_int64": ("i64", False, [0, 2, 3, 4], 41_000),
                                  "one_chunk": ("f64", False, [3, 2], 0)}[shape]
    keys, vals, valid = _order_free_data(dtype)
    res = pdist.groupby_order_free_chunked(Column.from_numpy(keys), Column.from_numpy(vals, valid if nulls else None), kinds, chunk)
    got = {(dtype, nulls): {"keys": res["keys"].cpu().numpy(), "first_rows": res["first_rows"].cpu().numpy(),
                            "outs": [(a.cpu().numpy(), None if b is None else b.cpu().numpy()) for a, b in res["outs"]]}}
    ids, uniq, _, first = orc.group_ids(keys)
    g = got[(dtype, nulls)]
    assert np.array_equal(g["keys"], uniq) and np.array_equal(g["first_rows"], first)
    for kind, (gv, gok) in zip(kinds, g["outs"]):
        ev, eok = orc.groupby_agg(kind, ids, len(uniq), vals, valid if nulls else None, nthreads=4)
        eok = np.asarray(eok, bool)
        assert (gok is None and eok.all()) or np.array_equal(gok, eok), (shape, kind)
        if ev.dtype == np.float64:
            a, b = gv[eok], ev[eok]
            same = (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
            if kind == 3 and nulls:
                same |= (a == 0.0) & (b == 0.0)  # the documented corner: a maximum that is a tie of zeros across chunks in a group with a null
            assert same.all(), (shape, kind, int((~same).sum()))
        else:
            assert np.array_equal(gv[eok], ev[eok]), (shape, kind)
    torch.cuda.synchronize()


@pytest.mark.parametrize("shape", ["three_full_chunks", "ragged_million_keys", "one_key_a_third_of_the_rows", "few_large_groups"])
def test_chunked_groupby_fused_record_emission(shape):
    """Dense keys and chunks of more than 2^22 rows: inside the orchestration the value sort stops after two passes and k_flr_emit does the
    last digit and the partial-tree records in one kernel (the open leaf of every group starts with prefix % 16 virtual rows, the counter at
    ceil(prefix / 16) with virtual bits: orphans and pending nodes).  Every chunk but the first has non-zero prefixes, the last chunk of the
    ragged case is too small for the narrowing sort (classic sort + per-group fill in the same exchange), and the skewed case holds a run
    longer than one workgroup takes (third pass after all).  NaN / inf values ride along.  Bit-identical to ONE tree over the column."""
    import torch

    from pandasarrow_amd import _lib as L
    from pandasarrow_amd import dist as pdist
    from pandasarrow_amd.column import Column

    L.check(L.load().pdx_init(0))
    n, nk, chunk = {"three_full_chunks": (12_900_000, 300_000, 4_300_000), "ragged_million_keys": (17_500_003, 1_000_000, 4_250_001),
                    "one_key_a_third_of_the_rows": (9_000_000, 200_000, 4_500_000), "few_large_groups": (13_100_000, 3_000, 4_333_333)}[shape]
    rng = np.random.default_rng(len(shape))
    keys = orc.synth_keys(0, n, nk) + 1000
    if shape == "few_large_groups":  # 3000 keys spread over a 19-bit dense domain: ~1400 rows per group and chunk -> orphans up to level 6
        keys = (keys - 1000) * 97 + 1000
    if shape == "one_key_a_third_of_the_rows":
        keys[rng.random(n) < 0.33] = 1234
    vals = orc.synth_vals(0, n) - 0.5
    m = rng.integers(0, n, n // 2000)
    vals[m] = rng.choice(np.array([np.nan, -np.nan, np.inf, -np.inf, 1e300, -1e300]), m.size)
    import ctypes as C

    lib = L.load()
    L.check(lib.pdx_profile_enable(1))
    L.check(lib.pdx_profile_reset())
    try:
        res = pdist.groupby_sum_mean_count_chunked(Column.from_numpy(keys), Column.from_numpy(vals), chunk)
        buf = C.create_string_buffer(1 << 16)
        L.check(lib.pdx_profile_report(buf, len(buf)))
    finally:
        lib.pdx_profile_enable(0)
    tags = {ln.split()[0]: int(ln.split()[1]) for ln in buf.value.decode().splitlines() if ln.strip()}
    # (the path under test did run: the fused emission in every chunk that is large enough and not skewed, the per-group kernels elsewhere)
    fused, classic = tags.get("partial_fill_fused", 0), tags.get("partial_fill", 0)
    assert (fused, classic) == {"three_full_chunks": (3, 0), "ragged_million_keys": (4, 1), "one_key_a_third_of_the_rows": (0, 2),
                                "few_large_groups": (3, 1)}[shape], tags
    _check(_to_host(res), keys, vals, None)
    torch.cuda.synchronize()
