"""GPU: the RCCL branches of pandasarrow_amd/dist.py on the one GPU of the test box.

RCCL refuses two ranks on one device, so the multi-rank tests (tests/test_gpu_dist.py) run over gloo, whose all-to-all is an
emulation branch.  Here ONE rank initialises the real `nccl` backend (= RCCL on ROCm) in a fresh child process and
PDX_DIST_FORCE_COLLECTIVES=1 keeps every collective on the wire at world size 1: all_gather with padded buffers,
all_to_all_single with split sizes, the count exchanges, on DEVICE tensors through RCCL -- the code the driver's 8-GPU run
executes.  Results are compared bit-for-bit with the CPU oracle.  bench.py's N > 1 path is rehearsed the same way
(PDX_BENCH_FORCE_DIST=1) under torch.distributed.run."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rccl_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PDX_DIST_FORCE_COLLECTIVES="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    out = {}
    try:
        from pandasarrow_amd import _lib as L
        from pandasarrow_amd import dist as pdist
        from pandasarrow_amd.column import Column

        L.check(L.load().pdx_init(0))
        assert dist.get_backend() == "nccl" and not pdist._solo()
        dev = torch.device("cuda", 0)
        # ---- the collective helpers themselves
        t = torch.arange(1000, dtype=torch.int64, device=dev) * 3
        f = torch.arange(1000, dtype=torch.float64, device=dev) / 7
        assert pdist.all_gather_sizes(1000, dev) == [1000]
        assert torch.equal(pdist.all_gather_v(t, [1000]), t)
        a, b, c = pdist.all_gather_v_multi([t, f, (t % 2 == 0)], [1000])
        assert torch.equal(a, t) and torch.equal(b, f) and torch.equal(c, t % 2 == 0) and c.dtype == torch.bool
        assert torch.equal(pdist.all_to_all_v([t])[0], t)
        assert torch.equal(pdist.all_to_all_v([t], [[1000]])[0], t)  # counts known to the caller: no count exchange
        ra, rb = pdist.all_to_all_v_pairs([t], [f])
        assert torch.equal(ra, t) and torch.equal(rb, f) and rb.dtype == torch.float64
        ra, rb = pdist.all_to_all_v_pairs([t], [f], [[1000]])
        assert torch.equal(ra, t) and torch.equal(rb, f)
        assert torch.equal(pdist.all_to_all_v([t[:0]])[0], t[:0])  # empty exchange
        assert torch.equal(pdist._gather_scalars(t[:3]), t[:3][None])
        out["helpers"] = True
        # ---- the sharded operators end to end
        eng = pdist.HipEngine()
        n, nk = 300_007, 3000
        keys = orc.synth_keys(0, n, nk) * 7919 - 12345
        vals = orc.synth_vals(0, n) - 0.5
        kinds = [0, 1, 4, 2, 3]
        res = pdist.groupby_agg_sharded(eng, Column.from_numpy(keys), Column.from_numpy(vals), kinds)
        out["agg"] = {"G": res["G"], "keys": res["keys"].cpu().numpy(), "first": res["first_rows"].cpu().numpy(),
                      "outs": [v.cpu().numpy() for v, _ in res["outs"]]}
        fast = pdist.groupby_sum_mean_count_sharded(eng, Column.from_numpy(keys), Column.from_numpy(vals))
        out["fast"] = {"G": fast["G"], "keys": fast["keys"].cpu().numpy(), "first": fast["first_rows"].cpu().numpy(),
                       "outs": [v.cpu().numpy() for v, _ in fast["outs"]]}
        rng = np.random.default_rng(3)
        m = 100_003
        minute = 60 * 10**9
        ts = 1_600_000_000 * 10**9 + np.sort(rng.integers(0, 900 * minute, m)).astype(np.int64)
        v = rng.standard_normal(m) * 10.0 ** rng.integers(-4, 7, m)
        ok = rng.random(m) > 0.07
        out["ops_in"] = (ts, v, ok)
        out["agg_f64"] = [pdist.aggregate_sharded(eng, Column.from_numpy(v), k) for k in (0, 1, 2, 3, 4)]
        out["agg_f64_nulls"] = [pdist.aggregate_sharded(eng, Column.from_numpy(v, ok), k) for k in (0, 1, 2, 3, 4)]
        out["concat"] = pdist.concat_sharded(eng, Column.from_numpy(v, ok)).to_numpy()
        rs = pdist.resample_agg_sharded(eng, Column.from_numpy(ts, dtype=L.TIMESTAMP_NS), Column.from_numpy(v, ok), [0, 1, 4], 5 * minute)
        out["resample"] = {"labels": rs["labels"].cpu().numpy(), "outs": [(a.cpu().numpy(), None if b is None else b.cpu().numpy()) for a, b in rs["outs"]]}
        torch.cuda.synchronize()
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_rccl_branches_world_size_one():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    got = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert got["helpers"]
    n, nk = 300_007, 3000
    keys = orc.synth_keys(0, n, nk) * 7919 - 12345
    vals = orc.synth_vals(0, n) - 0.5
    ids, uniq, _, first = orc.group_ids(keys)
    exp = {k: orc.groupby_agg(k, ids, len(uniq), vals, nthreads=4)[0] for k in (0, 1, 4, 2, 3)}
    for name, kinds in (("agg", [0, 1, 4, 2, 3]), ("fast", [0, 1, 4])):
        r = got[name]
        assert r["G"] == len(uniq) and np.array_equal(r["keys"], uniq) and np.array_equal(r["first"], first), name
        for g, k in zip(r["outs"], kinds):
            e = exp[k]
            assert (np.array_equal(g.view(np.uint64), e.view(np.uint64)) if e.dtype == np.float64 else np.array_equal(g, e)), (name, k)
    ts, v, ok = got["ops_in"]
    for name, valid in (("agg_f64", None), ("agg_f64_nulls", ok)):
        for kind in (0, 1, 2, 3, 4):
            ev, ecnt = orc.agg(kind, v, valid)
            gv, gcnt = got[name][kind]
            if kind == 4:
                assert gv == ev
            else:
                assert gcnt == ecnt and np.float64(gv).view(np.uint64) == np.float64(ev).view(np.uint64), (name, kind)
    cv, cok = got["concat"]
    assert np.array_equal(cv.view(np.uint64), v.view(np.uint64)) and np.array_equal(cok, ok)
    expr = [orc.resample_agg(k, ts, v, 5 * 60 * 10**9, valid=ok) for k in (0, 1, 4)]
    assert np.array_equal(got["resample"]["labels"], expr[0][0])
    for (gv, gok), (_, ev, eok) in zip(got["resample"]["outs"], expr):
        eok = np.asarray(eok, bool)
        assert (gok is None and eok.all()) or np.array_equal(gok, eok)
        if ev.dtype == np.float64:
            assert np.array_equal(gv.view(np.uint64)[eok], ev.view(np.uint64)[eok])
        else:
            assert np.array_equal(gv[eok], ev[eok])


def test_bench_under_torchrun_nccl():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` with the N > 1 code path forced: init_process_group
    ("nccl", device_id=...), the sharded step with its RCCL collectives, the MAX all-reduce of the time, one JSON line."""
    for which, path in (("c", "sharded-c-abi"), ("torch", "sharded-torch")):  # the library's own RCCL calls; the older torch.distributed orchestration
        env = dict(os.environ, PDX_BENCH_FORCE_DIST="1", PDX_DIST_FORCE_COLLECTIVES="1", PDX_BENCH_DIST=which)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
               "--rows", "2e7", "--keys", "1e5", "--no-cpu-baseline"]
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["n_gpus"] == 1 and line["config"]["path"] == path and line["value"] > 0, (which, line["config"])
        assert all(v for v in line["check"].values() if isinstance(v, bool))


def test_bench_refuses_gpus_mismatch():
    """inside a launcher's environment (WORLD_SIZE set) --gpus must agree with it; without WORLD_SIZE bench.py starts the ranks itself
    (tests/test_gpu_bench_contract.py::test_bench_two_ranks_without_a_launcher)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
