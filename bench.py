#!/usr/bin/env python3
"""bench.py -- headline benchmark: hash group-by (sum, mean, count) of 1e9 int64-key rows x 1e6 keys (BASELINE.json
configs[2], "group_by(int64 key).agg(sum,mean,count), 1e9 rows / 1e6 keys, 1 GPU"; configs[3] at N > 1).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full pass of the hot path over the resident synthetic columns: pdx_groupby_create (hash keys ->
first-occurrence group ids) + pdx_groupby_agg(sum, mean, count) (stable sort by group + Arrow-order pairwise reduce),
inputs already in HBM, results left in HBM.  Rank 0 prints ONE JSON line (see README/DESIGN.md for the fields).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec)
ALGO_BYTES_PER_ROW = 16.0  # 8 B int64 key + 8 B fp64 value per row


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=float, default=1e9, help="total rows (strong scaling: split by row range over the ranks)")
    ap.add_argument("--keys", type=float, default=1e6)
    ap.add_argument("--cpu-sample-rows", type=float, default=2.5e8, help="rows of the same workload timed on the host cores (rank 0, N=1), ~10-20 s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the size-independent result checks after the timed region")
    return ap.parse_args()


def profile_report(lib):
    buf = C.create_string_buffer(1 << 16)
    from pandasarrow_amd import _lib as L

    L.check(lib.pdx_profile_report(buf, len(buf)))
    out = {}
    for line in buf.value.decode().splitlines():
        tag, cnt, ms = line.split()
        out[tag] = (int(cnt), float(ms))
    return out


def cpu_baseline(sample_rows, nkeys, gpu_check=None):
    """The oracle's restatement of the reference's Arrow-CPU call sequence, timed on this host ("port").  The same sample is
    also pushed through the HIP path and compared bit-for-bit (the oracle as checker, outside every timed region)."""
    import numpy as np
    import oracle as orc

    threads = max(1, min(16, os.cpu_count() or 1))
    keys = orc.synth_keys(0, sample_rows, nkeys)
    vals = orc.synth_vals(0, sample_rows)
    t0 = time.perf_counter()
    uk, s, m, c = orc.groupby_sum_mean_count(keys, vals, nthreads=threads)
    dt = time.perf_counter() - t0
    out = {"value": sample_rows / dt / 1e9, "unit": "Grows/s", "cores": threads, "kind": "port",
           "sample": f"first {sample_rows:.3g} rows of the same synthetic workload ({len(uk)} groups), "
                     f"oracle/pdx_oracle.c orc_groupby_sum_mean_count, {dt:.1f} s"}
    # context only: Arrow's own multi-threaded hash aggregate on a 1e8-row slice (strongest readily available CPU number; its
    # fp64 sums use a different summation order than the reference's per-group scalar sum, so it is not the parity target)
    try:
        import pyarrow as pa

        rows_pa = min(sample_rows, 100_000_000)
        tbl = pa.table({"k": keys[:rows_pa], "v": vals[:rows_pa]})
        t1 = time.perf_counter()
        tbl.group_by("k").aggregate([("v", "sum"), ("v", "mean"), ("v", "count")])
        out["arrow_hash_aggregate_Grows_per_s"] = rows_pa / (time.perf_counter() - t1) / 1e9
        out["arrow_version"] = pa.__version__
    except Exception:  # pyarrow missing on the box: the port above is the baseline
        pass
    if gpu_check is not None:
        gk, gs, gm, gc = gpu_check(sample_rows)
        out["gpu_matches_oracle_bit_exact"] = bool(np.array_equal(gk, uk) and np.array_equal(gs.view(np.uint64), s.view(np.uint64))
                                                   and np.array_equal(gm.view(np.uint64), m.view(np.uint64)) and np.array_equal(gc, c))
    return out


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # PDX_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices, collectives
    # go through the host) -- never a measurement; the driver's runs use RCCL, one rank per GPU
    backend = os.environ.get("PDX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from pandasarrow_amd import _lib as L
    from pandasarrow_amd import column as K

    lib = L.load()
    L.check(lib.pdx_init(local_rank))

    n_total, nkeys = int(args.rows), int(args.keys)
    lo = n_total * rank // world
    hi = n_total * (rank + 1) // world
    n_local = hi - lo
    keys = K.synth_keys(lo, n_local, nkeys)   # resident in HBM before the timed region
    vals = K.synth_vals(lo, n_local)
    kinds = [L.AGG_SUM, L.AGG_MEAN, L.AGG_COUNT]

    if world == 1:
        def step():
            gb = K.GroupByHandle.create(keys)
            outs = gb.agg(vals, kinds)
            return gb, outs
    else:
        from pandasarrow_amd import dist as pdist

        engine = pdist.HipEngine()

        def step():
            # sum/mean/count with the exact partial-tree exchange (fragments + aligned subtree nodes, no rows shipped)
            return pdist.groupby_sum_mean_count_sharded(engine, keys, vals, row_offset=lo)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    res = None
    for _ in range(args.warmup):
        res = step()  # same object lifetime pattern as the timed loop, so the scratch pool reaches its steady state here
    lib.pdx_profile_reset()
    lib.pdx_profile_enable(1)
    barrier()
    t0 = time.perf_counter()
    step_marks = []
    for _ in range(args.steps):
        res = step()
        step_marks.append(time.perf_counter())  # (steps end with a stream sync inside the library; marks are informational)
    barrier()
    dt = time.perf_counter() - t0
    if rank == 0 and os.environ.get("PDX_BENCH_VERBOSE"):
        prev = t0
        for i, m in enumerate(step_marks):
            print(f"step {i}: {(m - prev) * 1e3:.2f} ms", file=sys.stderr)
            prev = m
    lib.pdx_profile_enable(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = profile_report(lib)

    ms_per_step = dt / args.steps * 1e3
    value = n_total / (dt / args.steps) / 1e9

    # dominant kernel family (by device time inside the timed region), priced against the HBM roofline with the
    # ALGORITHMIC bytes of the rows one launch processes (16 B/row x local rows)
    dom, roof = None, None
    if prof:
        dom = max(prof.items(), key=lambda kv: kv[1][1])
        tag, (cnt, ms) = dom
        avg_ms = ms / cnt
        achieved = ALGO_BYTES_PER_ROW * n_local / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": tag, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "launches_per_step": cnt / args.steps, "avg_launch_ms": avg_ms,
                "whole_step_frac": ALGO_BYTES_PER_ROW * n_local / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "kernel_ms_per_step": {k: round(v[1] / args.steps, 3) for k, v in sorted(prof.items())}}

    # HBM traffic of the dominant kernel: PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate passes) collected on the same
    # build by tools/collect_profiles.sh and committed under profiles/ -- only valid for the configuration it was measured on
    if roof is not None:
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                pmc = json.load(f)
            if pmc.get("rows") == n_total and pmc.get("n_gpus") == world:
                roof["traffic"] = pmc["by_bench_tag_hbm_bytes_per_launch"].get(roof["kernel"])
                roof["traffic_source"] = "profiles/r01_pmc_traffic.json"
        except (OSError, ValueError, KeyError):
            pass

    # size-independent checks on the last result (outside the timed region)
    check = None
    if not args.no_check:
        if world == 1:
            gb, outs = res
            G = gb.num_groups
            cnt = outs[2].values[:G]
            sm, mean = outs[0].values[:G], outs[1].values[:G]
            ok_counts = int(cnt.sum().item()) == n_local
            ok_mean = bool(torch.equal(mean, sm / cnt.to(torch.float64)))
            fr = gb.first_rows()
            ok_order = bool((fr[1:] > fr[:-1]).all().item()) if G > 1 else True
            check = {"groups": G, "counts_sum_to_rows": ok_counts, "mean_is_sum_over_count": ok_mean, "first_occurrence_order": ok_order}
        else:
            check = pdist.check_result(res, n_total)
        if rank == 0 and not all(v for k, v in check.items() if isinstance(v, bool)):
            raise SystemExit(f"result check failed: {check}")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        def gpu_on_sample(m):
            gb = K.GroupByHandle.create(keys.slice(0, m))
            o = gb.agg(vals.slice(0, m), kinds)
            return (gb.unique_keys().to_numpy()[0], o[0].to_numpy()[0], o[1].to_numpy()[0], o[2].to_numpy()[0])

        cpu = cpu_baseline(int(min(args.cpu_sample_rows, n_total)), nkeys, gpu_on_sample)
        if cpu.get("gpu_matches_oracle_bit_exact") is False:
            raise SystemExit("HIP result differs from the oracle on the cpu_baseline sample")

    if rank == 0:
        line = {
            "metric": "Grows/sec hash group-by-sum, 1e9 int64 rows x 1e6 keys", "value": value, "unit": "Grows/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"group_by(int64 key).agg(sum,mean,count), {n_total:.3g} rows / {nkeys:.3g} keys"
                                   + (", 1 GPU" if world == 1 else f", row-range sharded over {world} GPUs (RCCL exchange)"),
                       "rows": n_total, "keys": nkeys, "rows_per_gpu": n_local, "parity": "bit-exact vs Arrow-order pairwise sum"},
            "roofline": roof, "cpu_baseline": cpu, "check": check,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
