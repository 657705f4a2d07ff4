#!/usr/bin/env python3
"""bench.py -- headline benchmark: hash group-by (sum, mean, count) of 1e9 int64-key rows x 1e6 keys (BASELINE.json
configs[2], "group_by(int64 key).agg(sum,mean,count), 1e9 rows / 1e6 keys, 1 GPU"; configs[3] at N > 1).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full pass of the hot path over the resident synthetic columns: pdx_groupby_create (keys -> first-occurrence
group ids) + pdx_groupby_agg(sum, mean, count) (stable sort by group + Arrow-order pairwise reduce), inputs already in HBM,
results left in HBM.  Rank 0 prints ONE JSON line:

  value / ms_per_step  whole-job rows per second of the timed region (barrier + synchronize on both sides, MAX over ranks)
  roofline             bound hbm; `achieved` / `frac` = ALGORITHMIC bytes of the step (16 B/row x local rows) / step time
                       / 8 TB/s -- the whole step, every pass included.  `dominant_kernel` prices the kernel family with the most
                       device time (HIP events recorded by the library on the launch stream) two ways: `achieved` with the whole
                       operator's 16 B/row per launch (the contract's definition; it flatters a multi-pass step) and `own_bytes`
                       with the bytes that kernel itself has to move.  `traffic` = HBM bytes per launch of that kernel from
                       rocprofv3 PMC counters (profiles/r03_pmc_traffic.json), dropped when the library sources changed since.
  cpu_baseline         the reference's Arrow call sequence on this box's host cores: Arrow C++ itself when the pyarrow wheel's
                       libarrow is present (oracle/_build/arrow_seq, kind "port": same kernels the reference calls, without its
                       unordered_map<ScalarPtr> bookkeeping), else the oracle's C restatement.
  secondary            (N = 1) the other BASELINE configs and the general-keys (hash table) form of the headline, each with
                       its own algorithmic-bytes roofline fraction; not part of `value`.
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec)
ALGO_BYTES_PER_ROW = 16.0  # 8 B int64 key + 8 B fp64 value per row
# bytes one launch of a kernel family has to move per row of its input (reads + writes)
OWN_BYTES_PER_ROW = {"radix_scatter": 20.5,   # narrowing passes: (4 + 8 -> 2 + 8) and (2 + 8 -> 1 + 8), mean of the two
                     "dense_slots": 12.0,     # 8 B key read, 4 B slot written
                     "fused_last_digit_reduce": 9.0,  # 1 B key + 8 B value read once
                     "seg_reduce": 8.0, "hash_probe_lds": 12.0, "radix_hist": 2.0}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=float, default=1e9, help="total rows (strong scaling: split by row range over the ranks)")
    ap.add_argument("--keys", type=float, default=1e6)
    ap.add_argument("--cpu-sample-rows", type=float, default=1e8,
                    help="rows of the same workload timed on the host cores (rank 0, N=1): SURVEY 8d's 1e8 rows / 1e6 keys = 100 rows per group, "
                         "~10-15 s of Arrow C++ (3e6 per-group CallFunction dispatches + the single-threaded ApplyGroupings)")
    ap.add_argument("--chain-check-rows", type=float, default=1.35e8,
                    help="rows of the same workload pushed through the HIP path and the oracle after the timed region: enough for the default "
                         "thresholds to pick the SAME kernel chain as the timed step (narrowing sort + fused last digit need >= 8192 rows per run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other configs / the general-keys run after the timed region")
    ap.add_argument("--no-check", action="store_true", help="skip the size-independent result checks after the timed region")
    return ap.parse_args()


def source_hash():
    """sha256 over the library sources: ties a PMC traffic file to the build it was measured on."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "pandasarrow_amd", "csrc", "*.h*")) + [os.path.join(ROOT, "include", "pdx", "abi.h")]):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


TRAFFIC_FILES = ("r04_pmc_traffic.json", "r04_pmc_traffic_general.json")


def load_traffic(slots, n_total, world):
    """The counter-traffic file under profiles/ that was measured on THIS configuration: same rows, GPU count, key -> slot plan
    ('dense' / 'hash_lds') and library sources.  None when there is none (a stale or foreign number is worse than none)."""
    for name in TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                pmc = json.load(f)
        except (OSError, ValueError):
            continue
        if pmc.get("rows") == n_total and pmc.get("n_gpus") == world and pmc.get("slots") == slots and pmc.get("source_hash") == source_hash():
            pmc["file"] = "profiles/" + name
            return pmc
    return None


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
    """The reference's Arrow-CPU call sequence on this host (Consume -> MakeGroupings -> ApplyGroupings x3 -> per-group
    sum / mean / count with a thread pool over the groups, src/dataframe.cpp:1539-1600 + src/pd_core_macros.h:5-147), timed on a
    bounded sample of the same workload.  The same sample also goes through the HIP path and is compared bit-for-bit."""
    import numpy as np
    import oracle as orc

    ncpu = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = ncpu
    threads = max(1, min(usable, 64))
    out = {"unit": "Grows/s", "cores": ncpu, "threads": threads, "usable_cores": usable, "kind": "port",
           "rows_per_group": round(sample_rows / max(1, min(nkeys, sample_rows)), 1)}
    arrow = None
    try:
        arrow = orc.arrow_seq_run(sample_rows, nkeys, threads)  # Arrow C++ itself (pyarrow wheel's libarrow), if it could be built
    except Exception as e:  # noqa: BLE001  (no libarrow / headers on this box: the C restatement below is the baseline)
        out["arrow_seq_unavailable"] = str(e)[:200]
    keys = orc.synth_keys(0, sample_rows, nkeys)
    vals = orc.synth_vals(0, sample_rows)
    t0 = time.perf_counter()
    uk, s, m, c = orc.groupby_sum_mean_count(keys, vals, nthreads=threads)
    dt = time.perf_counter() - t0
    port = {"value": sample_rows / dt / 1e9, "seconds": dt, "what": "oracle/pdx_oracle.c orc_groupby_sum_mean_count (C restatement, OpenMP over groups)"}
    if arrow is not None:
        # per-key comparison: Arrow's Grouper hands out ids per 1024-row mini-batch of its swiss table, and keys that collide
        # inside a mini-batch get their ids after the others, so on large inputs its group ORDER is first-occurrence only
        # approximately; the per-key aggregates are what must be bit-identical
        oa, oo = np.argsort(arrow["keys"], kind="stable"), np.argsort(uk, kind="stable")
        same = bool(len(uk) == len(arrow["keys"]) and np.array_equal(arrow["keys"][oa], uk[oo])
                    and np.array_equal(arrow["sum"][oa].view(np.uint64), s[oo].view(np.uint64))
                    and np.array_equal(arrow["mean"][oa].view(np.uint64), m[oo].view(np.uint64)) and np.array_equal(arrow["count"][oa], c[oo]))
        out.update(value=sample_rows / arrow["seconds"] / 1e9, engine=f"Arrow C++ {arrow['arrow_version']} (Grouper::Consume/MakeGroupings/ApplyGroupings + "
                   "CallFunction(sum|mean|count) per group)", phases_s=arrow["phases"], arrow_matches_oracle_bit_exact=same,
                   arrow_group_order_is_first_occurrence=bool(len(uk) == len(arrow["keys"]) and np.array_equal(arrow["keys"], uk)), c_restatement=port,
                   sample=f"first {sample_rows:.3g} rows of the same synthetic workload ({len(uk)} groups), {arrow['seconds']:.1f} s")
    else:
        out.update(value=port["value"], engine=port["what"],
                   sample=f"first {sample_rows:.3g} rows of the same synthetic workload ({len(uk)} groups), {dt:.1f} s")
    # context only: Arrow's own multi-threaded hash aggregate (different summation order than the reference's per-group sum)
    try:
        import pyarrow as pa

        rows_pa = min(sample_rows, 100_000_000)
        tbl = pa.table({"k": keys[:rows_pa], "v": vals[:rows_pa]})
        t1 = time.perf_counter()
        tbl.group_by("k").aggregate([("v", "sum"), ("v", "mean"), ("v", "count")])
        out["arrow_hash_aggregate_Grows_per_s"] = rows_pa / (time.perf_counter() - t1) / 1e9
    except Exception:  # noqa: BLE001
        pass
    if gpu_check is not None:
        gk, gs, gm, gc = gpu_check(sample_rows)
        out["gpu_matches_oracle_bit_exact"] = bool(np.array_equal(gk, uk) and np.array_equal(gs.view(np.uint64), s.view(np.uint64))
                                                   and np.array_equal(gm.view(np.uint64), m.view(np.uint64)) and np.array_equal(gc, c))
    return out


def secondary_runs(torch, L, K, api, keys, vals, kinds, n_total, nkeys, steps):
    """The other BASELINE configs + the general-keys form of the headline, N = 1, after the timed region.  Each entry: ms (median
    of `reps`), Grows/s, algorithmic GB/s (SURVEY 8d bytes) and its fraction of the 8 TB/s roofline."""
    out = {}

    def timeit(fn, reps=3, warm=1):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2]

    def put(name, rows, algo_bytes, dt, **extra):
        gbs = algo_bytes / dt / 1e9
        out[name] = dict(rows=rows, ms=round(dt * 1e3, 3), Grows_per_s=round(rows / dt / 1e9, 3), algo_GBps=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4), **extra)

    def gb_step():
        gb = K.GroupByHandle.create(keys)
        return gb, gb.agg(vals, kinds)

    # (a) the same step when the keys may NOT use the dense-integer shortcut: hash partition + LDS-resident open addressing
    prev_dense = os.environ.get("PDX_GROUPBY_DENSE")
    os.environ["PDX_GROUPBY_DENSE"] = "0"
    try:
        dt_hash = timeit(gb_step, reps=max(2, min(steps, 3)))
        gbh, _ = gb_step()
        trg = load_traffic(gbh.last_plan().get("slots"), n_total, 1)
        put("groupby_general_keys_hash_path", n_total, ALGO_BYTES_PER_ROW * n_total, dt_hash, plan=gbh.last_plan(),
            traffic=None if trg is None else trg.get("step_hbm_bytes"), traffic_source=None if trg is None else trg["file"],
            workload=f"same {n_total:.3g} rows / {nkeys:.3g} keys with PDX_GROUPBY_DENSE=0 (every key through the hash table)")
        del gbh
    finally:  # (the caller's own setting comes back: a general-keys collection run must stay general for the rows below)
        if prev_dense is None:
            del os.environ["PDX_GROUPBY_DENSE"]
        else:
            os.environ["PDX_GROUPBY_DENSE"] = prev_dense

    # (a2) the reference's API has no multi-kind call: user code is gb.sum(c); gb.mean(c); gb.count(c) (src/group_by.h:85-139) on the
    # per-group arrays its constructor built once (processEach, src/dataframe.cpp:1539-1554).  Same here with a bound column: the
    # first call sorts by group and reduces, the other two are served from the handle.
    def gb_three_calls():
        gb = K.GroupByHandle.create(keys)
        gb.bind(vals)
        return gb, [gb.agg(vals, [k])[0] for k in kinds]

    dt3 = timeit(gb_three_calls, reps=max(2, min(steps, 3)))
    dt1 = timeit(gb_step, reps=max(2, min(steps, 3)))
    put("groupby_reference_api_three_calls", n_total, ALGO_BYTES_PER_ROW * n_total, dt3, fused_call_ms=round(dt1 * 1e3, 3),
        ratio_to_fused_call=round(dt3 / dt1, 3), workload="create + bind + sum(); mean(); count() as three pdx_groupby_agg calls")
    # (a3) the order-free kinds north_star names beside sum / mean -- min / max / count (and int64 sum) -- on an EXISTING handle (the
    # reference constructs the GroupBy once, then calls gb.min(c), gb.max(c), gb.count(c): src/group_by.h:85-139): no value sort, one
    # partition pass + accumulators in LDS (gb_acc.hpp).  Algorithmic bytes: 4 B slot id + 8 B value per row (count: the slot ids alone).
    gb0 = K.GroupByHandle.create(keys)
    ikeys = K.Column(L.INT64, n_total, keys.values, None, 0, 0)  # an int64 value column that is already resident: the keys themselves
    os.environ["PDX_ACC_SIZES_CACHE"] = "0"  # (count of a column without nulls is otherwise served from the handle after the first call)
    try:
        for name, col, kk, bpr in (("groupby_min_max", vals, [L.AGG_MIN, L.AGG_MAX], 12.0), ("groupby_count", vals, [L.AGG_COUNT], 4.0),
                                   ("groupby_int64_sum", ikeys, [L.AGG_SUM], 12.0)):
            dta = timeit(lambda: gb0.agg(col, kk), reps=3)
            put(name, n_total, bpr * n_total, dta, plan=gb0.last_plan(), algo_bytes_per_row=bpr,
                workload=f"pdx_groupby_agg({name.split('_', 1)[1]}) on an existing handle, {n_total:.3g} rows / {nkeys:.3g} keys")
    finally:
        del os.environ["PDX_ACC_SIZES_CACHE"]
    del gb0, ikeys
    # (b) 5 % null values (SURVEY 8d secondary run)
    vmask = K.compare(L.NE, K.synth_keys(3, n_total, 20), 0)
    vn = K.Column(L.FLOAT64, n_total, vals.values, vmask.values, 0, -1)

    def gb_nulls():
        gb = K.GroupByHandle.create(keys)
        return gb.agg(vn, kinds)

    put("groupby_5pct_null_values", n_total, (ALGO_BYTES_PER_ROW + 0.125) * n_total, timeit(gb_nulls, reps=2))
    del vn, vmask
    # (b2) skewed keys: one key holds 5 % of the rows (its run of the fused layout is far over the kernels' limit: side form)
    hot = K.Column(keys.dtype, n_total, keys.values.clone(), None, 0, 0)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    for s0 in range(0, n_total, 1 << 27):
        e0 = min(n_total, s0 + (1 << 27))
        hot.values[s0:e0][torch.rand(e0 - s0, device="cuda", generator=gen) < 0.05] = 12345
    plan_hot = {}

    def gb_hot():
        gb = K.GroupByHandle.create(hot)
        r = gb.agg(vals, kinds)
        plan_hot.update(gb.last_plan())
        return r

    dth = timeit(gb_hot, reps=2)
    put("groupby_one_hot_key_5pct", n_total, ALGO_BYTES_PER_ROW * n_total, dth, plan=dict(plan_hot),
        workload="the headline's keys with 5 % of the rows moved to one key")
    del hot
    # (c) C1: Series<double> add + sum at 1e6 rows (plumbing) and at the headline's row count
    for n in (1_000_000, n_total):
        x, y = K.synth_vals(0, n, 1), K.synth_vals(0, n, 2)
        # (at 1e6 rows a call is tens of microseconds of launch / sync latency: median of 200 calls, not of 3)
        reps, warm = (200, 20) if n <= 1e7 else (3, 1)
        put(f"C1_add_f64[{n:.0e}]", n, 24.0 * n, timeit(lambda: K.binary(L.ADD, x, y), reps, warm))
        put(f"C1_sum_f64[{n:.0e}]", n, 8.0 * n, timeit(lambda: K.aggregate(L.AGG_SUM, x), reps, warm))
        del x, y
    # (d) C2: DataFrame boolean-mask filter + take, 1e8 rows x 8 fp64 cols (+ index)
    n = min(100_000_000, n_total)
    cols = {f"c{j}": K.synth_vals(0, n, 20 + j) for j in range(8)}
    df = api.DataFrame(cols, index=K.synth_keys(0, n, 1 << 62))
    mask = df["c0"] > 0.5
    s = K.filter_count(mask.col) / n
    put("C2_filter_8cols+index", n, (0.125 + 8 * 9 * (1 + s)) * n, timeit(lambda: df.where(mask)), selectivity=round(s, 4))
    m = n // 2
    take_idx = api.Series(K.synth_keys(7, m, n))
    dt_take = timeit(lambda: df.take(take_idx))
    put("C2_take_8cols+index", m, (8 + 16 * 9) * m, dt_take, note="rows = output rows (random 8-B gathers)",
        random_gathers_per_s=round(9 * m / dt_take), sector_GBps_64B=round(9 * m * 64 / dt_take / 1e9, 1))
    del df, cols, mask, take_idx
    # (e) C5: resample('1min').mean() on a timestamp + fp64 Series (100 ms spacing -> 600 rows per bin)
    ts = K.synth_ts(0, n_total, 946_684_800 * 10**9, 100_000_000)
    ser = api.Series(vals, index=ts, name="v")
    put("C5_resample_1min_mean", n_total, 16.0 * n_total, timeit(lambda: ser.resample("1min").mean()))
    # (f) a12: DataFrame::downsample('1T') on the same axis = floor/ceil_temporal (16 B/row) + group-by of the binned labels
    put("a12_round_temporal_minute", n_total, 16.0 * n_total, timeit(lambda: K.round_temporal(ts, 1, L.UNIT_MINUTE, True, True, True)))
    df12 = api.DataFrame({"v": vals}, index=ts)
    put("a12_downsample_1T_mean", n_total, 16.0 * n_total, timeit(lambda: df12.downsample("1T").mean(), reps=2),
        note="end to end: ceil_temporal + group-by of the rounded labels (sorted axis: runs of equal labels, no value sort) + mean")
    del ser, ts, df12
    # (g) keys that arrive sorted (1000 consecutive rows per key): runs of equal keys, no dictionary, no value sort
    rows = K.synth_ts(0, n_total, 0, 1)  # 0, 1, 2, ...
    skeys = K.binary(L.DIV, K.Column(L.INT64, rows.length, rows.values, None, 0, 0), 1000, True)
    del rows

    def gb_sorted():
        gb = K.GroupByHandle.create(skeys)
        return gb.agg(vals, kinds)

    put("groupby_sorted_keys", n_total, ALGO_BYTES_PER_ROW * n_total, timeit(gb_sorted, reps=2), workload=f"{n_total:.3g} rows, keys = row // 1000")
    del skeys
    # (h) SURVEY 8(f)-3: argsort of 1e8 random float64 values (8 B read + 8 B written per row)
    m = min(100_000_000, n_total)
    sv = K.synth_vals(5, m, 77)
    put("8f3_argsort_f64", m, 16.0 * m, timeit(lambda: K.argsort(sv), reps=2))
    del sv
    # (i) SURVEY 8(f)-4: Parquet / Arrow IPC ingest straight into device columns (DataFrame::readParquet / readBinary,
    # src/dataframe.cpp:646-683, 754-791): host bytes of a pyarrow-written file -> pdx_parquet_load / pdx_ipc_load -> resident columns.
    # Rate = DECODED bytes (16 B/row: int64 key + fp64 value) per second, the host->device copy of the file included (PCIe bound, not HBM).
    try:
        import io

        import pyarrow as pa
        import pyarrow.parquet as pq

        m = min(20_000_000, n_total)
        hk, hv = keys.values[:m].cpu().numpy(), vals.values[:m].cpu().numpy()
        tbl = pa.table({"k": hk, "v": hv})
        sink = io.BytesIO()
        pq.write_table(tbl, sink, compression="snappy", use_dictionary=True, row_group_size=m)
        pq_blob = sink.getvalue()
        sink = pa.BufferOutputStream()
        with pa.ipc.new_stream(sink, tbl.schema) as w:
            w.write_table(tbl, max_chunksize=m)
        ipc_blob = sink.getvalue().to_pybytes()
        del tbl, sink

        def load_parquet():
            return K.ParquetFile(pq_blob).load()

        def load_ipc():
            return K.IpcFrame(ipc_blob).load()

        cols = load_parquet()
        same = bool(torch.equal(cols[0][1].values[:m], keys.values[:m]) and torch.equal(cols[1][1].values[:m].view(torch.int64), vals.values[:m].view(torch.int64)))
        del cols
        put("8f4_parquet_ingest_snappy_dict", m, 16.0 * m, timeit(load_parquet, reps=3), file_bytes=len(pq_blob), decoded_equals_source=same, bound="pcie + decode",
            note="pyarrow-written file (Snappy, dictionary with PLAIN fallback, one row group) -> device columns; frac is against HBM and not the bound here")
        cols = load_ipc()
        same = bool(torch.equal(cols[0][1].values[:m], keys.values[:m]) and torch.equal(cols[1][1].values[:m].view(torch.int64), vals.values[:m].view(torch.int64)))
        del cols
        put("8f4_ipc_ingest", m, 16.0 * m, timeit(load_ipc, reps=3), file_bytes=len(ipc_blob), decoded_equals_source=same, bound="pcie",
            note="Arrow IPC stream of one record batch: one host->device copy of the body, columns alias it")
        del pq_blob, ipc_blob, hk, hv
    except ImportError as e:  # no pyarrow on this box: nothing can write the files
        out["8f4_ingest_skipped"] = str(e)
    return out


def launch_ranks(n):
    """One child process per rank on 127.0.0.1 (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run sets them), same command
    line.  Rank 0's stdout is this process's stdout, so exactly one JSON line comes out.  A rank that fails takes the others down (by pid)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"[bench] rank {r} exited with status {code}: stopping the other ranks", file=sys.stderr)
                for o in alive:
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before anything initialises HIP/HSA (dmabuf IPC for RCCL)
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as a plain command: start the N ranks here -- fresh child processes, created before this process has
        # imported torch or touched the GPU (never a re-exec of a process that holds a GPU context) -- and exit with their status.
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # before any GPU call, so a launcher can still start the ranks cleanly
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks with torch.distributed.run (one per GPU), or drop "
                         "WORLD_SIZE from the environment and let bench.py start them")
    import torch
    import torch.distributed as dist

    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # PDX_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices, collectives
    # go through the host) -- never a measurement; the driver's runs use RCCL, one rank per GPU.
    # PDX_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, sharded step, collectives on the wire) at world size 1.
    backend = os.environ.get("PDX_BENCH_BACKEND", "nccl")
    sharded = world > 1 or os.environ.get("PDX_BENCH_FORCE_DIST") == "1"
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    # librccl prints a version banner on STDOUT when a communicator is created ("RCCL version : ..."): stdout must carry exactly one
    # JSON line, so file descriptor 1 points at stderr until the timed region is over (communicators are created lazily, at the first
    # collective) and is restored before rank 0 prints.
    saved_stdout = None
    if sharded:
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
    if sharded:
        if "MASTER_ADDR" not in os.environ:  # FORCE_DIST outside a launcher
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from pandasarrow_amd import _lib as L
    from pandasarrow_amd import api
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

    if not sharded:
        def step():
            gb = K.GroupByHandle.create(keys)
            outs = gb.agg(vals, kinds)
            return gb, outs
    else:
        from pandasarrow_amd import dist as pdist

        # The sharded step runs inside the library (csrc/dist.hip behind pdx_dist_*): orchestration, glue kernels and the RCCL calls
        # (ncclAllGather, grouped ncclSend / ncclRecv) -- python only creates the communicator.  PDX_BENCH_DIST=torch keeps the older
        # orchestration in pandasarrow_amd/dist.py over torch.distributed; it is also the fallback when the communicator cannot be
        # created (reported in config.path).  With a non-RCCL process group (rehearsals on one GPU) the library's custom transport
        # rides on that group.
        dist_path = "c-abi"
        cd = None
        if os.environ.get("PDX_BENCH_DIST", "c") != "torch":
            try:
                cd = pdist.CDist("rccl" if backend == "nccl" else "torch")
            except Exception as e:  # noqa: BLE001
                print(f"[bench] rank {rank}: pdx_dist communicator failed ({type(e).__name__}: {e})", file=sys.stderr)
            # the ranks must agree: the two orchestrations use different collectives, a split decision would deadlock the timed step
            okf = torch.tensor([1 if cd is not None else 0], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if int(okf.item()) == 0:
                if cd is not None:
                    cd.close()
                cd = None
                if rank == 0:
                    print("[bench] not every rank has a pdx_dist communicator: all ranks use the torch.distributed orchestration", file=sys.stderr)
        if cd is not None:
            def step():
                # sum/mean/count with the exact partial-tree exchange (fragments + aligned subtree nodes, no rows shipped)
                return cd.groupby_sum_mean_count(keys, vals, row_offset=lo)
        else:
            dist_path = "torch"
            engine = pdist.HipEngine()

            def step():
                return pdist.groupby_sum_mean_count_sharded(engine, keys, vals, row_offset=lo)

    def barrier():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    res = None
    late_inputs = None
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
    if sharded:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = profile_report(lib)

    ms_per_step = dt / args.steps * 1e3
    value = n_total / (dt / args.steps) / 1e9

    # roofline: the WHOLE step against the HBM peak with the operator's algorithmic bytes (this rank's shard); the dominant
    # kernel family (most device time inside the timed region, HIP events on the launch stream) is priced beside it
    step_gbs = ALGO_BYTES_PER_ROW * n_local / (ms_per_step * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": step_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_gbs / HBM_PEAK_GBS,
            "scope": "whole step: 16 B/row x local rows / ms_per_step", "traffic": None, "dominant_kernel": None}
    if prof:
        tag, (cnt, ms) = max(prof.items(), key=lambda kv: kv[1][1])
        avg_ms = ms / cnt
        dk = {"kernel": tag, "launches_per_step": cnt / args.steps, "avg_launch_ms": avg_ms,
              "achieved": ALGO_BYTES_PER_ROW * n_local / (avg_ms * 1e-3) / 1e9, "traffic": None}
        dk["frac"] = dk["achieved"] / HBM_PEAK_GBS
        if tag in OWN_BYTES_PER_ROW:
            own = OWN_BYTES_PER_ROW[tag] * n_local / (avg_ms * 1e-3) / 1e9
            dk["own_bytes"] = {"bytes_per_row": OWN_BYTES_PER_ROW[tag], "achieved": own, "frac": own / HBM_PEAK_GBS}
        roof["dominant_kernel"] = dk
        roof["kernel_ms_per_step"] = {k: round(v[1] / args.steps, 3) for k, v in sorted(prof.items())}
        roof["kernel_ms_sum"] = round(sum(v[1] for v in prof.values()) / args.steps, 3)
        # HBM traffic per launch from PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes, tools/collect_profiles.sh):
        # only valid for the configuration, the PLAN (dense slots / hash-partitioned slots run different passes) AND the sources it was
        # measured on (the sharded step -- also when forced at world size 1 -- runs other passes: never attached)
        slots = None
        if not sharded and res is not None:
            slots = res[0].last_plan().get("slots")
        tr = None if sharded else load_traffic(slots, n_total, world)
        if tr is not None:
            dk["traffic"] = tr["by_bench_tag_hbm_bytes_per_launch"].get(tag)
            roof["traffic"] = tr.get("step_hbm_bytes")
            roof["traffic_source"] = "%s (slots=%s, source_hash %s)" % (tr["file"], tr.get("slots"), tr["source_hash"])
        elif not sharded:
            roof["traffic_note"] = "no counter traffic under profiles/ for this configuration, plan (slots=%s) and build: dropped" % slots

    # size-independent checks on the last result (outside the timed region)
    check = None
    if not args.no_check:
        if not sharded:
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
            # (on by default in the rehearsal modes -- gloo ranks sharing a GPU, RCCL forced at world size 1 -- and with
            #  PDX_BENCH_CROSSCHECK=1; a real multi-rank RCCL run does not start a second, never-exercised orchestration by default:
            #  the properties above already hold the result to account, and a stall there would cost the whole measurement)
            # A real multi-rank RCCL run does the same cross-check AFTER rank 0 has printed the JSON line (below): the library's hand-declared
            # RCCL binding has never run with more than one rank on this pool, so its first run must be verified -- but a stall in the second
            # orchestration must not cost the measurement.  PDX_BENCH_CROSSCHECK=0 / 1 forces it off / into the line.
            rehearsal = backend != "nccl" or world == 1
            crosscheck = os.environ.get("PDX_BENCH_CROSSCHECK", "1" if rehearsal else "late")
            late_crosscheck = cd is not None and crosscheck == "late"
            if late_crosscheck:
                late_inputs = (res, keys, vals, lo)
            if cd is not None and crosscheck == "1":
                # the library's orchestration (raw RCCL calls) against the older one over torch.distributed's collectives, same shards:
                # keys, first rows, sums, means and counts bit for bit, on every rank (outside the timed region)
                ref = pdist.groupby_sum_mean_count_sharded(pdist.HipEngine(), keys, vals, row_offset=lo)
                same = (int(ref["G"]) == int(res["G"]) and torch.equal(ref["keys"], res["keys"]) and torch.equal(ref["first_rows"], res["first_rows"])
                        and all(torch.equal(a[0].view(torch.int64), b[0].view(torch.int64)) for a, b in zip(ref["outs"], res["outs"])))
                flag = torch.tensor([1 if same else 0], dtype=torch.int64, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                check["c_abi_matches_torch_orchestration_all_ranks"] = bool(flag.item())
                ref = None
        if rank == 0 and not all(v for k, v in check.items() if isinstance(v, bool)):
            raise SystemExit(f"result check failed: {check}")
    res = None

    secondary = None
    if rank == 0 and not sharded and not args.no_secondary:
        lib.pdx_trim_pool()
        secondary = secondary_runs(torch, L, K, api, keys, vals, kinds, n_total, nkeys, args.steps)

    cpu = None
    if rank == 0 and not sharded and not args.no_cpu_baseline:
        def gpu_on_sample(m):
            gb = K.GroupByHandle.create(keys.slice(0, m))
            o = gb.agg(vals.slice(0, m), kinds)
            return (gb.unique_keys().to_numpy()[0], o[0].to_numpy()[0], o[1].to_numpy()[0], o[2].to_numpy()[0])

        cpu = cpu_baseline(int(min(args.cpu_sample_rows, n_total)), nkeys, gpu_on_sample)
        if cpu.get("gpu_matches_oracle_bit_exact") is False:
            raise SystemExit("HIP result differs from the oracle on the cpu_baseline sample")

    # the timed kernel chain against the oracle: a prefix of the same workload long enough for the default thresholds to choose the
    # same path as the full-size step (asserted through the plan), compared per key bit for bit
    chain = None
    if rank == 0 and not sharded and not args.no_check and not args.no_cpu_baseline:
        import numpy as np
        import oracle as orc

        m = int(min(args.chain_check_rows, n_total))
        gbf = K.GroupByHandle.create(keys)
        gbf.agg(vals, kinds)
        full_plan = gbf.last_plan()
        del gbf
        gbc = K.GroupByHandle.create(keys.slice(0, m))
        oc = gbc.agg(vals.slice(0, m), kinds)
        ek, es, em, ec = orc.groupby_sum_mean_count(orc.synth_keys(0, m, nkeys), orc.synth_vals(0, m), nthreads=min(64, os.cpu_count() or 1))
        same = bool(np.array_equal(gbc.unique_keys().to_numpy()[0], ek) and np.array_equal(oc[0].to_numpy()[0].view(np.uint64), es.view(np.uint64))
                    and np.array_equal(oc[1].to_numpy()[0].view(np.uint64), em.view(np.uint64)) and np.array_equal(oc[2].to_numpy()[0], ec))
        chain = {"rows": m, "plan": gbc.last_plan(), "timed_step_plan": full_plan, "same_path_as_timed_step": gbc.last_plan() == full_plan,
                 "gpu_matches_oracle_bit_exact": same}
        if not same:
            raise SystemExit(f"HIP result differs from the oracle on the production-chain sample: {chain}")

    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        line = {
            "metric": "Grows/sec hash group-by-sum, 1e9 int64 rows x 1e6 keys", "value": value, "unit": "Grows/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"group_by(int64 key).agg(sum,mean,count), {n_total:.3g} rows / {nkeys:.3g} keys"
                                   + (", 1 GPU" if world == 1 else f", row-range sharded over {world} GPUs (RCCL exchange)"),
                       "rows": n_total, "keys": nkeys, "rows_per_gpu": n_local, "parity": "bit-exact vs Arrow-order pairwise sum",
                       "path": ("sharded-" + dist_path) if sharded else "single"},
            "roofline": roof, "cpu_baseline": cpu, "check": check, "chain_check": chain,
            # the metric's literal reading ("hash group-by"): the same step with every key through the LDS-bucketed hash table
            "general_keys_hash_path": (secondary or {}).get("groupby_general_keys_hash_path"),
            "secondary": secondary,
        }
        print(json.dumps(line), flush=True)
    if late_inputs is not None:
        # first multi-rank RCCL run of this build: the library's orchestration against the older one over torch.distributed's collectives,
        # same shards, bit for bit on every rank -- after the line is out; the verdict goes to stderr and into the exit status
        lres, lkeys, lvals, llo = late_inputs
        # (a watchdog: the line is out; a stall of this second orchestration must end the run, not hang it until the driver's limit)
        import threading

        def _give_up():
            print("[bench] late cross-check did not finish within 120 s: abandoned (the JSON line above stands)", file=sys.stderr, flush=True)
            os._exit(0)

        watchdog = threading.Timer(float(os.environ.get("PDX_BENCH_LATE_CHECK_LIMIT_S", "120")), _give_up)
        watchdog.daemon = True
        watchdog.start()
        ref = pdist.groupby_sum_mean_count_sharded(pdist.HipEngine(), lkeys, lvals, row_offset=llo)
        same = (int(ref["G"]) == int(lres["G"]) and torch.equal(ref["keys"], lres["keys"]) and torch.equal(ref["first_rows"], lres["first_rows"])
                and all(torch.equal(a[0].view(torch.int64), b[0].view(torch.int64)) for a, b in zip(ref["outs"], lres["outs"])))
        flag = torch.tensor([1 if same else 0], dtype=torch.int64, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            print(f"[bench] cross-check of the C-ABI orchestration (raw RCCL calls) against torch.distributed's collectives on all {world} ranks: "
                  + ("bit-identical" if flag.item() else "MISMATCH"), file=sys.stderr, flush=True)
        watchdog.cancel()
        if not flag.item():
            raise SystemExit(3)
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
