#!/usr/bin/env python3
"""Order-free group aggregates (count / min / max / int64 sum) on an existing handle: gb_acc.hpp against the sorted-layout path
(PDX_GROUPBY_ACC=0).  python tools/bench_acc.py [rows] [keys]; PDX_GROUPBY_DENSE=0 for the hash-partitioned slots."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pandasarrow_amd import _lib as L, column as K
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
nk = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
lib = L.load()
L.check(lib.pdx_init(0))
keys, vals = K.synth_keys(0, n, nk), K.synth_vals(0, n)
hot_share = float(os.environ.get("PDX_BENCH_HOT_SHARE", "0"))  # this share of the rows moves to ONE key
if hot_share > 0:
    sel = K.synth_keys(5, n, 1000).values < int(hot_share * 1000)
    keys = K.Column(L.INT64, n, torch.where(sel, 4242, keys.values), None, 0, 0)
    del sel
ivals = K.Column(L.INT64, n, keys.values, None, 0, 0)
valid = K.compare(L.NE, K.synth_keys(3, n, 20), 0)
vals_n = K.Column(L.FLOAT64, n, vals.values, valid.values, 0, -1)
MIN, MAX, COUNT, SUM = L.AGG_MIN, L.AGG_MAX, L.AGG_COUNT, L.AGG_SUM
t0 = time.perf_counter(); gb = K.GroupByHandle.create(keys); torch.cuda.synchronize()
gb = K.GroupByHandle.create(keys); torch.cuda.synchronize(); t0 = time.perf_counter(); gb = K.GroupByHandle.create(keys); torch.cuda.synchronize()
print(f"create: {(time.perf_counter() - t0) * 1e3:.2f} ms, G = {gb.num_groups}", flush=True)
os.environ["PDX_ACC_SIZES_CACHE"] = "0"
cases = [("min+max f64", vals, [MIN, MAX]), ("min f64", vals, [MIN]), ("count", vals, [COUNT]), ("int64 sum", ivals, [SUM]),
         ("int64 sum+min+max+count", ivals, [SUM, MIN, MAX, COUNT]), ("min+max+count 5% nulls", vals_n, [MIN, MAX, COUNT]), ("count 5% nulls", vals_n, [COUNT])]
for acc in ("1", "0"):
    os.environ["PDX_GROUPBY_ACC"] = acc
    for name, col, kk in cases:
        for _ in range(2): gb.agg(col, kk)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): gb.agg(col, kk)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        lib.pdx_profile_reset(); lib.pdx_profile_enable(1)
        gb.agg(col, kk); torch.cuda.synchronize()
        lib.pdx_profile_enable(0)
        buf = C.create_string_buffer(1 << 16)
        L.check(lib.pdx_profile_report(buf, len(buf)))
        ks = "  ".join(f"{t}={float(ms):.2f}" for t, c, ms in (l.split() for l in buf.value.decode().splitlines()) if float(ms) >= 0.05)
        print(f"acc={acc} {name:28s} {dt*1e3:7.2f} ms  {n/dt/1e9:6.1f} Grows/s  [{gb.last_plan().get('reducer')}]  {ks}", flush=True)
