#!/usr/bin/env python3
"""Times the per-rank work of the sharded group-by on ONE GPU (W = 1): what a rank of an 8-GPU run spends outside the
collectives.  Usage: python tools/time_sharded_local.py [rows] [keys]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pandasarrow_amd import _lib as L, column as K, dist as pdist

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 125_000_000
nk = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
L.check(L.load().pdx_init(0))
keys, vals = K.synth_keys(0, n, nk), K.synth_vals(0, n)
eng = pdist.HipEngine()
for name, fn in (("partial-tree", lambda: pdist.groupby_sum_mean_count_sharded(eng, keys, vals)),
                 ("row-exchange", lambda: pdist.groupby_agg_sharded(eng, keys, vals, [0, 1, 4])),
                 ("single-gpu", lambda: K.GroupByHandle.create(keys).agg(vals, [0, 1, 4]))):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    print(f"{name:14s} rows={n:.3g}: {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms/step")

os.environ["PDX_DIST_TIMING"] = "1"
pdist.TIMING.clear()
for _ in range(3):
    res = pdist.groupby_sum_mean_count_sharded(eng, keys, vals)
print("partial-tree stages (ms/step, synchronised):", {k: round(v / 3, 2) for k, v in sorted(pdist.TIMING.items())}, "records:", res["records"])
