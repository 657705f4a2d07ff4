#!/usr/bin/env python3
"""Per-rank time of the sharded order-free group aggregates (pdx_dist_groupby_order_free, RCCL forced at world size 1) at a shard's size.
Usage: python tools/time_sharded_order_free.py [rows] [keys]"""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ["PDX_DIST_FORCE_COLLECTIVES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from pandasarrow_amd import _lib as L, column as K, dist as pdist
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 125_000_000
nk = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
L.check(L.load().pdx_init(0))
dist.init_process_group("nccl", rank=0, world_size=1)
cd = pdist.CDist("rccl")
keys, vals = K.synth_keys(0, n, nk), K.synth_vals(0, n)
for kinds, name in (([2, 3], "min+max"), ([4], "count"), ([2, 3, 4], "min+max+count")):
    for _ in range(2):
        cd.groupby_order_free(keys, vals, kinds)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        cd.groupby_order_free(keys, vals, kinds)
    torch.cuda.synchronize()
    print(f"{name:14s} rows={n:.3g}: {(time.perf_counter() - t0) / 5 * 1e3:7.2f} ms/step", flush=True)
gb = K.GroupByHandle.create(keys)
for _ in range(2):
    K.GroupByHandle.create(keys).agg(vals, [2, 3, 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    K.GroupByHandle.create(keys).agg(vals, [2, 3, 4])
torch.cuda.synchronize()
print(f"single-GPU create + min/max/count: {(time.perf_counter() - t0) / 5 * 1e3:7.2f} ms/step")
dist.destroy_process_group()
