#!/usr/bin/env python3
"""Parquet / Arrow IPC ingest into device columns (pdx_parquet_load / pdx_ipc_load), pyarrow-written files of `rows` rows x (int64 key,
fp64 value).  python tools/bench_ingest.py [rows] [compression] [dictionary 0|1]"""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pyarrow as pa, pyarrow.parquet as pq
from pandasarrow_amd import _lib as L, column as K
m = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
comp = sys.argv[2] if len(sys.argv) > 2 else "snappy"
use_dict = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
L.check(L.load().pdx_init(0))
keys, vals = K.synth_keys(0, m, 1_000_000), K.synth_vals(0, m)
tbl = pa.table({"k": keys.values[:m].cpu().numpy(), "v": vals.values[:m].cpu().numpy()})
sink = io.BytesIO()
t0 = time.perf_counter()
pq.write_table(tbl, sink, compression=None if comp == "none" else comp, use_dictionary=use_dict, row_group_size=m)
blob = sink.getvalue()
print(f"pyarrow wrote {len(blob) / 1e6:.1f} MB in {time.perf_counter() - t0:.2f} s ({comp}, dictionary={use_dict})", flush=True)
t0 = time.perf_counter(); pq.read_table(io.BytesIO(blob)); print(f"pyarrow read_table (host, {os.cpu_count()} cores): {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
for _ in range(2):
    K.ParquetFile(blob).load()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); f = K.ParquetFile(blob); t1 = time.perf_counter(); cols = f.load(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t2 - t0, t1 - t0))
ts.sort()
dt, dopen = ts[len(ts) // 2]
ok = torch.equal(cols[0][1].values[:m], keys.values[:m]) and torch.equal(cols[1][1].values[:m].view(torch.int64), vals.values[:m].view(torch.int64))
print(f"pdx_parquet_open + load: {dt * 1e3:.1f} ms (open {dopen * 1e3:.2f} ms)  {16 * m / dt / 1e9:.2f} GB/s decoded  equal={ok}", flush=True)
