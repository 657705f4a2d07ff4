#!/usr/bin/env python3
"""Whole-column pairwise sum of a float64 column with 5 % nulls, 1e9 rows: for rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pandasarrow_amd import _lib as L, column as K
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
L.check(L.load().pdx_init(0))
vals = K.synth_vals(0, n)
valid = K.compare(L.NE, K.synth_keys(3, n, 20), 0)
col = K.Column(L.FLOAT64, n, vals.values, valid.values, 0, -1)
for _ in range(2): K.aggregate(L.AGG_SUM, col)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): r = K.aggregate(L.AGG_SUM, col)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"sum, 5% nulls: {dt*1e3:.2f} ms  {8*n/dt/1e12:.2f} TB/s algorithmic", r)
