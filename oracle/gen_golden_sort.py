#!/usr/bin/env python3
"""Golden vectors for pdx_argsort: Arrow 25.0.0's array_sort_indices (the kernel Series::argsort / Series::sort call,
src/series.cpp:864-868, 978-992) on fixed-seed inputs -> tests/golden/sort_golden.npz.  Run here (pyarrow present); the file
is data and travels with the repo."""
import os
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

rng = np.random.default_rng(20261004)
out = {}
cases = []
for n in (0, 1, 2, 17, 1000, 40_003):
    for dt in ("f64", "i64", "u64"):
        for nulls in (False, True):
            cases.append((n, dt, nulls))
for ci, (n, dt, nulls) in enumerate(cases):
    if dt == "f64":
        v = rng.standard_normal(n)
        v[rng.random(n) < 0.1] = np.nan
        v[rng.random(n) < 0.1] = 0.0
        v[rng.random(n) < 0.1] = -0.0
        v[rng.random(n) < 0.05] = np.inf
        v[rng.random(n) < 0.05] = -np.inf
        if n:
            v[rng.integers(0, n, n // 3)] = np.round(v[rng.integers(0, n, n // 3)], 1)  # duplicates
    elif dt == "i64":
        v = rng.integers(-50, 50, n).astype(np.int64)
        if n > 4:
            v[:2] = [np.iinfo(np.int64).max, np.iinfo(np.int64).min]
    else:
        v = rng.integers(0, 100, n).astype(np.uint64)
        if n > 4:
            v[:2] = [np.iinfo(np.uint64).max, 0]
    valid = (rng.random(n) > 0.15) if nulls else np.ones(n, bool)
    arr = pa.array(v, mask=~valid)
    name = f"sort_{dt}_{n}_{'nulls' if nulls else 'dense'}"
    out[name + "_v"] = v
    out[name + "_valid"] = valid
    out[name + "_asc"] = pc.array_sort_indices(arr, order="ascending").to_numpy().astype(np.uint64)
    out[name + "_desc"] = pc.array_sort_indices(arr, order="descending").to_numpy().astype(np.uint64)
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sort_golden.npz")
np.savez_compressed(path, **out)
print(len(cases), "cases ->", path, os.path.getsize(path), "bytes")
