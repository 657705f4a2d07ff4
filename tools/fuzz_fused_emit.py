#!/usr/bin/env python3
"""One-off differential fuzz of the fused record emission (k_flr_emit) through pdx_groupby_sum_mean_count_chunked: dense keys, chunks of more
than 2^22 rows (so every chunk but the first has non-zero prefixes), random cardinalities / chunk sizes / hot keys / NaN and inf values, against
the C oracle bit for bit.  Usage: python tools/fuzz_fused_emit.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as orc
from pandasarrow_amd import _lib as L, dist as pdist
from pandasarrow_amd.column import Column
L.check(L.load().pdx_init(0))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    nk = int(rng.choice([140_000, 260_000, 520_000, 1_000_000, 2_000_000]))
    chunk = int(rng.integers(4_200_000, 4_700_000))
    n = int(chunk * rng.integers(2, 4) + rng.integers(0, chunk))
    used = nk if rng.random() < 0.6 else int(rng.integers(500, 20_000))  # few large groups inside a wide dense domain
    keys = rng.integers(0, used, n).astype(np.int64) * (nk // used) + int(rng.integers(-50, 50))
    if rng.random() < 0.3:
        keys[rng.random(n) < float(rng.uniform(0.005, 0.04))] = int(keys[0])  # a warm key (short of a skewed run)
    vals = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4, n)
    m = rng.integers(0, n, n // 3000)
    vals[m] = rng.choice(np.array([np.nan, -np.nan, np.inf, -np.inf, 0.0, -0.0]), m.size)
    res = pdist.groupby_sum_mean_count_chunked(Column.from_numpy(keys), Column.from_numpy(vals), chunk)
    ids, uniq, _, firstrow = orc.group_ids(keys, None)
    ok = res["G"] == len(uniq) and np.array_equal(res["keys"].cpu().numpy(), uniq) and np.array_equal(res["first_rows"].cpu().numpy(), firstrow)
    for j, kind in enumerate((0, 1, 4)):
        exp = orc.groupby_agg(kind, ids, len(uniq), vals, nthreads=8)[0]
        got = res["outs"][j][0].cpu().numpy()
        ok = ok and np.array_equal(got.view(np.uint64), exp.view(np.uint64))
    bad += not ok
    print(f"seed {seed}: n={n} nk={nk} used={used} chunk={chunk} G={len(uniq)} {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"done: {count} cases, {bad} mismatches")
