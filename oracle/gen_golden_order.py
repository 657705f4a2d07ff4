#!/usr/bin/env python3
"""Generate tests/golden/group_order_arrow25.npz -- the ONE place where this backend's result ROW ORDER deliberately differs from
the reference's on large inputs.

TEST INFRASTRUCTURE.  The reference takes group ids from arrow::compute::Grouper::Consume / GetUniques (src/dataframe.cpp:1580-1591).
Arrow 25's Grouper hands out ids per 1024-row mini-batch of its swiss table; a new key that collides inside a mini-batch is inserted in
a later round and gets its id after the mini-batch's other new keys, so on inputs with many new keys per mini-batch the order is only
approximately first-occurrence.  This backend (like pandas sort=False and the oracle) defines FIRST-OCCURRENCE order; per key every
aggregate is bit-identical.  The fixture freezes Arrow 25.0.0's actual order (pyarrow Table.group_by(use_threads=False), which drives
the same Grouper) on two seeded inputs where the orders differ, so tests/test_oracle_golden_r3.py / test_gpu_round3.py can state the
deviation exactly: same key set, same per-key results, a counted number of positions that differ.

Run:  python oracle/gen_golden_order.py
"""
import json
import os
import sys

import numpy as np
import pyarrow as pa

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "group_order_arrow25.npz")

store, manifest = {}, {"arrow_version": pa.__version__, "cases": {}}
for name, (n, nk, seed) in {"rows1e5_keys5e4": (100_000, 50_000, 1), "rows2e6_keys1e4": (2_000_000, 10_000, 1)}.items():
    rng = np.random.default_rng(seed)
    keys = rng.integers(0, nk, n).astype(np.int64)
    vals = rng.standard_normal(n)
    res = pa.table({"k": keys, "v": vals}).group_by("k", use_threads=False).aggregate([("v", "count")])
    arrow_order = res["k"].to_numpy()
    _, first = np.unique(keys, return_index=True)
    first_occ = keys[np.sort(first)]
    differ = int((arrow_order != first_occ).sum())
    store[f"{name}/arrow_order"] = arrow_order
    manifest["cases"][name] = {"rows": n, "keys": nk, "seed": seed, "groups": int(len(first_occ)), "positions_that_differ": differ}
    print(name, len(first_occ), "groups,", differ, "positions differ from first-occurrence order")
store["manifest"] = np.array(json.dumps(manifest))
np.savez_compressed(OUT, **store)
print("wrote", OUT, os.path.getsize(OUT) // 1000, "kB")
