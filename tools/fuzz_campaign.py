#!/usr/bin/env python3
"""One-off larger differential fuzz of the group-by path (same generator as tests/test_gpu_fuzz.py, other seeds, bigger inputs and a
bias towards the narrowing-sort / fused-last-digit configurations).  Usage: python tools/fuzz_campaign.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as orc
from pandasarrow_amd import _lib as L, column as K
import test_gpu_fuzz as F

L.check(L.load().pdx_init(0))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
MODES = {"default": {}, "fused_dense": {"PDX_FUSED_LAST_DIGIT_MIN_ROWS": "0", "PDX_FUSED_LAST_DIGIT_MIN_LOW_BITS": "4", "PDX_FUSED_LAST_DIGIT_MIN_RUN": "0"},
         "hash_tail": {"PDX_GROUPBY_DENSE": "0", "PDX_HASH_HEAD_ROWS": "4096"}, "null_pw": {"PDX_FLR_NULL_PW": "1", "PDX_FUSED_LAST_DIGIT_MIN_ROWS": "0",
         "PDX_FUSED_LAST_DIGIT_MIN_LOW_BITS": "4", "PDX_FUSED_LAST_DIGIT_MIN_RUN": "0"},
         # hot keys: runs over 20 000 rows are "long" -> the side form of the fused layout (or the classic route when there are too many)
         "side": {"PDX_FLR_MAX_RUN": "20000", "PDX_FUSED_LAST_DIGIT_MIN_ROWS": "0", "PDX_FUSED_LAST_DIGIT_MIN_LOW_BITS": "4", "PDX_FUSED_LAST_DIGIT_MIN_RUN": "0"},
         # round 4: requests of order-free kinds only (count / min / max / int64 sum) -> the accumulate path (gb_acc.hpp), dense and hash slots
         "acc": {"PDX_ACC_MIN_ROWS": "1"}, "acc_hash": {"PDX_ACC_MIN_ROWS": "1", "PDX_GROUPBY_DENSE": "0"}}
ALL_ENV = sorted({k for m in MODES.values() for k in m})
bad = 0
plans = {}
for i in range(count):
    seed = first + i
    rng = np.random.default_rng(seed)
    keys, kvalid, vals, vvalid, kinds = F._make_case(seed)
    if rng.random() < 0.5:  # a large high-cardinality case: two sort digits below the fused one
        n = int(rng.integers(3_000_000, 6_000_000))
        card = int(rng.choice([200_000, 1_000_000, 3_000_000]))
        keys = rng.integers(0, card, n).astype(np.int64) + int(rng.integers(-5, 5)) * 1000
        if rng.random() < 0.3:
            keys[rng.random(n) < 0.2] = 77
        elif rng.random() < 0.4:  # a few keys with a few per cent each
            u = rng.random(n)
            lo = 0.0
            for _ in range(int(rng.integers(1, 4))):
                share = float(rng.uniform(0.01, 0.05))
                keys[(u >= lo) & (u < lo + share)] = int(rng.integers(0, card))
                lo += share
        kvalid = (rng.random(n) > 0.03) if rng.random() < 0.3 else None
        vals = rng.standard_normal(n) if rng.random() < 0.6 else rng.integers(-50, 50, n).astype(np.int64)
        vvalid = (rng.random(n) > rng.choice([0.02, 0.5])) if rng.random() < 0.5 else None
        if rng.random() < 0.5:
            kinds = [0, 1, 4]
    mode = list(MODES)[seed % len(MODES)]
    if mode.startswith("acc"):
        pool = [2, 3, 4] + ([0] if vals.dtype == np.int64 else [])
        kinds = [k for k in pool if rng.random() < 0.6] or [2, 3]
        if rng.random() < 0.5 and vals.dtype == np.float64:  # tied zeros of both signs: the zero-ties pass
            z = rng.random(len(vals)) < 0.3
            vals = vals.copy()
            vals[z] = rng.choice(np.array([0.0, -0.0]), int(z.sum()))
    for k in ALL_ENV:
        os.environ.pop(k, None)
    os.environ.update(MODES[mode])
    gb = K.GroupByHandle.create(K.Column.from_numpy(keys, kvalid, offset=seed % 3))
    ids, uniq, isnull, frst = orc.group_ids(keys, kvalid)
    G = len(uniq)
    ok_all = gb.num_groups == G and np.array_equal(gb.group_ids().cpu().numpy().astype(np.uint32), ids) and np.array_equal(gb.first_rows().cpu().numpy(), frst)
    if seed % 3 == 0:  # Grouper::MakeGroupings: a stable argsort of the ids
        rows, off = gb.groupings()
        ok_all = ok_all and np.array_equal(rows.cpu().numpy(), np.argsort(ids, kind="stable")) and np.array_equal(
            off.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(ids, minlength=G))]))
    outs = gb.agg(K.Column.from_numpy(vals, vvalid, offset=seed % 5), kinds)
    pl = gb.last_plan()
    pkey = (mode, pl.get("layout", "").split(":")[0], pl.get("reducer"), "side" if "side" in pl else pl.get("skew", ""), "ties" if "zero_ties" in pl else "")
    plans[pkey] = plans.get(pkey, 0) + 1
    for kind, out in zip(kinds, outs):
        got, ok = out.to_numpy()
        exp, eok = orc.groupby_agg(kind, ids, G, vals, vvalid, nthreads=8)
        ok_all = ok_all and ((ok is None and eok.all()) or np.array_equal(ok, eok)) and F._bits_equal(got, exp, eok)
    if not ok_all:
        bad += 1
        print("MISMATCH seed", seed, "mode", mode, "n", len(keys), "G", G, "kinds", kinds, flush=True)
    if i % 25 == 24:
        print(f"{i + 1} cases, {bad} mismatches", flush=True)
print("paths:", sorted(plans.items()))
print("done:", count, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
