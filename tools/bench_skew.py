#!/usr/bin/env python3
"""Skewed / null-heavy key shapes of the group-by (sum, mean, count), 1e9 rows by default: a hot key holding 30 % of the rows
(dense and general keys), 5 % null values on top, and general keys where half of the rows have a null key."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pandasarrow_amd import _lib as L, column as K
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
L.check(L.load().pdx_init(0))
keys, vals = K.synth_keys(0, n, 1_000_000), K.synth_vals(0, n)
r = K.synth_keys(3, n, 20)
vals_n = K.Column(L.FLOAT64, n, vals.values, K.compare(L.NE, r, 0).values, 0, -1)
hot = torch.where(K.synth_keys(5, n, 10).values < 3, 4242, keys.values)
hot_keys = K.Column(L.INT64, n, hot, None, 0, 0)
half_null = K.Column(L.INT64, n, keys.values, K.compare(L.LT, K.synth_keys(7, n, 2), 1).values, 0, -1)
del r


def run(name, kcol, vcol):
    def step():
        gb = K.GroupByHandle.create(kcol)
        return gb.agg(vcol, [L.AGG_SUM, L.AGG_MEAN, L.AGG_COUNT])
    step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {dt*1e3:.1f} ms/step  {n/dt/1e9:.2f} Grows/s", flush=True)
    if os.environ.get("PDX_SKEW_KERNELS"):  # per-kernel device time of one more step (HIP events inside the library)
        import ctypes as C
        lib = L.load()
        lib.pdx_profile_reset(); lib.pdx_profile_enable(1)
        step(); torch.cuda.synchronize()
        lib.pdx_profile_enable(0)
        buf = C.create_string_buffer(1 << 16)
        L.check(lib.pdx_profile_report(buf, len(buf)))
        print("   " + "  ".join(f"{t}={float(ms):.1f}" for t, c, ms in (l.split() for l in buf.value.decode().splitlines()) if float(ms) >= 0.5), flush=True)


for dense in ("1", "0"):
    os.environ["PDX_GROUPBY_DENSE"] = dense
    tag = "dense" if dense == "1" else "hash"
    run(f"{tag}: uniform", keys, vals)
    run(f"{tag}: hot key 30%", hot_keys, vals)
    run(f"{tag}: hot key 30% + 5% null values", hot_keys, vals_n)
    run(f"{tag}: 50% null keys", half_null, vals)
