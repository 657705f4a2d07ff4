#!/usr/bin/env python3
"""Secondary run of SURVEY.md 8d: group-by sum/mean/count with 5 % null values (mix(i+3) % 20 == 0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pandasarrow_amd import _lib as L, column as K
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
L.check(L.load().pdx_init(0))
keys, vals = K.synth_keys(0, n, 1_000_000), K.synth_vals(0, n)
# validity bitmap: ~5 % nulls, built on device from the synthetic stream (compare kernel output reused as a bitmap)
r = K.synth_keys(3, n, 20)
valid = K.compare(L.NE, r, 0)
vals_n = K.Column(L.FLOAT64, n, vals.values, valid.values, 0, -1)
for name, col in (("no nulls", vals), ("5% nulls", vals_n)):
    def step():
        gb = K.GroupByHandle.create(keys)
        return gb.agg(col, [L.AGG_SUM, L.AGG_MEAN, L.AGG_COUNT])
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {dt*1e3:.2f} ms/step  {n/dt/1e9:.2f} Grows/s", flush=True)
    if os.environ.get("PDX_SKEW_KERNELS"):  # per-kernel device time of one more step (HIP events inside the library)
        import ctypes as C
        lib = L.load()
        lib.pdx_profile_reset(); lib.pdx_profile_enable(1)
        step(); torch.cuda.synchronize()
        lib.pdx_profile_enable(0)
        buf = C.create_string_buffer(1 << 16)
        L.check(lib.pdx_profile_report(buf, len(buf)))
        print("   " + "  ".join(f"{t}={float(ms):.2f}" for t, c, ms in (l.split() for l in buf.value.decode().splitlines()) if float(ms) >= 0.2), flush=True)
