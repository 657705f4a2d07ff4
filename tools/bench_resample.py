#!/usr/bin/env python3
"""resample('1min').mean() on 1e9 sorted timestamps + fp64 values (BASELINE configs[4] on one GPU): step time and the
per-kernel device time (HIP events inside the library)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pandasarrow_amd import _lib as L, column as K, api
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
lib = L.load()
L.check(lib.pdx_init(0))
ts = K.synth_ts(0, n, 946_684_800 * 10**9, 100_000_000)
ser = api.Series(K.synth_vals(0, n), index=ts, name="v")
step = lambda: ser.resample("1min").mean()
step(); step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"resample 1min mean: {dt*1e3:.2f} ms/step  {n/dt/1e9:.1f} Grows/s  {16*n/dt/1e12:.2f} TB/s algorithmic", flush=True)
lib.pdx_profile_reset(); lib.pdx_profile_enable(1)
step(); torch.cuda.synchronize()
lib.pdx_profile_enable(0)
buf = C.create_string_buffer(1 << 16)
L.check(lib.pdx_profile_report(buf, len(buf)))
print("   " + "  ".join(f"{t}={float(ms):.2f}" for t, c, ms in (l.split() for l in buf.value.decode().splitlines())), flush=True)
