import sys, time, torch
sys.path.insert(0, '.')
from pandasarrow_amd import _lib as L, column as K
L.check(L.load().pdx_init(0))
for W in (2, 4, 8):
    n = W * 1_000_000
    keys = K.synth_keys(3, n, 1_000_000)
    for _ in range(3):
        gb = K.GroupByHandle.create(keys); del gb
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        gb = K.GroupByHandle.create(keys); fr = gb.first_rows(); del gb
    torch.cuda.synchronize()
    print(f"W={W}: create + first_rows over {n} dictionary entries: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
