"""Cycles per phase of the dense fused last-digit reduce.  Needs a -DPDX_FLR_TIMING build:
   python tools/build_variant.py tm -DPDX_FLR_TIMING=1
   gpurun -- 'PDX_LIB_PATH=$PWD/tools/_ab/libpdx_tm.so python tools/flr_phase_timing.py'"""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
from pandasarrow_amd import _lib as L, column as K
lib = L.load(); L.check(lib.pdx_init(0))
n, nk = int(1e9), int(1e6)
keys = K.synth_keys(0, n, nk); vals = K.synth_vals(0, n)
kinds = [L.AGG_SUM, L.AGG_MEAN, L.AGG_COUNT]
raw = ctypes.CDLL(L.LIB_PATH)
raw.pdx_debug_flr_cycles.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 24)()
for it in range(3):
    gb = K.GroupByHandle.create(keys); outs = gb.agg(vals, kinds); torch.cuda.synchronize()
    raw.pdx_debug_flr_cycles(buf, 1)
names = ["push(prev)", "rank", "bar1", "prefix", "bar2", "stage", "bar3", "leaf", "bar4", "issue"]
tiles = 1e9 / 2560
for w in range(2):
    tot = sum(buf[w * 12 + i] for i in range(10))
    print("wave", w, "total cycles/tile", round(tot / tiles))
    for i, nm in enumerate(names):
        print(f"   {nm:12s} {buf[w*12+i]/tiles:9.0f}  {100*buf[w*12+i]/max(tot,1):5.1f} %")
