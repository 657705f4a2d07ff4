#!/usr/bin/env python3
"""Golden vectors for the one-column element-wise functions (negate, abs, sign, sqrt, exp, bit_wise_not, power) from Arrow C++
25.0.0 through pyarrow.compute -- the kernels DataFrame::unary / UNARY_FUNCTION / pow forward to (reference src/dataframe.cpp:251-275,
919-935).  TEST INFRASTRUCTURE: run here (pyarrow is not assumed on the GPU box); writes tests/golden/unary_golden.npz, which
tests/test_oracle_golden_r2.py (oracle vs Arrow) and tests/test_gpu_round2.py (HIP vs Arrow) read.
All values are stored as 8-byte patterns (uint64) so NaN payloads and zero signs survive."""
import json
import os
import sys

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "unary_golden.npz")
store = {}
manifest = {"arrow_version": pa.__version__, "cases": [], "errors": []}
OPS = ["negate", "abs", "sign", "sqrt", "exp", "bit_wise_not"]
EXPONENTS = [2.0, 0.5, -1.0, 3.0, 0.0, 1.5, -0.5, 10.0]


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64) if a.dtype.itemsize == 8 else a.astype(np.int64).view(np.uint64)


def result_bits(r):
    valid = np.array([x is not None for x in r.to_pylist()], bool) if r.null_count else np.ones(len(r), bool)
    if pa.types.is_floating(r.type):
        v = np.asarray(r.fill_null(0.0).to_numpy(zero_copy_only=False), np.float64)
    elif pa.types.is_unsigned_integer(r.type):
        v = np.asarray(r.fill_null(0).to_numpy(zero_copy_only=False), np.uint64)
    else:  # int8 (sign) / int64 -> int64
        v = np.asarray(r.fill_null(0).to_numpy(zero_copy_only=False)).astype(np.int64)
    return bits(v), valid


def inputs():
    rng = np.random.default_rng(2026)
    nan = lambda payload, neg=False: np.array([0x7FF8000000000000 | payload | (0x8000000000000000 if neg else 0)], np.uint64).view(np.float64)[0]
    f_special = np.array([0.0, -0.0, 1.0, -1.0, 2.5, -3.5, 4.0, 1e-320, -1e-320, 1e308, -1e308, np.inf, -np.inf, nan(0), nan(0x123, True), nan(0x7FFFF),
                          0.5, 709.0, 710.0, -745.0, -746.0, 1e-300, 3.0, 9.0, 1e16 + 2.0])
    f_rand = np.concatenate([rng.standard_normal(400) * 10.0 ** rng.integers(-8, 9, 400), rng.random(200) * 700.0, -rng.random(100) * 700.0])
    i_vals = np.concatenate([np.array([0, 1, -1, 5, -7, 2**53, -(2**53), 4, 9, 1 << 40], np.int64), rng.integers(-(2**53), 2**53, 300)])
    i_big = np.array([0, 3, -(2**63), 2**63 - 1, 2**53 + 1, -5], np.int64)
    u_vals = np.concatenate([np.array([0, 1, 5, 2**53, 16, 1 << 50], np.uint64), rng.integers(0, 2**53, 200).astype(np.uint64)])
    u_big = np.array([0, 7, 2**64 - 1, 2**63], np.uint64)
    return {"f_special": f_special, "f_rand": f_rand, "i_vals": i_vals, "i_big": i_big, "u_vals": u_vals, "u_big": u_big}


def main():
    rng = np.random.default_rng(7)
    for name, v in inputs().items():
        typ = {"float64": pa.float64(), "int64": pa.int64(), "uint64": pa.uint64()}[v.dtype.name]
        for with_nulls in (False, True):
            valid = (rng.random(len(v)) > 0.25) if with_nulls else None
            a = pa.array(v, type=typ, mask=None if valid is None else ~valid)
            case = f"{name}_{'nulls' if with_nulls else 'dense'}"
            store[f"{case}/in"] = bits(v)
            store[f"{case}/dtype"] = np.array(v.dtype.name)
            store[f"{case}/valid"] = np.ones(len(v), bool) if valid is None else valid
            done = []
            for op in OPS:
                try:
                    r = pc.call_function(op, [a])
                except (pa.ArrowInvalid, pa.ArrowNotImplementedError) as e:
                    manifest["errors"].append({"case": case, "op": op, "message": str(e).splitlines()[0]})
                    continue
                store[f"{case}/{op}"], rv = result_bits(r)
                assert np.array_equal(rv, store[f"{case}/valid"])
                done.append(op)
            for k, e in enumerate(EXPONENTS):
                try:
                    r = pc.power(a, pa.scalar(e, pa.float64()))
                except (pa.ArrowInvalid, pa.ArrowNotImplementedError) as ex:
                    if k == 0:
                        manifest["errors"].append({"case": case, "op": "power", "message": str(ex).splitlines()[0]})
                    continue
                store[f"{case}/power_{k}"], _ = result_bits(r)
                done.append(f"power_{k}")
            manifest["cases"].append({"case": case, "ops": done})
    # ---- bit_wise_or / and / xor, shift_left / shift_right on int64 (src/series.cpp:237-245, src/dataframe.cpp:553-561)
    n = 600
    a = np.concatenate([np.array([1, -8, 5, 2**62, -1, 7, 0, -(2**63), 2**63 - 1], np.int64), rng.integers(-(2**63), 2**63 - 1, n)])
    b = np.concatenate([np.array([1, 2, 64, 1, 63, -1, 62, 1, 62], np.int64), rng.integers(-3, 70, n)])
    va, vb = rng.random(len(a)) > 0.2, rng.random(len(a)) > 0.2
    store["bw/a"], store["bw/b"], store["bw/va"], store["bw/vb"] = a, b, va, vb
    pa_a, pa_b = pa.array(a, mask=~va), pa.array(b, mask=~vb)
    manifest["bitwise_scalars"] = [3, -1, 63, 0, 40]
    for k, fn in enumerate(["bit_wise_or", "bit_wise_and", "bit_wise_xor", "shift_left", "shift_right"]):
        r = pc.call_function(fn, [pa_a, pa_b])
        store[f"bw/{fn}"], store[f"bw/{fn}_valid"] = result_bits(r)
        for j, sc in enumerate(manifest["bitwise_scalars"]):
            store[f"bw/{fn}_rhs{j}"], _ = result_bits(pc.call_function(fn, [pa_a, pa.scalar(sc, pa.int64())]))
            store[f"bw/{fn}_lhs{j}"], _ = result_bits(pc.call_function(fn, [pa.scalar(sc, pa.int64()), pa_b]))
    # ---- if_else(cond, a, b) (Series::if_else / where(cond, other), src/series.cpp:1203-1209, 1247-1253)
    m = 500
    cond, cv = rng.random(m) > 0.5, rng.random(m) > 0.15
    ai, bi, af, bf = rng.integers(-99, 99, m), rng.integers(1000, 2000, m), rng.standard_normal(m), rng.standard_normal(m) + 100.0
    va_, vb_ = rng.random(m) > 0.2, rng.random(m) > 0.2
    store["ie/cond"], store["ie/cv"], store["ie/ai"], store["ie/bi"], store["ie/af"], store["ie/bf"], store["ie/va"], store["ie/vb"] = cond, cv, ai, bi, af, bf, va_, vb_
    pc_cond = pa.array(cond, mask=~cv)
    combos = {"ii": (pa.array(ai, mask=~va_), pa.array(bi, mask=~vb_)), "ff": (pa.array(af, mask=~va_), pa.array(bf, mask=~vb_)),
              "if": (pa.array(ai, mask=~va_), pa.array(bf, mask=~vb_)), "fi": (pa.array(af, mask=~va_), pa.array(bi, mask=~vb_)),
              "i_s7": (pa.array(ai, mask=~va_), pa.scalar(7, pa.int64())), "f_snull": (pa.array(af, mask=~va_), pa.scalar(None, pa.float64())),
              "s2.5_i": (pa.scalar(2.5, pa.float64()), pa.array(bi, mask=~vb_)), "i_s1.5": (pa.array(ai, mask=~va_), pa.scalar(1.5, pa.float64()))}
    manifest["if_else"] = list(combos)
    for name, (x, y) in combos.items():
        store[f"ie/{name}"], store[f"ie/{name}_valid"] = result_bits(pc.if_else(pc_cond, x, y))
    manifest["exponents"] = EXPONENTS
    store["manifest"] = np.array(json.dumps(manifest))
    np.savez_compressed(OUT, **store)
    print(f"wrote {OUT}: {len(store)} arrays, {os.path.getsize(OUT)} bytes; arrow {pa.__version__}; {len(manifest['errors'])} error cases")


if __name__ == "__main__":
    sys.exit(main())
