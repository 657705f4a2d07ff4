#!/usr/bin/env python3
"""Secondary measurements for the other rows of SURVEY.md section 8 (not the contract bench): achieved GB/s against each op's
ALGORITHMIC bytes on BASELINE.json's configs C1 / C2 / C5 and a 1e9-row element-wise / whole-array pass.
Usage: python tools/bench_ops.py [--scale 1.0]   (prints one JSON line per op)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pandasarrow_amd import _lib as L  # noqa: E402
from pandasarrow_amd import api  # noqa: E402
from pandasarrow_amd import column as K  # noqa: E402

PEAK = 8000.0


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def report(name, rows, algo_bytes, dt, extra=None):
    gbs = algo_bytes / dt / 1e9
    line = {"op": name, "rows": rows, "ms": dt * 1e3, "Grows/s": rows / dt / 1e9, "algo_GB/s": gbs, "frac_of_8TB/s": gbs / PEAK}
    if extra:
        line.update(extra)
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    a = ap.parse_args()
    L.check(L.load().pdx_init(0))
    sc = a.scale
    # C1: Series<double> add + sum, 1e6 rows (plumbing) and the same at 1e9 rows
    for n in (int(1e6), int(1e9 * sc)):
        x, y = K.synth_vals(0, n, 1), K.synth_vals(0, n, 2)
        report(f"add_f64[{n:.0e}]", n, 24.0 * n, timeit(lambda: K.binary(L.ADD, x, y)))
        report(f"sum_f64[{n:.0e}]", n, 8.0 * n, timeit(lambda: K.aggregate(L.AGG_SUM, x)))
        report(f"minmax_f64[{n:.0e}]", n, 8.0 * n, timeit(lambda: K.aggregate(L.AGG_MIN, x)))
        report(f"greater_f64[{n:.0e}]", n, (16.0 + 0.125) * n, timeit(lambda: K.compare(L.GT, x, y)))
        # 5 % nulls (SURVEY 8d secondary run): validity bitmap = (mix(i+3) % 20 != 0)
        vmask = K.compare(L.NE, K.synth_keys(3, n, 20), 0)
        xn = K.Column(L.FLOAT64, n, x.values, vmask.values, 0, -1)
        report(f"sum_f64_5%nulls[{n:.0e}]", n, (8.0 + 0.125) * n, timeit(lambda: K.aggregate(L.AGG_SUM, xn)))
        yn = K.Column(L.FLOAT64, n, y.values, vmask.values, 0, -1)  # (a second value buffer: x + x would read 8 B/row less)
        report(f"add_f64_5%nulls[{n:.0e}]", n, (24.0 + 0.375) * n, timeit(lambda: K.binary(L.ADD, xn, yn)))
        del x, y, xn, yn, vmask
    # C2: boolean-mask filter + take, 1e8 rows x 8 fp64 columns (+ uint64 index)
    n = int(1e8 * sc)
    cols = {f"c{j}": K.synth_vals(0, n, 20 + j) for j in range(8)}
    idx = K.synth_keys(0, n, 1 << 62)  # stands in for an explicit index column
    df = api.DataFrame(cols, index=idx)
    mask = df["c0"] > 0.5
    sel = K.filter_count(mask.col)
    s = sel / n
    report("filter_8cols+index[1e8]", n, (0.125 + 8 * 9 * (1 + s)) * n, timeit(lambda: df.where(mask)), {"selectivity": s})
    # the same with 5 % nulls in every column (validity bitmaps travel too: + 9 x (1 + s) / 8 B per row)
    vm = K.compare(L.NE, K.synth_keys(3, n, 20), 0)
    colsn = {f"c{j}": K.Column(L.FLOAT64, n, cols[f"c{j}"].values, vm.values, 0, -1) for j in range(8)}
    dfn = api.DataFrame(colsn, index=idx)
    report("filter_8cols+index_5%nulls[1e8]", n, (0.125 + (8 + 0.125) * 9 * (1 + s)) * n, timeit(lambda: dfn.where(mask)), {"selectivity": s})
    del dfn, colsn, vm
    m = n // 2
    take_idx = api.Series(K.synth_keys(7, m, n))
    report("take_8cols+index[5e7 of 1e8]", m, (8 + 16 * 9) * m, timeit(lambda: df.take(take_idx)))
    del df, cols, idx, mask, take_idx
    # C5: resample('1min').mean() on a 1e9-row timestamp + fp64 Series (100 ms spacing -> 600 rows per bin)
    n = int(1e9 * sc)
    ts = K.synth_ts(0, n, 946_684_800 * 10**9, 100_000_000)
    ser = api.Series(K.synth_vals(0, n), index=ts, name="v")
    report("resample_1min_mean[1e9]", n, 16.0 * n, timeit(lambda: ser.resample("1min").mean()))
    # argsort of 1e8 float64 values (algorithmic bytes: 8 read + 8 written per row)
    m = int(1e8 * sc)
    sv = K.synth_vals(11, m)
    report("argsort_f64[1e8]", m, 16.0 * m, timeit(lambda: K.argsort(sv)))
    del sv
    # concat of 8 shards (the all-gatherv merge)
    parts = [K.synth_vals(i, n // 8, 0) for i in range(8)]
    report("concat_8parts[1e9]", n // 8 * 8, 16.0 * (n // 8 * 8), timeit(lambda: K.concat(parts)))
    del parts, ser, ts
    # "next" rows: more group-by aggregations on the grouped layout, 1e9 rows / 1e6 keys (16 B/row algorithmic)
    keys, vals = K.synth_keys(0, n, 1_000_000), K.synth_vals(0, n)
    gb = K.GroupByHandle.create(keys)
    for nm, kinds in (("variance", [L.AGG_VARIANCE]), ("stddev+mean", [L.AGG_STDDEV, L.AGG_MEAN]), ("product", [L.AGG_PRODUCT]),
                      ("first+last", [L.AGG_FIRST, L.AGG_LAST])):
        report(f"groupby_agg_{nm}[1e9/1e6]", n, 16.0 * n, timeit(lambda: gb.agg(vals, kinds), reps=3, warm=1))
    del gb, keys, vals
    # index alignment: union of two 1e8-label indexes sharing half of their labels, and the reindex of one onto the union
    m = int(1e8 * sc)
    ia = K.binary(L.MUL, K.synth_keys(0, m, 1 << 40), 1)          # ~unique int64 labels
    ib = K.concat([ia.slice(m // 2, m - m // 2), K.binary(L.ADD, K.synth_keys(11, m // 2, 1 << 40), 1 << 41)])
    report("index_union[1e8+1e8]", 2 * m, 8.0 * 2 * m + 8.0 * 1.5 * m, timeit(lambda: K.index_union(ia, ib), reps=3, warm=1))
    uni = K.index_union(ia, ib)
    report("reindex_indices[1e8 -> 1.5e8]", uni.length, 8.0 * m + 16.0 * uni.length, timeit(lambda: K.reindex_indices(ia, uni), reps=3, warm=1))


if __name__ == "__main__":
    main()
