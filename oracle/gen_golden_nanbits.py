#!/usr/bin/env python3
"""Golden vectors for the NaN SIGN / PAYLOAD of fp64 sums (round 4): Arrow C++ 25.0.0 through pyarrow on this x86-64 host, frozen in
tests/golden/nan_bits_golden.npz (OUTPUTS only; the seeded inputs come from tests/_nanbits_inputs.py).  Arrow's pairwise `sum` propagates
NaNs the way its compiled SSE adds do: inside a 16-value leaf the accumulator's (earlier) NaN wins, in every merge of the tree the later
operand's, inf + -inf makes the negative default NaN; `mean` divides that sum by the count (the NaN's bits survive).  Cases: whole-column
sum / mean with and without nulls over many lengths, and per-group sum / mean (the group's rows gathered in row order, one pc.sum per
group -- the reference's call sequence, src/pd_core_macros.h:80-147).
Run: python oracle/gen_golden_nanbits.py"""
import json
import os
import sys

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _nanbits_inputs as inp  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "nan_bits_golden.npz")


def arrow_sum_mean(v, valid):
    a = pa.array(v, mask=None if valid is None else ~valid)
    s, m = pc.sum(a).as_py(), pc.mean(a).as_py()
    return (np.float64(np.nan) if s is None else np.float64(s)), (np.float64(np.nan) if m is None else np.float64(m)), s is not None


def first_occurrence_groups(keys):
    uniq, first = np.unique(keys, return_index=True)
    uniq = uniq[np.argsort(first, kind="stable")]
    srt = np.argsort(uniq, kind="stable")          # sorted position -> first-occurrence id
    ids = srt[np.searchsorted(uniq[srt], keys)]
    return uniq, ids


def main():
    out, manifest = {}, {"whole": [], "group": []}
    for name, v, valid in inp.whole_cases():
        s, m, ok = arrow_sum_mean(v, valid)
        out[name] = np.array([s, m, 1.0 if ok else 0.0])
        manifest["whole"].append(name)
    for name, keys, v, valid in inp.group_cases():
        uniq, ids = first_occurrence_groups(keys)
        rows = np.argsort(ids, kind="stable")
        bounds = np.r_[0, np.cumsum(np.bincount(ids, minlength=len(uniq)))]
        res = np.empty((3, len(uniq)))
        for g in range(len(uniq)):
            r = rows[bounds[g]:bounds[g + 1]]
            s, m, ok = arrow_sum_mean(v[r], None if valid is None else valid[r])
            res[:, g] = (s, m, 1.0 if ok else 0.0)
        out[name + "/uniq"] = uniq
        out[name] = res
        manifest["group"].append(name)
    out["manifest"] = np.array(json.dumps({"arrow": pa.__version__, "cases": manifest}))
    np.savez_compressed(OUT, **out)
    print(f"{len(manifest['whole'])} whole-column cases, {len(manifest['group'])} group cases -> {OUT} ({os.path.getsize(OUT) / 1e6:.2f} MB)")


if __name__ == "__main__":
    main()
