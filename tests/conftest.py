"""pytest configuration: registers the ``gpu`` marker and shared fixtures.

``-m "not gpu"`` : oracle-vs-golden, host logic, C-ABI symbol checks (no GPU needed).
``-m gpu``       : parity tests proper -- HIP path called through the C ABI vs the oracle.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


class Golden:
    """Accessor over tests/golden/arrow_golden*.npz (cases are 'name/field' keys)."""

    def __init__(self, fname="arrow_golden.npz"):
        self.z = np.load(os.path.join(GOLDEN_DIR, fname))
        self.manifest = json.loads(str(self.z["manifest"]))

    def cases(self, family):
        return self.manifest["cases"][family]

    def case(self, name):
        pre = name + "/"
        return {k[len(pre):]: self.z[k] for k in self.z.files if k.startswith(pre)}


_golden = None


def golden():
    global _golden
    if _golden is None:
        _golden = Golden()
    return _golden


_golden2 = None


def golden2():
    """Round-2 additions (oracle/gen_golden_r2.py): scalar-lhs ops, temporal rounding, the remaining group-by aggregations."""
    global _golden2
    if _golden2 is None:
        _golden2 = Golden("arrow_golden_r2.npz")
    return _golden2


_golden3 = None


def golden3():
    """Round-3 additions (oracle/gen_golden_r3.py): DataFrame comparisons / and / or, reindex with a fill value."""
    global _golden3
    if _golden3 is None:
        _golden3 = Golden("arrow_golden_r3.npz")
    return _golden3


@pytest.fixture(scope="session")
def gold():
    return golden()


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(GOLDEN_DIR, "kat_reference.json")) as f:
        return json.load(f)


def bits_equal(a, b):
    """bit-exact comparison of float64 arrays (NaN == NaN, -0.0 != 0.0)."""
    a = np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
    b = np.ascontiguousarray(b, dtype=np.float64).view(np.uint64)
    return a.shape == b.shape and bool(np.all(a == b))


def assert_f64_bits(a, b, valid=None, what="", nan_bits=True):
    """bit-exact float64 comparison.  nan_bits=True (the default since round 4): sign and payload of a NaN must match too -- the
    element-wise kernels AND the sum trees spell out the x86 operand rules the reference's Arrow kernels were compiled to (leaf: the
    earlier NaN, merges: the later operand's, inf + -inf: the negative default NaN; tests/golden/nan_bits_golden.npz).  nan_bits=False:
    any NaN equals any NaN (variance / stddev: Arrow's own two-pass formula, where only NaN-ness is pinned)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} != {b.shape}"
    ua, ub = a.view(np.uint64), b.view(np.uint64)
    neq = ua != ub
    if not nan_bits:
        neq &= ~(np.isnan(a) & np.isnan(b))
    if valid is not None:
        neq &= np.asarray(valid, bool)
    if neq.any():
        i = int(np.flatnonzero(neq)[0])
        raise AssertionError(f"{what}: {int(neq.sum())} mismatches, first at {i}: {a[i]!r} ({ua[i]:#x}) != {b[i]!r} ({ub[i]:#x})")
