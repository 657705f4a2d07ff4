"""Pins the round-2 additions of the CPU oracle against Arrow 25.0.0 golden vectors (oracle/gen_golden_r2.py ->
tests/golden/arrow_golden_r2.npz) and the reference's own known answers.  CPU only."""
import numpy as np
import pytest

import oracle as orc
from conftest import assert_f64_bits, golden2

G2 = golden2()
OPS = {"add": orc.ADD, "sub": orc.SUB, "mul": orc.MUL, "div": orc.DIV}
CMPS = {"eq": orc.EQ, "ne": orc.NE, "lt": orc.LT, "le": orc.LE, "gt": orc.GT, "ge": orc.GE}


# ------------------------------------------------------------------ Scalar op Series (src/scalar.cpp:24-56)
@pytest.mark.parametrize("name", G2.cases("scalar_lhs"))
@pytest.mark.parametrize("offset", [0, 5])
def test_scalar_lhs(name, offset):
    c = G2.case(name)
    s = c["s"].item()
    sv = None if bool(c["s_valid"]) else np.array([False])
    vb = None if c["vb"].all() else c["vb"]
    for k, op in OPS.items():
        vals, valid = orc.binary(op, s, c["b"], sv, vb, offset)
        ev = c[f"{k}_valid"]
        if valid is not None:
            assert np.array_equal(valid, ev), f"{name} {k} validity"
        else:
            assert ev.all()
        if vals.dtype == np.float64:
            assert_f64_bits(vals, c[k], valid=ev, what=f"{name} {k}")
        else:
            assert np.array_equal(vals[ev], c[k][ev]), f"{name} {k}"
    if "eq" not in c:
        return
    for k, op in CMPS.items():
        vals, valid = orc.compare(op, s, c["b"], sv, vb, offset)
        ev = c[f"{k}_valid"]
        assert np.array_equal(vals[ev], c[k][ev]), f"{name} {k}"
        if valid is not None:
            assert np.array_equal(valid, ev)


def test_scalar_lhs_divide_by_zero():
    with pytest.raises(orc.OracleError) as e:
        orc.binary(orc.DIV, 7, np.array([2, 0]))
    assert str(e.value) == G2.manifest["scalar_lhs_div_by_zero_message"]
    vals, valid = orc.binary(orc.DIV, 7, np.array([2, 0]), None, np.array([True, False]))
    assert vals[0] == 3 and list(valid) == [True, False]
