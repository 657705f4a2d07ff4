"""GPU: runs the C++ facade test program (tests/cpp/test_facade.cpp): the reference's Catch2 cases for the hot path replayed
through pandasarrow_amd/cpp/pdx.hpp -> C ABI -> HIP kernels."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_cpp_facade_reference_cases():
    import __graft_entry__ as ge

    exe = os.path.join(ROOT, "tests", "cpp", "test_facade")
    if not os.path.exists(exe):
        ge.build_hip()
        exe = ge.build_cpp_facade_test()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 failed" in r.stdout
