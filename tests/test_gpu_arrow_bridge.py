"""GPU: the reference-side binding compiled against REAL Arrow types (tests/cpp/arrow_bridge_test.cpp): DeviceArray(const
arrow::ArrayData&), a CallFunction(name, {Datum...}, options) dispatcher for the kernel names the reference uses, and a Grouper-shaped
wrapper over pdx_groupby_*, every result compared with arrow::compute::CallFunction / arrow::compute::Grouper itself (Arrow C++ 25 from
the pyarrow wheel) on seeded inputs: sliced arrays with non-zero offsets, nulls, NaN, 0 / 1 / 17 / 1e5 rows.  Skipped when the wheel
(headers + libarrow) is not on the box."""
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_reference_side_binding_against_real_arrow():
    import __graft_entry__ as ge

    try:
        exe = ge.build_arrow_bridge_test()
    except (ImportError, FileNotFoundError, StopIteration) as e:
        pytest.skip(f"pyarrow wheel (Arrow C++ headers / libarrow) not available: {e}")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    assert "0 mismatches" in p.stdout, p.stdout[-2000:]
