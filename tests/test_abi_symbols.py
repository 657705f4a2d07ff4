"""CPU-only: the C-ABI library builds, loads, and exports exactly the entry points include/pdx/abi.h declares
(no compute calls -- there is no GPU here); the product refuses to run without a device instead of falling back."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge

    ge.build_hip()
    from pandasarrow_amd import _lib

    return _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pdx", "abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pdx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    assert declared_symbols() == sorted(lib.ABI_SYMBOLS), "include/pdx/abi.h and pandasarrow_amd/_lib.py drifted apart"


def test_library_exports_every_symbol(lib):
    so = C.CDLL(lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(so, name), f"libpdx_hip.so does not export {name}"
    assert lib.load().pdx_abi_version() == 1


def test_loaded_library_was_built_from_these_sources(lib):
    """pdx_build_info() carries the digest of the sources the library was compiled from (written into runtime.hip's object by
    __graft_entry__.build_hip): the in-tree .so that the GPU tests load cannot be a leftover of older sources"""
    import __graft_entry__ as ge
    import bench

    info = lib.load().pdx_build_info().decode()
    assert info.startswith("pdx-hip abi 1 gfx950 sources "), info
    assert info.split()[-1] == ge.source_hash() == bench.source_hash(), (info, ge.source_hash())


def test_no_cpu_fallback(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    handle = lib.load()
    assert handle.pdx_init(0) != lib.OK  # fails loudly, with a message
    assert b"device" in handle.pdx_last_error().lower()
    from pandasarrow_amd import column

    with pytest.raises(lib.PdxError):
        column.Column.from_numpy([1, 2, 3])


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pandasarrow_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "pdx_oracle" not in src and "liboracle" not in src, f
