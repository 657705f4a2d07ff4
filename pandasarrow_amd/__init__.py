"""pandasarrow_amd -- MI355X (gfx950) execution backend for PandasArrow's vectorized operator path.

Layout:
  csrc/       hand-written HIP kernels + the C ABI (include/pdx/abi.h) -> csrc/libpdx_hip.so
  cpp/        C++ facade mirroring pd::Series / DataFrame / GroupBy / Resampler over the C ABI
  _lib.py     ctypes binding of the C ABI (fails loudly when the library is missing; no CPU fallback)
  column.py   device-resident Arrow-layout columns (torch owns the memory) + one call per ABI entry point
  api.py      Python mirror of the reference operator interface (used by the parity tests)
  dist.py     row-range sharding across GPUs (torch.distributed / RCCL)
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
