"""typesense_amd — MI355X (gfx950) implementation of Typesense's query-time scoring hot path.

The product is the C-ABI shared library built from typesense_amd/csrc (include/tsgpu.h). This Python package is
only a thin ctypes binding over that ABI for tests, bench.py and multi-GPU plumbing (torch.distributed); it has
no compute of its own and NO CPU fallback: importing `typesense_amd.lib()` raises if libtsgpu.so is missing.
"""
from ._lib import lib, TsgpuError, LIB_PATH  # noqa: F401
from .index import GpuIndex, GpuGroup, KwQuery, Hits, GroupedHits  # noqa: F401
