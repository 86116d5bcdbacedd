"""akari_render_amd -- an MI355X (gfx950) implementation of akari_render's `pt` path-tracing integrator.

The product is libakari_hip.so (hand-written HIP kernels + a C++ host layer behind the C ABI of
include/akari_hip.h); this package only holds its sources (csrc/), the build script and a ctypes binding.
"""
from . import abi  # noqa: F401

__all__ = ["abi"]
