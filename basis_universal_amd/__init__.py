"""basis_universal_amd -- MI355X-native hot path of the Basis Universal encoder.

Layers:
  csrc/   hand-written gfx950 HIP kernels + the C ABI (include/basisu_hip.h) -> lib/libbasisu_hip.so
  capi    ctypes binding of that C ABI (no torch types cross the boundary)
  etc1s   host-side mirror of the reference's basisu_frontend over the device-resident layer

There is deliberately no CPU fallback anywhere in this package: if the HIP library is missing or no GPU is visible the
entry points raise.
"""
from .capi import HipLibrary, HipError, load_library, LIB_PATH  # noqa: F401

__all__ = ["HipLibrary", "HipError", "load_library", "LIB_PATH"]
