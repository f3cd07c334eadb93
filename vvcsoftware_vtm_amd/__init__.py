"""vvcsoftware_vtm_amd -- MI355X-native pixel hot path for VTM (reference studied: VTM 2.1).

The product is the C-ABI shared library `lib/libvvcgpu.so` (hand-written HIP for gfx950, sources in
`csrc/`, interface in `/include/vvcgpu.h`).  This Python package is the host-side plumbing used by the
tests and by bench.py: a ctypes binding (`capi`) and thin operator mirrors of the reference classes
(`ops`).  There is NO CPU fallback: every operator raises if the HIP library is missing.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
