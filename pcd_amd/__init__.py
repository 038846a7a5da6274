"""pcd_amd -- MI355X (gfx950) prover arithmetic for arkworks-style proof-carrying data.

The product is the C-ABI shared library `pcd_amd/libpcdhip.so` (include/pcdhip.h) built from the
hand-written HIP sources in `pcd_amd/csrc/`.  This package is a thin ctypes binding used by the
tests and by bench.py; it has no CPU path: importing `pcd_amd.capi` without the built library, or
creating a context without a GPU, raises.
"""
from .build import build_library, library_path  # noqa: F401
