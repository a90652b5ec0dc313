"""dipgenie_amd -- MI355X-native hot path of DipGenie (diploid DP + minimizer sketching).

The product is the C-ABI shared library ``dipgenie_amd/csrc/libdipgenie_hip.so`` (hand-written HIP for
gfx950, declared in ``include/dipgenie_hip.h``) plus the C++ host pipeline / CLI in
``dipgenie_amd/host``.  This package only holds thin ctypes bindings for tests, ``bench.py`` and the
read-sharded multi-GPU sketch (``dist_sketch``).  There is no CPU fallback: importing ``capi`` fails
loudly when the library has not been built, and every entry point errors out without a gfx950 device.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
