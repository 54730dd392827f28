"""gpqhe_amd -- MI355X-native engine for GPQHE's RNS/NTT hot path.

The product is libgpqhe_hip.so (hand-written HIP for gfx950 behind the C ABI
of include/gpqhe_hip.h); this package is the thin host-side mirror of the
reference's interface used by the tests and the benchmark.
"""
from ._native import GpqError, LIB_PATH, load  # noqa: F401
from .engine import PolyContext, StreamTimer, big_to_ints, ints_to_big, to_device, to_host  # noqa: F401
