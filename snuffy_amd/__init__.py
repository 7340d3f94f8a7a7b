"""snuffy_amd -- MI355X-native (gfx950) hot path of jafarinia/snuffy behind the reference's own module API.

    from snuffy_amd import snuffy            # drop-in for the reference's snuffy.py
    net = snuffy.MILNet(snuffy.FCLayer(D, 1), snuffy.BClassifier(...))

Kernels live in snuffy_amd/csrc (HIP, built into snuffy_amd/lib/libsnuffy_hip.so, C ABI in include/snuffy_hip.h).
"""
from . import _ffi  # noqa: F401
from ._ffi import SnuffyHipError  # noqa: F401

__version__ = "0.3.0"
