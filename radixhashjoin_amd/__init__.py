"""radixhashjoin_amd -- MI355X (gfx950) radix hash join engine.

The product is the C-ABI shared library ``librhj_hip.so`` (hand-written HIP kernels, see
``csrc/`` and ``include/rhj.h``) plus the C++ host mirror of the reference's
``relation / relation_info / Result / JobScheduler`` surface in ``host/``.  This Python package is
only the thin ctypes binding used by tests, ``bench.py`` and the multi-GPU driver; there is NO
CPU fallback: if the HIP library is missing or no GPU is present, construction of an
:class:`Engine` raises.
"""
from .binding import (  # noqa: F401
    PAIR,
    TUPLE,
    DeviceBuffer,
    Engine,
    Opts,
    RhjError,
    Timings,
    lib_path,
    load_library,
    mix64,
    unmix64,
)

__all__ = ["Engine", "Opts", "Timings", "DeviceBuffer", "RhjError", "TUPLE", "PAIR", "lib_path", "load_library", "mix64", "unmix64"]
__version__ = "0.1.0"
