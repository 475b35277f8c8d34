"""gpuspectral_amd -- MI355X-native Monte-Carlo path-tracing integrator.

Drop-in for the ``PathTracer`` pass of sunho/GPUSpectral: hand-written HIP
wavefront tracer (``csrc/``) behind the C ABI of ``include/gpuspectral_pt.h``,
with a C++ host layer (``host/``) that mirrors the reference's Scene / Loader /
PathTracer interfaces.  This Python package is only a ctypes binding for tests
and ``bench.py``; there is no CPU rendering path.
"""
from . import abi  # noqa: F401
from .pt import Context, GspError, MultiContext, device_count, lib_path  # noqa: F401

__all__ = ["abi", "Context", "MultiContext", "GspError", "device_count", "lib_path"]
