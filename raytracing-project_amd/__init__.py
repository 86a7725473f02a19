"""MI355X (gfx950) implementation of the reference ray tracer's per-pixel render loop.

The product is csrc/librt_hip.so (HIP kernels behind the C ABI of include/rt_hip.h) and
csrc/librt_host.so (C++ host side mirroring the reference's raytracing.h surface).
The Python modules here only bind those libraries for tests and bench.py.
"""
from . import capi, distributed, host, tiles  # noqa: F401
