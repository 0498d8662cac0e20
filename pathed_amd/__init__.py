"""pathed_amd — MI355X-native drop-in for pathed's PathTracer radiance loop.

Python here is plumbing over two native libraries:
  lib/libpathed_hip.so   HIP kernels + the C ABI of include/pathed_hip.h (the product)
  lib/libpathed_host.so  C++ host side: job.json / scene readers, Integrator, EXR
"""
from . import _capi  # noqa: F401
from .scene import LoadedScene  # noqa: F401

__version__ = "0.1.0"
