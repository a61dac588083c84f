"""illuminant_amd -- MI355X-native implementation of sq/Illuminant's two data-parallel hot paths.

The product is libilluminant_hip.so (hand-written HIP for gfx950 behind the C ABI of
include/illuminant_hip.h).  This package holds the kernels (csrc/), the ctypes binding of the
ABI (native.py, abi.py), the host-side mirror of the reference's ParticleSystem / LightingRenderer
interface (host/: a C++ library over the C ABI + its pybind11 module _host), the strip / chunk-ownership arithmetic of the
multi-GPU path (sharding.py) and synthetic scene generators (scenes.py).
"""
__all__ = ["abi", "native", "scenes"]
