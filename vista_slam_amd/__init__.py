"""MI355X-native Symmetric Two-view Association (STA) frontend for ViSTA-SLAM.

Product path = `STAFrontend` (Python shim) -> libsta_mi355.so (C ABI, include/sta_mi355.h)
-> hand-written gfx950 HIP kernels (csrc/).  No CPU fallback.
"""
from .weights import STAConfig, FULL, TINY  # noqa: F401


def __getattr__(name):
    if name in ("STAFrontend", "rope2d_inplace"):
        from . import sta_frontend
        return getattr(sta_frontend, name)
    raise AttributeError(name)
