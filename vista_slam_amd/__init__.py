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


def install_as_reference():
    """Make `from vista_slam.sta_model.sta_model import SymmetricTwoViewAssociation` resolve to `STAFrontend` - with ZERO edits
    to the reference tree: `vista_slam/slam.py:9` imports the model class by that path and `:96` calls `STA()` with no
    arguments; after this call (made once, before `import vista_slam.slam`) those two lines construct the MI355X frontend.

    Only the two module names of that import path are registered in `sys.modules` (a stub package `vista_slam.sta_model` whose
    one member is the stub module `vista_slam.sta_model.sta_model`); the reference's own `vista_slam` package, `slam.py`,
    `pose_graph.py`, `loop_detector.py`, ... are imported from wherever they are installed and stay untouched.  The reference's
    PyTorch model files are then never imported (nor xformers / curope, which they need).  Returns the stub module."""
    import sys
    import types
    from .sta_frontend import STAFrontend
    pkg = types.ModuleType("vista_slam.sta_model")
    pkg.__path__ = []                       # a package with no files: every submodule must already be in sys.modules
    pkg.__doc__ = "stub installed by vista_slam_amd.install_as_reference()"
    mod = types.ModuleType("vista_slam.sta_model.sta_model")
    mod.__doc__ = "vista_slam_amd.STAFrontend under the reference's import path (vista_slam/slam.py:9)"
    mod.SymmetricTwoViewAssociation = STAFrontend
    pkg.sta_model = mod
    sys.modules["vista_slam.sta_model"] = pkg
    sys.modules["vista_slam.sta_model.sta_model"] = mod
    parent = sys.modules.get("vista_slam")
    if parent is not None:
        parent.sta_model = pkg
    return mod
