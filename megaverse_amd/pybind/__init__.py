"""pybind11 flavour of the binding: `from megaverse_amd.pybind import megaverse` gives a module with the reference's exact
table (megaverse.extension.megaverse: MegaverseGym, set_megaverse_log_level) on top of libmegaverse_hip.so.  Built in-tree by
megaverse_amd/build.py (g++ + pybind11 headers, no CMake)."""
try:   # see megaverse_amd/extension.py:load_library -- torch's bundled HIP runtime must be loaded before the system one
    import torch  # noqa: F401
except ImportError:
    pass
from . import megaverse  # noqa: E402,F401
