"""megaverse_amd: MI355X-native batched voxel-world simulator behind the MegaverseEnv surface.

Hot path (reference VectorEnv::step) = hand-written HIP kernels for gfx950 in csrc/, reached through
the C ABI in include/megaverse_hip.h.  This package is the thin host mirror of the reference's
Python surface (megaverse/megaverse_env.py) plus batched / multi-GPU helpers.
"""
from .extension import MegaverseGym, set_megaverse_log_level, load_library  # noqa: F401
from .megaverse_env import MegaverseEnv, MEGAVERSE8, OBSTACLES_MULTITASK, make_env_multitask  # noqa: F401
from .multitask import MultiTaskGym, MEGAVERSE_IN_SCOPE  # noqa: F401
from .rl import make_megaverse, MEGAVERSE_ENVS  # noqa: F401
