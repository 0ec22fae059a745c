"""Build helper: compiles megaverse_amd/libmegaverse_hip.so (hipcc, gfx950) in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmegaverse_hip.so")
CSRC = os.path.join(HERE, "csrc")


def sources():
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h", ".cpp"))]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "megaverse_hip.h"))
    return srcs


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force=False, quiet=True):
    """hipcc --offload-arch=gfx950 ... -> libmegaverse_hip.so (cross-compiles without a GPU)."""
    if force or is_stale():
        cmd = ["make", "-C", CSRC] + (["-B"] if force else [])
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL if quiet else None)
    return LIB
