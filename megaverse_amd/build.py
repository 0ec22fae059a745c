"""Build helper: compiles megaverse_amd/libmegaverse_hip.so (hipcc, gfx950) in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("MV_LIB_PATH") or os.path.join(HERE, "libmegaverse_hip.so")   # MV_LIB_PATH: an experiment's variant build
CSRC = os.path.join(HERE, "csrc")


def sources():
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h", ".cpp"))]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "megaverse_hip.h"))
    return srcs


def is_stale():
    if os.environ.get("MV_LIB_PATH"):
        return False
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force=False, quiet=True):
    """hipcc --offload-arch=gfx950 ... -> libmegaverse_hip.so (cross-compiles without a GPU).  Several ranks importing the package at
    once must not run make concurrently: the build is serialised on a lock file, the ranks that waited find the library fresh."""
    if force or is_stale():
        import fcntl
        with open(os.path.join(HERE, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if force or is_stale():
                    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))] + (["-B"] if force else [])
                    subprocess.check_call(cmd, stdout=subprocess.DEVNULL if quiet else None)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


PYBIND_DIR = os.path.join(HERE, "pybind")


def pybind_module_path():
    import sysconfig
    return os.path.join(PYBIND_DIR, "megaverse" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_pybind(force=False, quiet=True):
    """g++ + pybind11 headers -> megaverse_amd/pybind/megaverse.<abi>.so, the module with the reference's own table
    (megaverse.extension.megaverse) on top of libmegaverse_hip.so.  Optional: the ctypes binding needs none of it."""
    import sysconfig
    import pybind11
    out, src = pybind_module_path(), os.path.join(PYBIND_DIR, "megaverse_module.cpp")
    build(force=False, quiet=quiet)
    fresh = os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(LIB))
    if force or not fresh:
        cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-I", pybind11.get_include(),
               "-I", sysconfig.get_paths()["include"], "-I", os.path.join(os.path.dirname(HERE), "include"), src, "-o", out,
               "-L", HERE, "-lmegaverse_hip", "-Wl,-rpath,$ORIGIN/.."]
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL if quiet else None)
    return out
