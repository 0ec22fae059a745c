import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import megaverse_amd.build as b
b.LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_stats_lib.so")
b.is_stale = lambda: False
import megaverse_amd.extension as ext
lib = ext.load_library()
g = ext.MegaverseGym("TowerBuilding", 128, 128, 1024, 1, 1, False, {})
g.seed(42); g.reset()
for st in range(300):
    g.sample_random_actions(1234, st); g.step_no_render()
out = (C.c_ulonglong * 16)()
lib.mv_debug_step_stats.argtypes = [C.c_void_p] * 3
lib.mv_debug_step_stats(C.addressof(out), None, None)
n = out[15]
names = ["kernarg", "actions[env] dword", "hdr.L dword", "hdr.pad[8] dword (same line)", "agents.pad dword", "objects (64 lanes)", "hdr struct 128B"]
for i, nm in enumerate(names): print(f"  {nm:32s} {out[i]/n*10:.0f} ns")
print("  rest", (out[10])/n*10, "ns")
