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
t0 = np.zeros(1024, np.uint64); t1 = np.zeros(1024, np.uint64)
lib.mv_debug_step_stats.argtypes = [C.c_void_p] * 3
lib.mv_debug_step_stats(C.addressof(out), t0.ctypes.data, t1.ctypes.data)
n = out[15]
names = ["loads+colliders", "intents", "physics", "scenario", "timers+stores"]
for i, nm in enumerate(names): print(f"  {nm:28s} {out[i]/n*10:.0f} ns")
b = t0.min(); st = (t0 - b).astype(float) * 10; en = (t1 - b).astype(float) * 10
print("last-step wave starts ns: min/med/p90/max", st.min(), np.median(st), np.percentile(st, 90), st.max())
print("ends ns: min/med/p90/max", en.min(), np.median(en), np.percentile(en, 90), en.max(), " durations med/p90/max", np.median(en - st), np.percentile(en-st, 90), (en - st).max())
