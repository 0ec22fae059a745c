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
    g.sample_random_actions(1234, st); g.step()
out = (C.c_ulonglong * 8)()
lib.mv_debug_raster_stats(out)
t, s, gs, nv, fr, ns = out[0], out[1], out[2], out[3], out[4], out[5]
print(f"tiles {t}  survivors/tile {s/t:.2f}  straddler survivors/tile {gs/t:.2f}  nVis/frame {nv/fr:.1f}  straddlers/frame {ns/fr:.2f}")
lib.mv_debug_raster_times.argtypes=[C.c_void_p]*3+[C.c_int]
N = 1024
t0 = np.zeros(N, np.uint64); tp = np.zeros(N, np.uint64); t1 = np.zeros(N, np.uint64)
lib.mv_debug_raster_times(t0.ctypes.data, tp.ctypes.data, t1.ctypes.data, N)   # clear
g.sample_random_actions(1234, 999); g.step(); g.synchronize()
lib.mv_debug_raster_times(t0.ctypes.data, tp.ctypes.data, t1.ctypes.data, N)
base = t0.min()
start = (t0 - base).astype(np.float64); pro = (tp - t0).astype(np.float64); dur = (t1 - t0).astype(np.float64); end = (t1 - base).astype(np.float64)
print("wall_clock ticks (100MHz => 10ns): start min/med/max", start.min(), np.median(start), start.max())
print("prologue med/max", np.median(pro), pro.max(), " duration min/med/p90/max", dur.min(), np.median(dur), np.percentile(dur, 90), dur.max())
print("end min/med/max", end.min(), np.median(end), end.max())
