"""host time of every batched call (mv_step_n, 8 ticks): does the host run ahead of the device or wait for it?
usage: probe_host_calls.py [envs] [calls]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from megaverse_amd.extension import MegaverseGym
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W = H = 128
g = MegaverseGym("TowerBuilding", W, H, N, 1, 8, False, {})
g.set_pixel_mode("fast")
ring = torch.zeros((8, N, H, W, 4), dtype=torch.uint8, device="cuda:0")
g.set_output_ring(8, ring.data_ptr())
g.seed(42); g.reset()
st = 0
for _ in range(10):
    g.step_n(8, "multidiscrete", 1234, st); st += 8
g.synchronize(); torch.cuda.synchronize()
ts = []
t00 = time.perf_counter()
for _ in range(calls):
    t0 = time.perf_counter()
    g.step_n(8, "multidiscrete", 1234, st); st += 8
    ts.append(time.perf_counter() - t0)
t_enq = time.perf_counter() - t00
g.synchronize(); torch.cuda.synchronize()
t_all = time.perf_counter() - t00
ts = np.array(ts) * 1e6
print("host us per call: first 6", ts[:6].round(0).tolist(), "median %.0f mean %.0f max %.0f | enqueue %.0f us/call, with drain %.0f us/call -> %.2f M obs/s" %
      (np.median(ts), ts.mean(), ts.max(), t_enq / calls * 1e6, t_all / calls * 1e6, N * 8 * calls / t_all / 1e6))
g.close()
