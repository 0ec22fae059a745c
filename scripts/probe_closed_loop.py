"""closed loop, one gym (bench.py's closed_loop leg on its own, for a kernel trace): python scripts/probe_closed_loop.py [envs] [ticks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaverse_amd.extension import MegaverseGym

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = "cuda:0"
sizes = torch.tensor([3, 3, 3, 2, 2, 3], dtype=torch.int32, device=dev)
g = MegaverseGym("TowerBuilding", 128, 128, n, 1, 8, False, {})
g.set_stream(torch.cuda.current_stream().cuda_stream)
g.set_pixel_mode("fast")
slab = torch.zeros((n, 128, 128, 4), dtype=torch.uint8, device=dev)
g.set_obs_buffer(slab.data_ptr())
g.seed(42); g.reset()
feat = slab.view(n, -1)[:, 37:37 + 6 * 97:97]
acts = torch.zeros((n, 6), dtype=torch.int32, device=dev)


def tick():
    torch.remainder(feat, sizes, out=acts)
    g.set_actions_device(acts.data_ptr())
    g.step()


for i in range(30):
    tick()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(ticks):
    tick()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("closed loop: %.2f us per tick of %d envs (%.2f M obs/s); host enqueue %.2f us per tick" % ((t2 - t0) / ticks * 1e6, n, n * ticks / (t2 - t0) / 1e6, (t1 - t0) / ticks * 1e6))
g.close()
