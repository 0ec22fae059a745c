"""How much does running sub-batches of one job on several HIP streams overlap the latency-bound step kernel with the issue-bound raster?
(experiment; prints obs/s for one gym and for the same envs split over K gyms / streams)"""
import sys, time
import torch
sys.path.insert(0, ".")
from megaverse_amd.extension import MegaverseGym
from megaverse_amd.multitask import MultiTaskGym

N, A, W, H, STEPS = 1024, 1, 128, 128, 3000


def run(make, n=N):
    g = make()
    g.seed(42); g.reset()
    for st in range(50):
        g.sample_random_actions(1234, st); g.step()
    g.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for st in range(50, 50 + STEPS):
        g.sample_random_actions(1234, st); g.step()
    g.synchronize(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    g.close()
    return n * A * STEPS / dt / 1e6, dt / STEPS * 1e3


def single():
    g = MegaverseGym("TowerBuilding", W, H, N, A, 8, False, {})
    g.set_pixel_mode("fast")
    return g


def split(k, n=N):
    def make():
        mt = MultiTaskGym(["TowerBuilding"] * k, W, H, n, A, 8)
        mt.attach("cuda:0"); mt.set_pixel_mode("fast")
        return mt
    return make


print("one gym           %.2f M obs/s  %.4f ms/step" % run(single))
for k in (2, 4, 8):
    print("%d gyms / streams  %.2f M obs/s  %.4f ms/step" % ((k,) + run(split(k))))

print("2 x 1024 envs on 2 streams  %.2f M obs/s  %.4f ms/step" % run(split(2, 2048), 2048))
print("one gym (again)   %.2f M obs/s  %.4f ms/step" % run(single))
print("4 gyms (again)    %.2f M obs/s  %.4f ms/step" % run(split(4)))
print("2 gyms (again)    %.2f M obs/s  %.4f ms/step" % run(split(2)))
