"""host cost of one mv_step: a tiny gym (GPU work negligible), steps per second with and without the pipelining"""
import os, sys, time
sys.path.insert(0, ".")
import torch
from megaverse_amd.extension import MegaverseGym

def run(scenario, pipe, n=8, steps=60):
    g = MegaverseGym(scenario, 16, 16, n, 1, 2, False, {})
    g.set_pixel_mode("fast"); g.set_pipelining(pipe)
    g.seed(1); g.reset()
    for st in range(200):
        g.sample_random_actions(1, st); g.step()
    g.synchronize()
    t0 = time.perf_counter()
    for st in range(200, 200 + steps):
        g.sample_random_actions(1, st); g.step()
    t1 = time.perf_counter()
    g.synchronize()
    t2 = time.perf_counter()
    g.close()
    return (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6

os.environ.setdefault("BOXOBAN_LEVELS", "tests/golden/boxoban")
for scen in ("TowerBuilding", "ObstaclesHard", "Sokoban"):
    for pipe in (True, False):
        e, t = run(scen, pipe)
        print("%-14s pipelined=%-5s enqueue %.1f us/step, total %.1f us/step" % (scen, pipe, e, t))
