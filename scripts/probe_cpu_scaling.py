"""oracle thread scaling on this box: physics-only and raster-only rates for a sweep of thread counts (cgroup quota printed first)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from megaverse_amd.rollout import action_masks, sample_actions
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "loadavg", os.getloadavg())
os.environ["MVO_PIN"] = os.environ.get("MVO_PIN", "1")
N = 1024
masks = [action_masks(sample_actions(1234, st, N)) for st in range(64)]
for T in [int(x) for x in (sys.argv[1:] or ["1", "4", "16", "32", "64", "128", "256"])]:
    g = oracle_lib.OracleGym("TowerBuilding", 128, 128, N, 1, T, False, {}); g.set_raster(True); g.seed(42); g.reset()
    for st in range(10):
        g.set_action_masks(masks[st]); g.step_norender()
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 1.0:
        g.set_action_masks(masks[k % 64]); g.step_norender(); k += 1
    ph = N * k / (time.perf_counter() - t0)
    t0 = time.perf_counter(); r = 0
    while time.perf_counter() - t0 < 1.5:
        g.render(); r += 1
    ra = N * r / (time.perf_counter() - t0)
    print(f"T={T:4d} physics {ph/1e3:8.1f} k steps/s   raster {ra:9.0f} frames/s", flush=True)
    g.close()
