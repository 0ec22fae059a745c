"""Soak run (not a pytest): long open-loop rollouts with short episodes -- thousands of auto-resets, ring refills, status read-backs, the step
kernels pipelined ahead of the raster -- twice per scenario; the two runs must end in the same state and the same observation slab, and
nothing may raise.  python scripts/soak.py [steps]"""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch
from megaverse_amd.extension import MegaverseGym
from hip_util import hip_snapshot

os.environ.setdefault("BOXOBAN_LEVELS", "tests/golden/boxoban")
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
CASES = [("TowerBuilding", 256, 2, {"episodeLengthSec": -215.0}), ("ObstaclesEasy", 256, 1, {}), ("Collect", 128, 2, {"episodeLengthSec": 3.0}),
         ("Rearrange", 256, 1, {"episodeLengthSec": 1.0}), ("Sokoban", 128, 2, {"episodeLengthSec": 2.0}), ("HexMemory", 128, 2, {"episodeLengthSec": 2.0}),
         ("HexExplore", 128, 1, {"episodeLengthSec": 4.0})]


def run(scenario, N, A, params):
    g = MegaverseGym(scenario, 64, 36, N, A, 8, False, params)
    g.set_pixel_mode("fast")
    obs = torch.zeros((N * A, 36, 64, 4), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    g.set_obs_buffer(obs.data_ptr())
    g.seed(123); g.reset()
    dones = 0
    t0 = time.perf_counter()
    st = 0
    while st < STEPS:
        if (st // 64) % 2 == 0 or st + 4 > STEPS:     # blocks of single ticks ...
            g.sample_random_actions(99, st)
            g.step()
            st += 1
        else:                                          # ... and blocks of batched calls (same action stream: tick j of a call draws from (seed, st + j))
            g.step_n(4, "multidiscrete", 99, st)
            st += 4
        if st % 997 < 4:
            dones += int(g.get_dones().sum())          # (a host read now and then: the mirrors path)
    g.synchronize(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    snaps = b"".join(hip_snapshot(g, e).tobytes() for e in range(0, N, 7))
    slab = obs.cpu().numpy().copy()
    g.close()
    return snaps, slab, dones, STEPS * N * A / dt


for scenario, N, A, params in CASES:
    a = run(scenario, N, A, params)
    b = run(scenario, N, A, params)
    same = a[0] == b[0] and np.array_equal(a[1], b[1])
    print("%-14s %d steps x %d envs x %d agents: %s, sampled dones %d, %.2f M obs/s" % (scenario, STEPS, N, A, "identical" if same else "DIFFERENT", a[2], a[3] / 1e6), flush=True)
    assert same, scenario


# ---- second part: long episodes, the one-launch batched calls (multi-tick step launch, one-launch observation passes into a ring two calls deep, the
# passes of consecutive calls overlapped) over thousands of calls -- the cost histograms go round their ring hundreds of times, each cleared by the pass
# that drew from it -- against the same rollout stepped tick by tick: state of the sampled envs and the last tick's slab
CASES2 = [("TowerBuilding", 256, 1), ("TowerBuilding", 96, 3), ("ObstaclesHard", 192, 1), ("Collect", 128, 1), ("Rearrange", 128, 1), ("Sokoban", 128, 1), ("HexMemory", 96, 1),
          ("HexExplore", 96, 1)]
STEPS2 = max(8, (STEPS // 3) // 8 * 8)


def run2(scenario, N, A, batched):
    W, H, R = 64, 36, 16
    g = MegaverseGym(scenario, W, H, N, A, 8, False, {})
    g.set_pixel_mode("fast")
    ring = torch.zeros((R, N * A, H, W, 4), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    g.set_output_ring(R, ring.data_ptr())
    if batched:
        g.set_pass_overlap(True)
    g.seed(321); g.reset()
    t0 = time.perf_counter()
    st = 0
    while st < STEPS2:
        if batched:
            g.step_n(8, "multidiscrete", 77, st); st += 8
        else:
            g.sample_random_actions(77, st); g.step(); st += 1
    g.synchronize(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    snaps = b"".join(hip_snapshot(g, e).tobytes() for e in range(0, N, 5))
    last = ring[(STEPS2 - 1) % R].cpu().numpy().copy()
    g.close()
    return snaps, last, STEPS2 * N * A / dt


for scenario, N, A in CASES2:
    a = run2(scenario, N, A, True)
    b = run2(scenario, N, A, False)
    same = a[0] == b[0] and np.array_equal(a[1], b[1])
    print("%-14s %d steps x %d envs x %d agents, calls of 8 (one launch each, overlapped passes) against single ticks: %s, %.2f / %.2f M obs/s" %
          (scenario, STEPS2, N, A, "identical" if same else "DIFFERENT", a[2] / 1e6, b[2] / 1e6), flush=True)
    assert same, scenario
print("soak ok")
