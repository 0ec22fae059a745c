"""scripts/soak.py at pytest length: the one-launch batched calls -- multi-tick step launch, the k observation passes as one launch into a
ring two calls deep, the passes of consecutive calls overlapped, cost histograms cleared by the pass that drew from them -- over dozens of calls, against the
SAME rollout stepped tick by tick through mv_step: the whole state of sampled envs and the last tick's slab, byte for byte; and the mixed single / batched
rollout with short episodes (auto-resets, ring refills, status read-backs) run twice.  HIP against HIP by construction -- the oracle comparisons of these
paths are tests/test_full_size_oracle_gpu.py and tests/test_pipelining_gpu.py; this one is about length: histogram rings going round many times."""
import os

import numpy as np
import pytest

from hip_util import hip_snapshot
from megaverse_amd.extension import MegaverseGym

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
BOXOBAN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "boxoban")


@pytest.mark.parametrize("scenario,N,A", [("TowerBuilding", 256, 1), ("TowerBuilding", 96, 3), ("ObstaclesHard", 192, 1), ("Collect", 128, 1), ("Rearrange", 128, 1),
                                          ("Sokoban", 128, 1), ("HexMemory", 96, 1), ("HexExplore", 96, 1), ("Empty", 128, 1)])
def test_batched_overlapped_calls_equal_single_ticks(hip, monkeypatch, scenario, N, A):
    import torch
    monkeypatch.setenv("BOXOBAN_LEVELS", BOXOBAN)
    W, H, R, STEPS = 64, 36, 16, 640

    def run(batched):
        g = MegaverseGym(scenario, W, H, N, A, 8, False, {})
        g.set_pixel_mode("fast")
        ring = torch.zeros((R, N * A, H, W, 4), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        g.set_output_ring(R, ring.data_ptr())
        if batched:
            g.set_pass_overlap(True)
        g.seed(321); g.reset()
        st = 0
        while st < STEPS:
            if batched:
                g.step_n(8, "multidiscrete", 77, st); st += 8
            else:
                g.sample_random_actions(77, st); g.step(); st += 1
        g.synchronize(); torch.cuda.synchronize()
        snaps = b"".join(hip_snapshot(g, e).tobytes() for e in range(0, N, 5))
        last = ring[(STEPS - 1) % R].cpu().numpy().copy()
        g.close()
        return snaps, last

    a, b = run(True), run(False)
    assert a[0] == b[0], f"{scenario}: state after {STEPS} ticks differs between calls of 8 and single ticks"
    assert np.array_equal(a[1], b[1]), f"{scenario}: the last tick's slab differs"
    assert int(a[1][..., :3].max()) > 0


@pytest.mark.parametrize("scenario,N,A,params", [("TowerBuilding", 128, 2, {"episodeLengthSec": -215.0}), ("Collect", 64, 2, {"episodeLengthSec": 3.0}),
                                                 ("HexExplore", 64, 1, {"episodeLengthSec": 4.0})])
def test_short_episode_rollouts_are_reproducible(hip, scenario, N, A, params):
    import torch
    STEPS = 1500

    def run():
        g = MegaverseGym(scenario, 64, 36, N, A, 8, False, params)
        g.set_pixel_mode("fast")
        obs = torch.zeros((N * A, 36, 64, 4), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        g.set_obs_buffer(obs.data_ptr())
        g.seed(123); g.reset()
        st = dones = 0
        while st < STEPS:
            if (st // 64) % 2 == 0 or st + 4 > STEPS:   # blocks of single ticks and blocks of batched calls, the same action stream
                g.sample_random_actions(99, st); g.step(); st += 1
            else:
                g.step_n(4, "multidiscrete", 99, st); st += 4
            if st % 97 < 4:
                dones += int(g.get_dones().sum())
        g.synchronize(); torch.cuda.synchronize()
        snaps = b"".join(hip_snapshot(g, e).tobytes() for e in range(0, N, 7))
        slab = obs.cpu().numpy().copy()
        g.close()
        return snaps, slab, dones

    a, b = run(), run()
    assert a[0] == b[0] and np.array_equal(a[1], b[1]), scenario
    assert a[2] > 0, "no episode ended inside the sampled ticks"
