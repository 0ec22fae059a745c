"""BASELINE.json's configurations AT THEIR FULL per-GPU SIZE against the oracle: the oracle's physics + scenario logic
costs ~5 ms per 1024-env tick on one thread -- only its brute-force raster is slow -- so the workloads the bench times are compared with it directly:
the HIP gym steps through the product's default path (device-drawn actions, fast pixels, pipelined) beside the oracle (mvo_step_norender);
  * rewards (bit patterns) and dones of EVERY env on EVERY tick, true objectives of the envs that finish;
  * the whole state of EVERY env (packed snapshots, bit for bit) every 50 ticks;
  * the pixels of eight sampled envs at 128 x 128 at those ticks: the exact mode byte for byte against the oracle's software raster of those envs
    (mvo_render_env), the fast mode within DESIGN.md's tolerance;
with natural and parameter-forced resets inside the run (asserted).  configs[1] TowerBuilding 1024 x 1, configs[3] 512 x 4, one GPU's share of
configs[2] (ObstaclesHard 512), one GPU's share of configs[4] (the eight megaverse8 scenarios dealt over 1024 envs)."""
import os

import numpy as np
import pytest

import oracle_lib
from hip_util import diff_snapshots, hip_snapshot
from megaverse_amd.extension import MegaverseGym
from megaverse_amd.multitask import MEGAVERSE_IN_SCOPE, MultiTaskGym
from megaverse_amd.rollout import action_masks, sample_actions

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1800)]
W = H = 128


def pixels_agree(og, hg, envs, A, what):
    hg.set_pixel_mode("exact"); hg.render()
    exact = {(e, a): hg.get_observation(e, a).copy() for e in envs for a in range(A)}
    hg.set_pixel_mode("fast"); hg.render()
    ndiff = ngt1 = npx = 0
    for e in envs:
        og.render_env(e)
        for a in range(A):
            ref = og.get_observation(e, a)
            assert np.array_equal(ref, exact[(e, a)]), f"{what}: env {e} agent {a}: exact pixels differ from the oracle"
            d = np.abs(ref.astype(np.int16) - hg.get_observation(e, a).astype(np.int16)).max(axis=-1)
            ndiff += int((d > 0).sum()); ngt1 += int((d > 1).sum()); npx += d.size
    assert ngt1 <= max(2, 1e-4 * npx) and ndiff <= max(4, 5e-4 * npx), f"{what}: fast pixels: {ndiff} differ, {ngt1} by more than 1 of {npx}"


@pytest.mark.parametrize("scenario,N,A,params,TICKS,EVERY", [
    ("TowerBuilding", 1024, 1, {"episodeLengthSec": -190.0}, 300, 50),   # configs[1] (short episodes: resets all through the run)
    ("TowerBuilding", 1024, 1, {}, 300, 50),                              # configs[1] exactly as benchmarked (episodes end naturally only after the run)
    ("TowerBuilding", 512, 4, {"episodeLengthSec": -150.0}, 300, 50),     # configs[3]
    ("ObstaclesHard", 512, 1, {}, 1200, 150),                             # one GPU's share of configs[2]: episodes of >= 70 s = 1050 ticks, the natural resets fall into the run
])
def test_full_size_rollout_equals_the_oracle(hip, scenario, N, A, params, TICKS, EVERY):
    og = oracle_lib.OracleGym(scenario, W, H, N, A, 16, False, params)
    hg = MegaverseGym(scenario, W, H, N, A, 8, False, params)
    hg.set_pixel_mode("fast")
    assert hg.pipelining()
    og.seed(42); hg.seed(42)
    og.reset(); hg.reset()
    sample = [int(e) for e in np.linspace(0, N - 1, 8)]
    ndone, rsum = 0, 0.0
    for st in range(TICKS):
        og.set_action_masks(action_masks(sample_actions(1234, st, N * A)))
        og.step_norender()
        hg.sample_random_actions(1234, st); hg.step()
        do = og.get_dones().astype(bool)
        assert np.array_equal(do, hg.get_dones().astype(bool)), f"dones differ at tick {st}: envs {np.nonzero(do != hg.get_dones().astype(bool))[0][:8].tolist()}"
        ro, rh = og.get_last_rewards(), hg.get_rewards_array()
        assert np.array_equal(ro.view(np.uint32), rh.view(np.uint32)), f"rewards differ at tick {st}: agents {np.nonzero(ro.view(np.uint32) != rh.view(np.uint32))[0][:8].tolist()}"
        ndone += int(do.sum()); rsum += float(np.abs(ro).sum())
        for e in np.nonzero(do)[0][:16]:
            for a in range(A):
                assert og.true_objective(int(e), a) == hg.true_objective(int(e), a), (st, int(e), a)
        if st % EVERY == EVERY - 1:
            for e in range(N):
                d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
                assert not d, (st, e, d[:4])
            pixels_agree(og, hg, sample, A, f"{scenario} {N}x{A} tick {st}")
    # ---- the batched open-loop path the bench's headline runs: mv_step_n, 8 ticks per call (one multi-tick step launch and, with an output ring,
    # one launch for the 8 observation passes), against the oracle's single ticks: state of every env afterwards, and the ring's last slab
    import torch
    ring = torch.zeros((8, N * A, H, W, 4), dtype=torch.uint8, device="cuda:0")
    hg.set_output_ring(8, ring.data_ptr())
    st = TICKS
    for _ in range(4):
        hg.step_n(8, "multidiscrete", 1234, st)
        for j in range(8):
            og.set_action_masks(action_masks(sample_actions(1234, st + j, N * A)))
            og.step_norender()
        st += 8
    hg.synchronize(); torch.cuda.synchronize()
    for e in range(N):
        d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
        assert not d, ("after batched calls", e, d[:4])
    last = ring[7].cpu().numpy()                 # tick st - 1 went to ring entry (32 - 1) % 8
    ndiff = ngt1 = npx = 0
    for e in sample:
        og.render_env(e)
        for a in range(A):
            dpx = np.abs(og.get_observation(e, a).astype(np.int16) - last[e * A + a].astype(np.int16)).max(axis=-1)
            ndiff += int((dpx > 0).sum()); ngt1 += int((dpx > 1).sum()); npx += dpx.size
    assert last[..., 3].min() == 255 and ngt1 <= max(2, 1e-4 * npx) and ndiff <= max(4, 5e-4 * npx), f"ring slab after batched calls: {ndiff} pixels differ, {ngt1} by more than 1"
    hg.set_output_ring(0)
    if params:
        assert ndone > N // 2, f"only {ndone} episodes ended in {TICKS} ticks"
    elif scenario != "TowerBuilding":
        assert ndone >= 20, f"only {ndone} episodes ended in {TICKS} ticks"   # (ObstaclesHard: 58 natural resets by tick 1200, episodes of 70 s and more)
    if scenario == "TowerBuilding":
        assert rsum > 0.0
    og.close(); hg.close()


def test_full_size_mixed_scenarios_equal_their_oracles(hip, monkeypatch):
    """one GPU's share of configs[4]: the eight megaverse8 scenarios dealt round-robin over 1024 envs, one union step launch and one observation
    launch per tick -- against eight oracles (the oracle has no env stride: each simulates all 1024 global envs as its scenario, the owned ones are compared):
    rewards and dones of every env on every tick, every env's whole state every 40 ticks and -- at those ticks -- the exact-mode pixels of one sampled env
    per scenario, 64 x 64, byte for byte against the oracle's software raster"""
    monkeypatch.setenv("BOXOBAN_LEVELS", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "boxoban"))
    N, A, S, ticks = 1024, 1, len(MEGAVERSE_IN_SCOPE), 120
    W = H = 64   # (configs[4]'s observation size)
    mt = MultiTaskGym(MEGAVERSE_IN_SCOPE, W, H, N, A, 8)
    mt.set_pixel_mode("fast")
    mt.attach("cuda:0")
    mt.seed(42); mt.reset()
    ogs = []
    for name in MEGAVERSE_IN_SCOPE:
        og = oracle_lib.OracleGym(name, W, H, N, A, 16)
        og.seed(42); og.reset()
        ogs.append(og)
    owned = [np.arange(k, N, S) for k in range(S)]
    for st in range(ticks):
        masks = action_masks(sample_actions(1234, st, N * A))
        for og in ogs:
            og.set_action_masks(masks)
            og.step_norender()
        mt.sample_random_actions(1234, st); mt.step()
        r = mt.get_last_rewards().reshape(N, A)
        for k in range(S):
            ro = ogs[k].get_last_rewards().reshape(N, A)[owned[k]]
            assert np.array_equal(ro.view(np.uint32), r[owned[k]].view(np.uint32)), (st, MEGAVERSE_IN_SCOPE[k])
            assert np.array_equal(ogs[k].get_dones()[owned[k]], mt.gyms[k].get_dones()), (st, MEGAVERSE_IN_SCOPE[k])
        if st % 40 == 39:
            mt.synchronize()
            for i in range(N):
                k, j = i % S, i // S
                d = diff_snapshots(ogs[k].snapshot(i), hip_snapshot(mt.gyms[k], j), A)
                assert not d, (st, i, MEGAVERSE_IN_SCOPE[k], d[:4])
            # pixels: one sampled env per scenario, exact mode (bit for bit against the oracle's software raster), then back to the product's fast mode
            for k in range(S):
                j = ((st // 40) * 37 + 5 * k) % (N // S)
                i = k + S * j
                g = mt.gyms[k]
                g.set_pixel_mode("exact"); g.render(); g.synchronize()
                ogs[k].render_env(i)
                assert np.array_equal(ogs[k].get_observation(i, 0), g.get_observation(j, 0)), (st, i, MEGAVERSE_IN_SCOPE[k], "pixels")
                g.set_pixel_mode("fast")
    for og in ogs:
        og.close()
    mt.close()
