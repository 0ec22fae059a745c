"""GPU parity for the Obstacles family: host-generated episodes + HIP step/raster against
the CPU oracle, bit-exact state / rewards / dones / pixels."""
import numpy as np
import pytest

from hip_util import diff_snapshots, hip_snapshot, make_pair, set_same_actions

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
SCENARIOS = ["ObstaclesEasy", "ObstaclesMedium", "ObstaclesHard", "ObstaclesWalls", "ObstaclesSteps", "ObstaclesLava"]


def frames(g, N, A):
    return np.stack([g.get_observation(e, a) for e in range(N) for a in range(A)])


@pytest.mark.parametrize("scenario", SCENARIOS)
@pytest.mark.parametrize("A", [1, 3])
def test_reset_parity(hip, scenario, A):
    N = 24
    og, hg = make_pair(N, A, 32, 32, seed=5 + A, scenario=scenario)
    for e in range(N):
        d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
        assert not d, (e, d[:5])
    og.close(); hg.close()


@pytest.mark.parametrize("scenario", SCENARIOS)
def test_pixels_after_reset(hip, scenario):
    N, A = 8, 2
    og, hg = make_pair(N, A, 128, 72, seed=31, scenario=scenario)
    fo, fh = frames(og, N, A), frames(hg, N, A)
    assert np.array_equal(fo, fh), f"{int((fo != fh).sum())} differing bytes"
    og.close(); hg.close()


@pytest.mark.parametrize("scenario,A,seed", [("ObstaclesHard", 1, 1), ("ObstaclesHard", 2, 2), ("ObstaclesEasy", 4, 3),
                                             ("ObstaclesLava", 2, 4), ("ObstaclesSteps", 1, 5), ("ObstaclesWalls", 3, 6),
                                             ("ObstaclesMedium", 8, 7)])
def test_rollout_parity(hip, scenario, A, seed):
    """state, rewards, dones every step; the ObstaclesEasy case runs long enough to cross auto-resets"""
    N = 8
    og, hg = make_pair(N, A, 32, 32, seed=seed, scenario=scenario)
    resets = 0
    steps = 1300 if scenario == "ObstaclesEasy" else 700   # Easy: 1-2 platforms, 35 s each -> episodes of 525..1050+ ticks
    for st in range(steps):
        set_same_actions(og, hg, N, A, 100 + seed, st)
        og.step_norender(); hg.step_no_render()
        ro, rh = og.get_last_rewards(), hg.get_rewards_array()
        assert ro.tobytes() == rh.tobytes(), (st, ro, rh)
        do = np.array([og.is_done(e) for e in range(N)]); dh = hg.get_dones()
        assert np.array_equal(do, dh.astype(bool)), (st, do, dh)
        resets += int(do.sum())
        if st % 25 == 0 or do.any():
            for e in range(N):
                d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
                assert not d, (st, e, d[:5])
        to = np.array([og.true_objective(e, a) for e in range(N) for a in range(A)], np.float32)
        assert to.tobytes() == hg.get_true_objectives().tobytes()
    og.render(); hg.render()
    assert np.array_equal(frames(og, N, A), frames(hg, N, A))
    if scenario == "ObstaclesEasy":
        assert resets > 0, "rollout never exercised the auto-reset / episode refill path"
    og.close(); hg.close()


def test_rollout_pixels_every_25_steps(hip):
    N, A = 4, 2
    og, hg = make_pair(N, A, 64, 64, seed=77, scenario="ObstaclesHard")
    for st in range(400):
        set_same_actions(og, hg, N, A, 9, st)
        if st % 25 == 24:
            og.step(); hg.step()
            fo, fh = frames(og, N, A), frames(hg, N, A)
            assert np.array_equal(fo, fh), (st, int((fo != fh).sum()))
        else:
            og.step_norender(); hg.step_no_render()
    og.close(); hg.close()


def test_reward_shaping_keys(hip):
    og, hg = make_pair(2, 2, 32, 32, seed=1, scenario="ObstaclesWalls")
    keys = list(hg.get_reward_shaping(0, 0).keys())
    assert keys == ["teamSpirit", "obstaclesAgentAtExit", "obstaclesAllAgentsAtExit", "obstaclesExtraReward",
                    "obstaclesAgentCarriedObjectToExit"]
    assert hg.get_reward_shaping(0, 1)["obstaclesAgentCarriedObjectToExit"] == 1.0
    assert hg.get_reward_shaping(0, 1)["obstaclesAllAgentsAtExit"] == 5.0
    og.close(); hg.close()


def test_reseed_mid_run_takes_effect_at_the_next_reset(hip):
    """Env::seed re-seeds the env's stream immediately (env.cpp:52-55): the episode that was generated ahead of time
    from the old stream must not be used"""
    N, A = 6, 2
    og, hg = make_pair(N, A, 32, 32, seed=21, scenario="ObstaclesMedium")
    for st in range(30):
        set_same_actions(og, hg, N, A, 5, st)
        og.step_norender(); hg.step_no_render()
    og.seed(99); hg.seed(99)
    og.reset(); hg.reset()
    for e in range(N):
        d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
        assert not d, (e, d[:5])
    for st in range(20):
        set_same_actions(og, hg, N, A, 6, st)
        og.step_norender(); hg.step_no_render()
    og.reset(); hg.reset()
    for e in range(N):
        d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
        assert not d, (e, d[:5])
    og.close(); hg.close()


def test_forced_reset_right_after_auto_resets(hip):
    """mv_reset a few ticks after some envs auto-reset: the host's periodic status read-back is then out of date and must
    be refreshed, or the forced reset would find an already consumed episode resident"""
    N, A = 16, 1
    og, hg = make_pair(N, A, 32, 32, seed=77, scenario="ObstaclesEasy")
    st, first_done = 0, None
    while first_done is None or st < first_done + 3:
        set_same_actions(og, hg, N, A, 41, st)
        og.step_norender(); hg.step_no_render()
        if first_done is None and any(og.is_done(e) for e in range(N)):
            first_done = st
        st += 1
        assert st < 1200
    og.reset(); hg.reset()
    for e in range(N):
        d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
        assert not d, (e, d[:5])
    for k in range(40):
        set_same_actions(og, hg, N, A, 42, k)
        og.step_norender(); hg.step_no_render()
    for e in range(N):
        d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
        assert not d, (e, d[:5])
    og.close(); hg.close()
