"""GPU parity for HexMemory and HexExplore: host-generated honeycomb mazes (mv_gen_hex.cpp) + HIP step (walls as
boxes in three rotated frames, per-agent broadphase into an LDS candidate list, collect / explore logic) + raster (boxes in the wall
frames, pillars / diamonds / spheres) against the CPU oracle: bit-exact state / rewards / dones / pixels."""
import numpy as np
import pytest

from hip_util import diff_snapshots, hip_snapshot, make_pair, set_same_actions

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
SCENARIOS = ["HexExplore", "HexMemory"]


def frames(g, N, A):
    return np.stack([g.get_observation(e, a) for e in range(N) for a in range(A)])


def same_state(og, hg, N, A, tag):
    for e in range(N):
        d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
        assert not d, (tag, e, d[:5])


@pytest.mark.parametrize("scenario", SCENARIOS)
@pytest.mark.parametrize("A,seed", [(1, 3), (2, 14), (5, 15), (8, 92)])
def test_reset_parity(hip, scenario, A, seed):
    N = 24
    og, hg = make_pair(N, A, 32, 32, seed=seed, scenario=scenario)
    same_state(og, hg, N, A, "reset")
    og.close(); hg.close()


@pytest.mark.parametrize("scenario", SCENARIOS)
@pytest.mark.parametrize("W,H", [(128, 72), (64, 64), (40, 24)])
def test_pixels_after_reset(hip, scenario, W, H):
    N, A = 12, 2
    og, hg = make_pair(N, A, W, H, seed=65, scenario=scenario)
    fo, fh = frames(og, N, A), frames(hg, N, A)
    bad = [i for i in range(N * A) if not np.array_equal(fo[i], fh[i])]
    assert not bad, (bad, int((fo != fh).sum()))
    assert (fo[..., :3] > 0).mean() > 0.2          # the maze is in view
    og.close(); hg.close()


@pytest.mark.parametrize("scenario", SCENARIOS)
@pytest.mark.parametrize("A,seed", [(1, 1), (2, 2), (4, 3), (8, 4)])
def test_rollout_parity(hip, scenario, A, seed):
    """state, rewards, dones every step over 1200 steps of short episodes (time-outs and auto-resets occur; HexMemory collects too)"""
    N = 8
    og, hg = make_pair(N, A, 32, 32, seed=seed, scenario=scenario, params={"episodeLengthSec": 6.0})
    resets, nonzero = 0, 0
    for st in range(1200):
        set_same_actions(og, hg, N, A, 300 + seed, st)
        og.step_norender(); hg.step_no_render()
        ro, rh = og.get_last_rewards(), hg.get_rewards_array()
        assert ro.tobytes() == rh.tobytes(), (st, ro, rh)
        nonzero += int((ro != 0).sum())
        do = np.array([og.is_done(e) for e in range(N)]); dh = hg.get_dones()
        assert np.array_equal(do, dh.astype(bool)), (st, do, dh)
        resets += int(do.sum())
        if st % 40 == 0 or do.any():
            same_state(og, hg, N, A, st)
        to = np.array([og.true_objective(e, a) for e in range(N) for a in range(A)], np.float32)
        assert to.tobytes() == hg.get_true_objectives().tobytes()
    og.render(); hg.render()
    assert np.array_equal(frames(og, N, A), frames(hg, N, A))
    assert resets > 0, resets
    og.close(); hg.close()


@pytest.mark.parametrize("scenario", SCENARIOS)
def test_rollout_pixels_every_20_steps(hip, scenario):
    N, A = 4, 2
    og, hg = make_pair(N, A, 64, 36, seed=8, scenario=scenario)
    for st in range(240):
        set_same_actions(og, hg, N, A, 19, st)
        if st % 20 == 19:
            og.step(); hg.step()
            fo, fh = frames(og, N, A), frames(hg, N, A)
            assert np.array_equal(fo, fh), (st, int((fo != fh).sum()))
        else:
            og.step_norender(); hg.step_no_render()
    og.close(); hg.close()


def test_explore_is_solved_by_reaching_the_reward_object(hip):
    N, A = 3, 2
    og, hg = make_pair(N, A, 48, 27, seed=3, scenario="HexExplore")
    t = og.snapshot(1)["hex_target"]
    for g in (og, hg):
        g.debug_set_agent_pos(1, 1, float(t[0]) + 0.4, 1.0, float(t[2]))
    og.step(); hg.step()
    ro, rh = og.get_last_rewards(), hg.get_rewards_array()
    assert ro.tobytes() == rh.tobytes() and ro.reshape(N, A)[1, 1] == 5.0
    same_state(og, hg, N, A, "solved")
    assert og.snapshot(1)["solved"] == 1
    assert np.array_equal(frames(og, N, A), frames(hg, N, A))          # the diamond is gone from both
    for st in range(6):
        og.step_norender(); hg.step_no_render()
        assert og.get_last_rewards().tobytes() == hg.get_rewards_array().tobytes()
        do = np.array([og.is_done(e) for e in range(N)])
        assert np.array_equal(do, hg.get_dones().astype(bool))
        if do[1]:
            break
    assert do[1] and hg.get_true_objectives().reshape(N, A)[1].tolist() == [1.0, 1.0]
    same_state(og, hg, N, A, "next episode")
    og.close(); hg.close()


def test_memory_collects_good_and_bad_objects_and_finishes(hip):
    N, A = 2, 2
    og, hg = make_pair(N, A, 48, 27, seed=9, scenario="HexMemory")
    s = og.snapshot(0)
    o = s["hex_objs"][: s["hex_num_objs"]]
    good = (o["meta"] >> 4) & 1
    goods = [i for i in range(1, len(o)) if good[i]]
    bads = [i for i in range(1, len(o)) if not good[i]]
    total = 0.0
    for n, i in enumerate(bads[:1] + goods):
        for g in (og, hg):
            g.debug_set_agent_pos(0, n % 2, float(o["a"][i, 0]), 0.8, float(o["a"][i, 2]))
        og.step(); hg.step()
        ro, rh = og.get_last_rewards(), hg.get_rewards_array()
        assert ro.tobytes() == rh.tobytes(), (n, ro, rh)
        assert ro[n % 2] == (1.0 if good[i] else -1.0)
        total += float(ro.sum())
        same_state(og, hg, N, A, ("collect", n))
        assert np.array_equal(frames(og, N, A), frames(hg, N, A))
    assert total == len(goods) - len(bads[:1])
    for st in range(8):
        og.step_norender(); hg.step_no_render()
        do = np.array([og.is_done(e) for e in range(N)])
        assert np.array_equal(do, hg.get_dones().astype(bool))
        if do[0]:
            break
    assert do[0] and hg.get_true_objectives().reshape(N, A)[0].tolist() == [1.0, 1.0]
    same_state(og, hg, N, A, "next episode")
    og.close(); hg.close()


def test_reward_shaping_keys(hip):
    og, hg = make_pair(2, 2, 32, 32, seed=1, scenario="HexMemory")
    assert hg.get_reward_shaping(1, 1) == {"teamSpirit": 0.0, "memoryCollectGood": 1.0, "memoryCollectBad": -1.0}
    og.close(); hg.close()
    og, hg = make_pair(2, 2, 32, 32, seed=1, scenario="HexExplore")
    assert hg.get_reward_shaping(0, 0) == {"teamSpirit": 0.0, "exploreSolved": 5.0}
    hg.set_reward_shaping(0, 1, {"exploreSolved": 2.5, "teamSpirit": 0.5})
    og.set_reward_shaping(0, 1, {"exploreSolved": 2.5, "teamSpirit": 0.5})
    t = og.snapshot(0)["hex_target"]
    for g in (og, hg):
        g.debug_set_agent_pos(0, 1, float(t[0]), 1.0, float(t[2]) - 0.3)
    og.step_norender(); hg.step_no_render()
    ro, rh = og.get_last_rewards(), hg.get_rewards_array()
    # rewardTeam: the finder gets 2.5 * (1 - 0.5) + 2.5 * 0.5 / 2, agent 0's team share is its own teamSpirit (0) times its own coefficient
    assert ro.tobytes() == rh.tobytes() and ro[1] == 1.875 and ro[0] == 0.0
    og.close(); hg.close()


def test_reseed_mid_run_takes_effect_at_the_next_reset(hip):
    N, A = 6, 2
    og, hg = make_pair(N, A, 32, 32, seed=21, scenario="HexExplore")
    for st in range(30):
        set_same_actions(og, hg, N, A, 5, st)
        og.step_norender(); hg.step_no_render()
    og.seed(99); hg.seed(99)
    og.reset(); hg.reset()
    same_state(og, hg, N, A, "reseed")
    og.close(); hg.close()
