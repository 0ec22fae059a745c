"""GPU parity for Collect: host-generated Perlin landscapes + HIP step (per-agent broadphase into an LDS
candidate list) + raster with up to ~1300 primitives per frame, against the CPU oracle: bit-exact state / rewards / dones /
pixels."""
import numpy as np
import pytest

from hip_util import diff_snapshots, hip_snapshot, make_pair, set_same_actions

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


def frames(g, N, A):
    return np.stack([g.get_observation(e, a) for e in range(N) for a in range(A)])


@pytest.mark.parametrize("A,seed", [(1, 3), (2, 14), (5, 15), (8, 92)])
def test_reset_parity(hip, A, seed):
    N = 32
    og, hg = make_pair(N, A, 32, 32, seed=seed, scenario="Collect")
    for e in range(N):
        d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
        assert not d, (e, d[:5])
    og.close(); hg.close()


@pytest.mark.parametrize("W,H", [(128, 72), (128, 128), (40, 24)])
def test_pixels_after_reset(hip, W, H):
    N, A = 24, 2
    og, hg = make_pair(N, A, W, H, seed=65, scenario="Collect")
    fo, fh = frames(og, N, A), frames(hg, N, A)
    bad = [i for i in range(N * A) if not np.array_equal(fo[i], fh[i])]
    assert not bad, (bad, int((fo != fh).sum()))
    og.close(); hg.close()


@pytest.mark.parametrize("A,seed", [(1, 1), (2, 2), (4, 3), (8, 4)])
def test_rollout_parity(hip, A, seed):
    """state, rewards, dones every step; diamonds, falls off the rim and auto-resets all occur within 1400 steps"""
    N = 10
    og, hg = make_pair(N, A, 32, 32, seed=seed, scenario="Collect")
    resets, pos, neg = 0, 0, 0
    for st in range(1400):
        set_same_actions(og, hg, N, A, 300 + seed, st)
        og.step_norender(); hg.step_no_render()
        ro, rh = og.get_last_rewards(), hg.get_rewards_array()
        assert ro.tobytes() == rh.tobytes(), (st, ro, rh)
        pos += int((ro > 0).sum()); neg += int((ro < 0).sum())
        do = np.array([og.is_done(e) for e in range(N)]); dh = hg.get_dones()
        assert np.array_equal(do, dh.astype(bool)), (st, do, dh)
        resets += int(do.sum())
        if st % 40 == 0 or do.any():
            for e in range(N):
                d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
                assert not d, (st, e, d[:5])
        to = np.array([og.true_objective(e, a) for e in range(N) for a in range(A)], np.float32)
        assert to.tobytes() == hg.get_true_objectives().tobytes()
    og.render(); hg.render()
    assert np.array_equal(frames(og, N, A), frames(hg, N, A))
    assert resets > 0 and pos > 0 and neg > 0, (resets, pos, neg)
    og.close(); hg.close()


def test_rollout_pixels_every_20_steps(hip):
    N, A = 6, 2
    og, hg = make_pair(N, A, 64, 64, seed=8, scenario="Collect")
    for st in range(300):
        set_same_actions(og, hg, N, A, 19, st)
        if st % 20 == 19:
            og.step(); hg.step()
            fo, fh = frames(og, N, A), frames(hg, N, A)
            assert np.array_equal(fo, fh), (st, int((fo != fh).sum()))
        else:
            og.step_norender(); hg.step_no_render()
    og.close(); hg.close()


def test_reward_shaping_keys(hip):
    og, hg = make_pair(2, 2, 32, 32, seed=1, scenario="Collect")
    sh = hg.get_reward_shaping(1, 1)
    assert sh == {"teamSpirit": 0.0, "collectSingleGood": 1.0, "collectSingleBad": -1.0, "collectAll": 5.0, "collectAbyss": -0.5}
    og.close(); hg.close()


def test_reseed_mid_run_takes_effect_at_the_next_reset(hip):
    """Env::seed re-seeds the env's stream immediately (env.cpp:52-55): the episode that was generated ahead of time
    from the old stream must not be used"""
    N, A = 6, 2
    og, hg = make_pair(N, A, 32, 32, seed=21, scenario="Collect")
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
