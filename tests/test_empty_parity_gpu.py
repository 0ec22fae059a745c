"""Empty (scenarios/init.hpp:34, scenario_empty.{hpp,cpp}): the scenario the reference's own performance test and its published
simulation figures use (README.md:243-247).  One static 20 x 2 x 20 box, every agent spawned at the same cell (they push each other
apart), no components: agents that walk off the platform keep falling.  HIP path vs the oracle, bit for bit."""
import numpy as np
import pytest

from hip_util import diff_snapshots, hip_snapshot, make_pair, set_same_actions

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@pytest.mark.parametrize("A,N,steps", [(1, 16, 1100), (4, 6, 500)])
def test_empty_rollout_and_pixels(hip, A, N, steps):
    og, hg = make_pair(N, A, 64, 64, seed=13, scenario="Empty")
    assert hg.get_reward_shaping(0, 0) == {"teamSpirit": 0.0}
    for e in range(N):
        assert not diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A), e
    fell = 0
    for st in range(steps):
        set_same_actions(og, hg, N, A, 21, st)
        og.step(); hg.step()
        assert np.array_equal(og.get_last_rewards().view(np.uint32), hg.get_rewards_array().view(np.uint32))
        assert float(np.abs(hg.get_rewards_array()).max()) == 0.0
        if st % 100 == 99 or st == steps - 1:
            assert [og.is_done(e) for e in range(N)] == hg.get_dones().astype(bool).tolist()
            for e in range(N):
                so = og.snapshot(e)
                assert not diff_snapshots(so, hip_snapshot(hg, e), A), (st, e)
                fell += int(so["agents"][0]["pos"][1] < -20.0)
                for a in range(A):
                    assert np.array_equal(og.get_observation(e, a), hg.get_observation(e, a)), (st, e, a)
    if A == 1:
        assert fell > 0, "nobody walked off the platform: the no-fall-detection branch was not exercised"   # 60 s episodes: also auto-resets at tick 900
    og.close(); hg.close()
