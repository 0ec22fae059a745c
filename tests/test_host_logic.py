"""CPU: host-side logic of the package (random-policy stream, spaces, sharding, seeds)."""
import numpy as np
import pytest

from megaverse_amd import distributed, spaces
from megaverse_amd.rollout import ACTION_SPACE_SIZES, action_masks, sample_actions


def test_sample_actions_is_a_pure_function_in_range():
    a = sample_actions(1234, 17, 4096)
    b = sample_actions(1234, 17, 4096)
    assert np.array_equal(a, b) and a.dtype == np.int32 and a.shape == (4096, 6)
    for k, size in enumerate(ACTION_SPACE_SIZES):
        assert a[:, k].min() == 0 and a[:, k].max() == size - 1
        freq = np.bincount(a[:, k], minlength=size) / len(a)
        assert np.all(np.abs(freq - 1.0 / size) < 0.05)
    assert not np.array_equal(a, sample_actions(1234, 18, 4096))
    assert not np.array_equal(a, sample_actions(1235, 17, 4096))


def test_sample_actions_shards_compose():
    full = sample_actions(7, 3, 64)
    assert np.array_equal(full[16:48], sample_actions(7, 3, 32, agent_offset=16))


def test_action_space_and_observation_space():
    sp = spaces.Tuple([spaces.Discrete(n) for n in ACTION_SPACE_SIZES])
    s = sp.sample()
    assert len(s) == 6 and all(0 <= v < n for v, n in zip(s, ACTION_SPACE_SIZES))
    box = spaces.Box(0, 255, (3, 72, 128), dtype=np.uint8)
    assert tuple(box.shape) == (3, 72, 128)


def test_shard_range_partitions():
    for total, world in [(1024, 8), (1000, 8), (7, 3), (5, 8)]:
        seen = []
        for r in range(world):
            off, cnt = distributed.shard_range(r, world, total)
            seen.extend(range(off, off + cnt))
        assert seen == list(range(total))


def test_env_seeds_host_restatement_matches_oracle_and_reference():
    import oracle_lib
    n = 700     # > 624: crosses a twist
    lo, hi = np.zeros(n, np.int32), np.full(n, 1 << 30, np.int32)
    want = np.empty(n, np.int32)
    oracle_lib.lib().mvo_rand_range_seq(42, lo.ctypes.data, hi.ctypes.data, n, want.ctypes.data)
    assert np.array_equal(distributed.env_seeds(42, n), want)
    ref = oracle_lib.ref_lib()
    if ref is not None:
        r = np.empty(n, np.int32)
        ref.mvref_env_seeds(42, n, r.ctypes.data)
        assert np.array_equal(r, want)


def test_rl_wrapper_bookkeeping_on_a_stub_env():
    """megaverse_amd/rl.py:Wrapper (reference: megaverse_rl/megaverse_utils.py:30-93) without a device: 5-tuple step, episode
    statistics on done, team-spirit annealing through the reward-shaping interface"""
    from megaverse_amd.rl import Wrapper

    class StubEnv:
        num_agents, num_agents_per_env, is_multiagent, scenario_name = 4, 2, True, "Collect"
        action_space = observation_space = None

        def __init__(self):
            self.shaping = [{"teamSpirit": 0.0, "collectAll": 5.0} for _ in range(4)]
            self.t = 0

        def reset(self):
            return ["obs"] * 4

        def step(self, actions):
            self.t += 1
            done = self.t % 3 == 0
            infos = [dict(true_reward=1.0) if done else {} for _ in range(4)]
            return ["obs"] * 4, [0.5, -1.0, 0.0, 2.0], [done] * 4, infos

        def get_default_reward_shaping(self): return dict(self.shaping[0])
        def get_current_reward_shaping(self, i): return dict(self.shaping[i])
        def set_reward_shaping(self, rs, i): self.shaping[i] = dict(rs)
        def close(self): pass

    env = Wrapper(StubEnv(), increase_team_spirit=True, max_team_spirit_steps=200.0)
    obs, info = env.reset()
    assert len(obs) == 4 and info == {}
    env.set_training_info({"approx_total_training_steps": 50})
    for t in range(1, 7):
        obs, rew, term, trunc, infos = env.step(None)
        assert trunc == [False] * 4 and len(rew) == 4
        if t % 3 == 0:
            for i, inf in enumerate(infos):
                st = inf["episode_extra_stats"]
                assert inf["true_objective"] == 1.0 and st["z_collect_true_objective"] == 1.0
                assert st["z_collect_reward"] == 3 * [0.5, -1.0, 0.0, 2.0][i]          # the sum over the episode, then reset to 0
                assert st["z_approx_total_training_steps"] == 50 and st["teamSpirit"] == 0.25
                assert env.get_current_reward_shaping(i) == {"teamSpirit": 0.25, "collectAll": 5.0}
        else:
            assert infos == [{}] * 4
    assert list(env.episode_rewards) == [0, 0, 0, 0]
    env.step(None)
    env.episode_rewards[1] = 7.0                      # the reference's list is mutable (megaverse_utils.py:41): writes go through
    assert list(env.episode_rewards) == [0.5, 7.0, 0.0, 2.0]
    env.episode_rewards = [0, 0, 0, 0]
    assert list(env.episode_rewards) == [0, 0, 0, 0]
    env.close()
