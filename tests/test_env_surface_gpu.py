"""GPU: the MegaverseEnv / MegaverseGym surface, mirroring the reference's own Python tests
(megaverse/tests/test_env.py) test for test, on TowerBuilding."""
import copy

import numpy as np
import pytest

from megaverse_amd import MegaverseEnv

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def sample_actions(e):
    return [e.action_space.sample() for _ in range(e.num_agents)]


def make_test_env(num_envs, num_agents_per_env, num_simulation_threads, use_vulkan=False, params=None):
    return MegaverseEnv('TowerBuilding', num_envs, num_agents_per_env, num_simulation_threads, use_vulkan, params)


def test_env(hip):                      # test_env.py:22-26
    e = make_test_env(1, 1, 1)
    o = e.reset()
    assert len(o) == 1 and o[0].shape == (3, 72, 128) and o[0].dtype == np.uint8
    obs, rew, dones, infos = e.step(sample_actions(e))
    assert len(obs) == len(rew) == len(dones) == len(infos) == 1
    e.close()


def test_env_close_immediately(hip):    # test_env.py:28-30
    e = make_test_env(1, 1, 1)
    e.close()
    e.close()                           # idempotent


def test_two_envs_same_process(hip):    # test_env.py:32-40
    e1, e2 = make_test_env(1, 1, 1), make_test_env(1, 1, 1)
    e1.reset(); e2.reset()
    e1.close(); e2.close()


def test_seeds(hip):                    # test_env.py:42-55
    e1 = make_test_env(1, 1, 1); e1.seed(42)
    e2 = make_test_env(1, 1, 1); e2.seed(42)
    obs1, obs2 = e1.reset(), e2.reset()
    assert np.array_equal(obs1, obs2)
    # unlike the reference ("after this we have randomness due to physics?") the whole rollout is deterministic
    for st in range(50):
        a = [tuple(int(v) for v in row) for row in np.random.default_rng(st).integers(0, 2, (1, 6))]
        o1, r1, d1, _ = e1.step(a)
        o2, r2, d2, _ = e2.step(a)
        assert np.array_equal(o1, o2) and r1 == r2 and d1 == d2
    e2.close(); e1.close()


@pytest.mark.parametrize("episode_length_sec", [60.0, -300.0])
def test_render_and_auto_reset(hip, episode_length_sec):    # test_env.py:57-88 (rendering, 1-second episodes)
    params = {'episodeLengthSec': episode_length_sec}
    e1 = make_test_env(2, 2, 2, params=params)
    e2 = make_test_env(1, 1, 1, params=params)
    e1.reset(); e2.reset()
    img = e1.render(mode='rgb_array')
    assert img.shape == (2 * 432, 2 * 768, 3) and img.dtype == np.uint8      # hires 768x432 tiles (megaverse.cpp:261)
    seen_done = False
    for i in range(30):
        obs, rew, dones, infos = e1.step(sample_actions(e1))
        assert len(obs) == 4 and len(dones) == 4 and dones[0] == dones[1] and dones[2] == dones[3]
        for d, info in zip(dones, infos):
            assert ('true_reward' in info) == bool(d)
        seen_done |= any(dones)
        e2.step(sample_actions(e2))
        if i % 10 == 0:
            e1.render(mode='rgb_array'); e2.render(mode='rgb_array')
    assert seen_done == (episode_length_sec < 0)
    e2.close(); e1.close()


def test_reward_shaping(hip):           # test_env.py:123-140
    e = MegaverseEnv('TowerBuilding', num_envs=3, num_agents_per_env=2, num_simulation_threads=2, use_vulkan=True)
    default_reward_shaping = e.get_default_reward_shaping()
    assert default_reward_shaping == pytest.approx({'teamSpirit': 0.1, 'towerPickedUpObject': 0.1,
                                                    'towerVisitedBuildingZoneWithObject': 0.1, 'towerBuildingReward': 1.0})
    for idx in (0, 1, 2, 5):
        assert default_reward_shaping == e.get_current_reward_shaping(idx)
    new_reward_shaping = copy.deepcopy(default_reward_shaping)
    for k, v in new_reward_shaping.items():
        new_reward_shaping[k] = v * 3
    e.set_reward_shaping(new_reward_shaping, 3)
    assert default_reward_shaping == e.get_current_reward_shaping(0)
    assert default_reward_shaping == e.get_current_reward_shaping(1)
    assert default_reward_shaping != e.get_current_reward_shaping(3)
    e.reset()
    assert e.get_current_reward_shaping(3) == pytest.approx(new_reward_shaping)   # survives episode boundaries
    e.close()


def test_params_must_be_float(hip):     # megaverse_env.py:62-68
    with pytest.raises(Exception):
        MegaverseEnv('TowerBuilding', 1, 1, 1, params={'episodeLengthSec': 1})


def test_batched_device_observations(hip):
    import torch
    e = MegaverseEnv('TowerBuilding', 8, 2, 1, img_w=128, img_h=128)
    e.seed(1)
    ref_list = e.reset()
    t = e.observations_tensor()
    assert t.is_cuda and tuple(t.shape) == (16, 3, 128, 128) and t.dtype == torch.uint8
    assert np.array_equal(t.cpu().numpy(), np.stack(ref_list))
    acts = torch.randint(0, 2, (16, 6), dtype=torch.int32, device='cuda')
    obs, rew, dones = e.step_batched(acts)
    assert obs.data_ptr() == t.data_ptr() and rew.shape == (16,) and dones.shape == (8,)
    one = e.env.get_observation(3, 1)                       # per-frame copy == slice of the slab
    assert np.array_equal(one[:, :, :3].transpose(2, 0, 1), obs[7].cpu().numpy())
    e.close()


def test_step_device_equals_step_batched(hip):
    """step_device: the same ticks as step_batched, its three outputs device tensors, no host synchronisation between steps; a policy on the device in the loop"""
    import torch
    spaces_t = torch.tensor([3, 3, 3, 2, 2, 3], dtype=torch.int32, device='cuda')

    def rollout(device_outputs):
        e = MegaverseEnv('TowerBuilding', 12, 2, 1, img_w=64, img_h=64, params={'episodeLengthSec': -300.0})
        e.seed(5)
        e.reset()
        obs = e.observations_tensor()
        rewards, dones, frames = [], [], []
        for st in range(70):   # (negative episode length, as in the reference's test_env.py:57-88: the envs reset inside the rollout)
            acts = ((obs.reshape(24, -1)[:, 37:37 + 6 * 97:97].to(torch.int32) + st) % spaces_t).contiguous()
            if device_outputs:
                obs, r, d = e.step_device(acts)
                rewards.append(r.clone()); dones.append(d.clone())
            else:
                obs, r, d = e.step_batched(acts)
                rewards.append(torch.from_numpy(r)); dones.append(torch.from_numpy(d.astype(np.uint8)))
            if st % 23 == 0:
                frames.append(obs.clone())
        torch.cuda.synchronize()
        out = (torch.stack([r.cpu() for r in rewards]).numpy(), torch.stack([d.cpu() for d in dones]).numpy(), torch.stack(frames).cpu().numpy(),
               e.env.get_rewards_array().copy(), e.env.get_dones().copy())
        e.close()
        return out

    a, b = rollout(True), rollout(False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])   # (the host getters read the ring's entry)
    assert b[1].sum() > 0                                                # an episode ended on the way


def test_rl_wrapper_bookkeeping(hip):
    """megaverse_rl/megaverse_utils.py:30-93: 5-tuple step, per-episode extra stats, team-spirit annealing"""
    from types import SimpleNamespace
    from megaverse_amd.rl import MEGAVERSE_ENVS, make_megaverse
    assert [s.name for s in MEGAVERSE_ENVS][:4] == ["TowerBuilding", "ObstaclesEasy", "ObstaclesHard", "Collect"]
    cfg = SimpleNamespace(megaverse_num_envs_per_instance=3, megaverse_num_agents_per_env=2, megaverse_num_simulation_threads=1,
                          megaverse_use_vulkan=False, megaverse_increase_team_spirit=True, megaverse_max_team_spirit_steps=1000.0)
    env = make_megaverse("TowerBuilding", cfg, img_w=32, img_h=32, params={"episodeLengthSec": -220.0})   # episodes of a few seconds
    assert env.num_agents == 6 and env.is_multiagent
    env.seed(3)
    obs, info = env.reset()
    assert len(obs) == 6 and obs[0].shape == (3, 32, 32) and info == {}
    env.set_training_info({"approx_total_training_steps": 250})
    from megaverse_amd.rollout import sample_actions
    seen_done = 0
    for st in range(400):
        obs, rewards, terminated, truncated, infos = env.step(sample_actions(7, st, 6))
        assert len(rewards) == 6 and len(terminated) == 6 and truncated == [False] * 6
        for i, inf in enumerate(infos):
            if terminated[i]:
                seen_done += 1
                st_ = inf["episode_extra_stats"]
                assert inf["true_objective"] == inf["true_reward"] == st_["z_towerbuilding_true_objective"]
                assert "z_towerbuilding_reward" in st_ and st_["z_approx_total_training_steps"] == 250
                assert st_["teamSpirit"] == 0.25 and env.get_current_reward_shaping(i)["teamSpirit"] == 0.25
            else:
                assert inf == {}
    assert seen_done >= 6
    obs_t, rewards, term, trunc, infos = env.step_batched(sample_actions(7, 999, 6))
    assert tuple(obs_t.shape) == (6, 3, 32, 32) and obs_t.is_cuda and rewards.shape == (6,) and term.shape == (6,)
    env.close()


def test_pybind_flavour_equals_ctypes_flavour(hip):
    """megaverse_amd/pybind (the reference's pybind table on the C ABI) and the ctypes binding drive the same library"""
    from megaverse_amd import build
    build.build_pybind()
    from megaverse_amd.pybind import megaverse as m
    from megaverse_amd.extension import MegaverseGym
    from megaverse_amd.rollout import sample_actions
    N, A = 3, 2
    a = m.MegaverseGym("ObstaclesEasy", 48, 32, N, A, 1, False, {"episodeLengthSec": 70.0})
    b = MegaverseGym("ObstaclesEasy", 48, 32, N, A, 1, False, {"episodeLengthSec": 70.0})
    assert a.num_agents() == b.num_agents() == A and a.action_space_sizes() == [3, 3, 3, 2, 2, 3]
    a.seed(5); b.seed(5); a.reset(); b.reset()
    assert a.get_reward_shaping(0, 1) == b.get_reward_shaping(0, 1)
    for st in range(25):
        acts = sample_actions(3, st, N * A)
        for e in range(N):
            for k in range(A):
                a.set_actions(e, k, [int(v) for v in acts[e * A + k]])
        b.set_actions_batched(acts)
        a.step(); b.step()
        assert np.allclose(a.get_last_rewards(), b.get_last_rewards(), rtol=0, atol=0)
        assert [a.is_done(e) for e in range(N)] == [b.is_done(e) for e in range(N)]
    for e in range(N):
        for k in range(A):
            fa, fb = a.get_observation(e, k), b.get_observation(e, k)
            assert fa.shape == (32, 48, 4) and np.array_equal(fa, fb)
            assert a.true_objective(e, k) == b.true_objective(e, k)
    a.draw_hires()
    assert a.get_hires_observation(0, 0).shape == (432, 768, 4)
    a.close(); b.close()


def test_pybind_flavour_places_its_envs_from_the_environment(hip, monkeypatch):
    """The reference's constructor has no shard argument; through the reference-named module the process environment places the gym in the
    job: MV_ENV_OFFSET / MV_TOTAL_ENVS / MV_ENV_STRIDE (a round-robin multi-task job: every k-th env from an offset) and MV_DEVICE /
    LOCAL_RANK (the latter modulo the visible device count: LOCAL_RANK=5 on a one-GPU box is device 0).  Same envs, same observations as
    the ctypes binding with the explicit arguments."""
    from megaverse_amd import build
    build.build_pybind()
    from megaverse_amd.pybind import megaverse as m
    from megaverse_amd.extension import MegaverseGym
    from megaverse_amd.rollout import sample_actions
    N, A, total, stride, offset = 4, 1, 24, 3, 2
    monkeypatch.setenv("MV_ENV_OFFSET", str(offset)); monkeypatch.setenv("MV_TOTAL_ENVS", str(total)); monkeypatch.setenv("MV_ENV_STRIDE", str(stride))
    monkeypatch.delenv("MV_DEVICE", raising=False); monkeypatch.setenv("LOCAL_RANK", "5")
    a = m.MegaverseGym("TowerBuilding", 40, 24, N, A, 1, False, {})
    for k in ("MV_ENV_OFFSET", "MV_TOTAL_ENVS", "MV_ENV_STRIDE", "LOCAL_RANK"):
        monkeypatch.delenv(k)
    b = MegaverseGym("TowerBuilding", 40, 24, N, A, 1, False, {}, env_offset=offset, total_envs=total, env_stride=stride)
    whole = MegaverseGym("TowerBuilding", 40, 24, total, A, 1, False, {})
    for g in (a, b, whole):
        g.seed(9); g.reset()
    for j in range(N):   # local env j is job-wide env offset + j * stride
        assert np.array_equal(a.get_observation(j, 0), whole.get_observation(offset + j * stride, 0))
        assert np.array_equal(a.get_observation(j, 0), b.get_observation(j, 0))
    for st in range(10):
        acts = sample_actions(3, st, N * A)
        for e in range(N):
            a.set_actions(e, 0, [int(v) for v in acts[e]])
        b.set_actions_batched(acts)
        a.step(); b.step()
    for j in range(N):
        assert np.array_equal(a.get_observation(j, 0), b.get_observation(j, 0))
    a.close(); b.close(); whole.close()


def test_call_size_advice_and_arena(hip, monkeypatch):
    """mv_recommended_ticks_per_call / mv_recommended_pass_overlap / mv_arena_bytes (include/megaverse_hip.h): the measured rules bench.py used to hold, behind the
    ABI -- 16 ticks per call for 1024 .. 2047 frames where the slot groups hold them (sized by footprint; MV_PIPE_BATCH overrides), 8 otherwise, 1 for few-tick
    episodes; overlapped passes wherever the passes bound a call (not Empty, not small or four-agent TowerBuilding gyms, not few-tick episodes); a step_n of more ticks than the slot groups hold is split by the library"""
    import os
    import torch
    from megaverse_amd.extension import MegaverseGym
    monkeypatch.setenv("BOXOBAN_LEVELS", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "boxoban"))
    cases = [("TowerBuilding", 1024, 1, {}, 16, True), ("TowerBuilding", 64, 1, {}, 8, False), ("TowerBuilding", 512, 4, {}, 8, False),
             ("TowerBuilding", 512, 2, {}, 16, True), ("ObstaclesHard", 1024, 1, {}, 16, True), ("Sokoban", 1024, 1, {}, 8, True),
             ("HexMemory", 1024, 1, {}, 8, True), ("Empty", 1024, 1, {}, 16, False), ("Rearrange", 64, 1, {"episodeLengthSec": 0.5}, 1, False)]
    for scenario, n, a, params, want, overlap in cases:
        g = MegaverseGym(scenario, 32, 32, n, a, 2, False, params)
        assert g.recommended_ticks_per_call() == want, (scenario, n, a, g.recommended_ticks_per_call())
        assert g.recommended_pass_overlap() == overlap, scenario
        assert g.arena_bytes() > n * a * 32 * 32 * 4
        g.close()
    monkeypatch.setenv("MV_PIPE_BATCH", "8")
    g = MegaverseGym("TowerBuilding", 32, 32, 1024, 1, 2, False, {})
    assert g.recommended_ticks_per_call() == 8
    small = g.arena_bytes()
    ring = torch.zeros((24, 1024, 32, 32, 4), dtype=torch.uint8, device="cuda:0")
    g.set_output_ring(24, ring.data_ptr())
    g.seed(3); g.reset()
    g.step_n(20, "multidiscrete", 5, 0)   # 8 + 8 + 4: split by the library
    g.synchronize(); torch.cuda.synchronize()
    assert int(ring[:20, :, :, :, 3].min()) == 255 and int(ring[20:].max()) == 0
    g.close()
    monkeypatch.setenv("MV_PIPE_BATCH", "16")
    g = MegaverseGym("TowerBuilding", 32, 32, 1024, 1, 2, False, {})
    assert g.arena_bytes() > small and g.recommended_ticks_per_call() == 16
    g.close()
