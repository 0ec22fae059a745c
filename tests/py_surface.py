"""The scripted rollouts behind tests/golden/py_surface_*: shared by the generator (tests/golden/make_py_surface_golden.py, which drives the
REFERENCE's MegaverseEnv + Wrapper classes in the build container) and by the tests that replay them through megaverse_amd."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# One scripted rollout per case.  `shaping_at`: {step: (actor_idx, {key: value})} applied through Wrapper.set_reward_shaping BEFORE that step;
# `training_steps_per_step`: what the learner's side would write into training_info["approx_total_training_steps"] (step * this).
CASES = {
    # `policy`: the generator's purposeful controller (make_py_surface_golden.py: POLICIES), mixed with the random script with probability `eps`; the actions it
    # chose are recorded in the fixture (npz "actions") and replayed blind.  `min_nonzero_rewards` / `reward_windows`: what the generator asserts about the
    # recorded rewards.
    "tower_a2": dict(scenario="TowerBuilding", num_envs=3, agents=2, seed=42, steps=300, params={"episodeLengthSec": -215.0},
                     increase_team_spirit=True, max_team_spirit_steps=1000, training_steps_per_step=5,
                     shaping_at={70: (1, {"towerPickedUpObject": 0.25}), 150: (4, {"teamSpirit": 0.5, "towerBuildingReward": 2.0})},
                     policy="tower", eps=0.25, min_nonzero_rewards=30, reward_windows=[[0, 70], [70, 150], [150, 300]]),
    "rearrange_a2": dict(scenario="Rearrange", num_envs=2, agents=2, seed=7, steps=300, params={"episodeLengthSec": 5.0},
                              increase_team_spirit=True, max_team_spirit_steps=600, training_steps_per_step=3,
                              shaping_at={90: (2, {"teamSpirit": 0.9})}, policy="rearrange", eps=0.15, min_nonzero_rewards=5, reward_windows=[[0, 90], [90, 300]]),
    "collect_a1": dict(scenario="Collect", num_envs=4, agents=1, seed=11, steps=300, params={"episodeLengthSec": -60.0},
                       increase_team_spirit=False, max_team_spirit_steps=1e9, training_steps_per_step=1, shaping_at={}, policy="collect", eps=0.2, min_nonzero_rewards=10),
    "multitask_megaverse8_task15": dict(scenario="multitask_megaverse8", task_idx=15, num_envs=2, agents=2, seed=5, steps=160, params={"episodeLengthSec": 4.0},
                                       increase_team_spirit=False, max_team_spirit_steps=1e9, training_steps_per_step=1, shaping_at={},
                                       policy="rearrange", eps=0.15, min_nonzero_rewards=2),
}


def scripted_actions(case_seed, step, num_agents):
    """the action script of a case: i.i.d. per head, from numpy's Philox keyed by (seed, step) -- the replay uses the same function"""
    g = np.random.Generator(np.random.Philox(key=[case_seed, step]))
    sizes = np.array([3, 3, 3, 2, 2, 3])
    return [[int(v) for v in (g.integers(0, 1 << 30, size=6) % sizes)] for _ in range(num_agents)]


def obs_digest(obs_list):
    """sha256 over the per-agent (3, H, W) uint8 frames in order"""
    h = hashlib.sha256()
    for o in obs_list:
        assert o.dtype == np.uint8 and o.ndim == 3 and o.shape[0] == 3
        h.update(np.ascontiguousarray(o).tobytes())
    return h.hexdigest()



def load(name):
    with open(os.path.join(GOLDEN, f"py_surface_{name}.json")) as f:
        rec = json.load(f)
    return rec, np.load(os.path.join(GOLDEN, f"py_surface_{name}.npz"))


def replay(w, rec, data, shaping_calls, check_obs):
    """Drive `w` (a megaverse_amd.rl.Wrapper over some env) through the case's script and compare, field by field, with what the REFERENCE's
    Wrapper over the reference's MegaverseEnv returned.  `shaping_calls`: the list the env's gym-level set_reward_shaping spy appends
    (env_idx, agent_idx, dict) to; `check_obs`: compare frames (needs a simulator underneath, exact pixel mode)."""
    spec, n = rec["spec"], rec["num_agents"]
    shaping_at = {int(k): v for k, v in rec["shaping_at"].items()}
    assert w.num_agents == n and w.is_multiagent == rec["is_multiagent"]
    assert w.get_default_reward_shaping() == rec["default_shaping"]
    w.seed(spec["seed"]) if hasattr(w, "seed") else w.env.seed(spec["seed"])
    obs, info = w.reset()
    assert info == {} and len(obs) == n
    if check_obs:
        assert list(obs[0].shape) == rec["obs_shape"] and obs[0].dtype == np.uint8
        assert np.array_equal(np.stack(obs), data["frames_0"]), "first frames after reset"
        assert obs_digest(obs) == rec["reset_obs"]
    for st in range(spec["steps"]):
        if st in shaping_at:
            actor, upd = shaping_at[st]
            cur = w.get_current_reward_shaping(actor)
            cur.update(upd)
            w.set_reward_shaping(cur, actor)
        w.set_training_info({"approx_total_training_steps": st * spec["training_steps_per_step"]})
        # the recorded actions where the fixture holds them (a purposeful script the generator read off the oracle's state), else the random script
        acts = [[int(v) for v in row] for row in data["actions"][st]] if "actions" in data.files else scripted_actions(spec["seed"], st, n)
        obs, rewards, terminated, truncated, infos = w.step(acts)
        want = rec["steps"][st]
        assert len(obs) == len(rewards) == len(terminated) == len(truncated) == len(infos) == n
        assert [float(r) for r in rewards] == data["rewards"][st].tolist(), f"rewards, step {st}"
        assert [bool(d) for d in terminated] == data["dones"][st].tolist(), f"dones, step {st}"
        assert not any(truncated)
        assert [dict(i) for i in infos] == want["infos"], f"infos, step {st}: {infos} != {want['infos']}"
        assert [float(v) for v in w.episode_rewards] == data["episode_rewards"][st].tolist(), f"running returns, step {st}"
        if want["shaping_after"] is not None:
            assert [w.get_current_reward_shaping(i) for i in range(n)] == want["shaping_after"], f"shaping, step {st}"
        if check_obs:
            if f"frames_{st + 1}" in data.files:
                assert np.array_equal(np.stack(obs), data[f"frames_{st + 1}"]), f"frames, step {st}"
            assert obs_digest(obs) == want["obs"], f"observation digest, step {st}"
    assert [list(c) for c in shaping_calls] == [c[1:] for c in rec["gym_calls"]], "the gym-level set_reward_shaping calls, in order"
