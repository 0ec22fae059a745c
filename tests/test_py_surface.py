"""The learner shim against THE REFERENCE'S OWN Wrapper, without a device.

tests/golden/py_surface_*.{json,npz} were written by tests/golden/make_py_surface_golden.py, which imports
/root/reference/megaverse/megaverse_env.py and /root/reference/megaverse_rl/megaverse_utils.py from where they lie and drives
Wrapper(MegaverseEnv(...)) over the oracle gym.  Here megaverse_amd.rl.Wrapper gets the ENV-level half of those records (rewards, dones,
`true_reward` infos, shaping storage with the gym's float32 rounding) from a replaying env and must return the WRAPPER-level half: the
5-tuple, `true_objective`, `episode_extra_stats`, the running returns, team-spirit annealing and the exact sequence of gym-level
set_reward_shaping calls (megaverse_utils.py:30-93).  tests/test_py_surface_gpu.py replays the same scripts through the HIP simulator."""
import os
import subprocess
import sys

import numpy as np
import pytest

import py_surface
from megaverse_amd.rl import Wrapper


class ReplayingEnv:
    """the env-level outputs of the recorded run, in MegaverseEnv's shape (megaverse_env.py:132-201)"""
    is_multiagent = True
    action_space = observation_space = None

    def __init__(self, rec, data, calls):
        self.rec, self.data, self.calls = rec, data, calls
        self.scenario_name = rec["scenario_name"]
        self.num_agents = rec["num_agents"]
        self.num_agents_per_env = rec["spec"]["agents"]
        self.shaping = [dict(rec["default_shaping"]) for _ in range(self.num_agents)]
        self.t = 0

    def seed(self, seed): assert seed == self.rec["spec"]["seed"]
    def reset(self): return [None] * self.num_agents

    def step(self, actions):
        want = [[int(v) for v in row] for row in self.data["actions"][self.t]] if "actions" in self.data.files else py_surface.scripted_actions(self.rec["spec"]["seed"], self.t, self.num_agents)
        assert actions == want   # (the recorded script reaches the env unchanged)
        dones = self.data["dones"][self.t].tolist()
        infos = [{"true_reward": i["true_reward"]} if d else {} for d, i in zip(dones, self.rec["steps"][self.t]["infos"])]
        rewards = self.data["rewards"][self.t].tolist()
        self.t += 1
        return [None] * self.num_agents, rewards, dones, infos

    def get_default_reward_shaping(self): return dict(self.rec["default_shaping"])
    def get_current_reward_shaping(self, i): return dict(self.shaping[i])

    def set_reward_shaping(self, rs, i):
        self.calls.append((i // self.num_agents_per_env, i % self.num_agents_per_env, {k: float(v) for k, v in rs.items()}))
        self.shaping[i].update({k: float(np.float32(v)) for k, v in rs.items()})     # the gym keeps floats (scenario.hpp:259-298)

    def close(self): pass


@pytest.mark.parametrize("name", sorted(py_surface.CASES))
def test_wrapper_equals_the_reference_wrapper(name):
    rec, data = py_surface.load(name)
    calls = []
    spec = rec["spec"]
    w = Wrapper(ReplayingEnv(rec, data, calls), spec["increase_team_spirit"], spec["max_team_spirit_steps"])
    py_surface.replay(w, rec, data, calls, check_obs=False)
    assert rec["episodes_finished"] >= 2 * spec["num_envs"]
    # the rewards the Wrapper's statistics were checked on are not all zeros: the purposeful scripts earn them on both sides of the
    # shaping changes
    assert int((data["rewards"] != 0).sum()) >= spec.get("min_nonzero_rewards", 0)
    for a, b in spec.get("reward_windows", []):
        assert (data["rewards"][a:b] != 0).any(), (name, a, b)


def test_fixture_specs_are_the_committed_scripts():
    """the fixtures on disk were generated from tests/py_surface.py:CASES as it stands"""
    for name, case in py_surface.CASES.items():
        rec, data = py_surface.load(name)
        assert rec["spec"] == {k: v for k, v in case.items() if k != "shaping_at"}
        assert {int(k): (v[0], v[1]) for k, v in rec["shaping_at"].items()} == {k: (v[0], v[1]) for k, v in case["shaping_at"].items()}
        assert data["rewards"].shape == (case["steps"], case["num_envs"] * case["agents"])
        assert rec["img"] == [128, 72, 3] and rec["obs_shape"] == [3, 72, 128] and rec["action_space_sizes"] == [3, 3, 3, 2, 2, 3]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference tree (build container only)")
def test_the_generator_reproduces_the_committed_fixtures(tmp_path):
    """run the reference's classes again (one case) and compare with what is committed: the fixtures are what the reference returns today"""
    gen = os.path.join(py_surface.GOLDEN, "make_py_surface_golden.py")
    subprocess.check_call([sys.executable, gen, "--out", str(tmp_path), "--case", "tower_a2"], stdout=subprocess.DEVNULL)
    for ext in ("json",):
        assert open(tmp_path / f"py_surface_tower_a2.{ext}").read() == open(os.path.join(py_surface.GOLDEN, f"py_surface_tower_a2.{ext}")).read()
    a, b = np.load(tmp_path / "py_surface_tower_a2.npz"), np.load(os.path.join(py_surface.GOLDEN, "py_surface_tower_a2.npz"))
    assert sorted(a.files) == sorted(b.files) and all(np.array_equal(a[k], b[k]) for k in a.files)
