"""GPU: megaverse_amd.MegaverseEnv + megaverse_amd.rl.Wrapper on the HIP simulator, replaying the scripted rollouts that THE REFERENCE'S OWN
`MegaverseEnv` (megaverse/megaverse_env.py:42-201) and `Wrapper` (megaverse_rl/megaverse_utils.py:30-93) produced over the oracle gym in the
build container (tests/golden/make_py_surface_golden.py -> tests/golden/py_surface_*).  Field by field: frames (exact pixel mode: the CHW,
un-flipped, alpha-dropped bytes), rewards, dones, infos incl. `true_reward` / `true_objective` / `episode_extra_stats`, running returns, the
shaping dicts after every done and every scripted change, and the sequence of gym-level set_reward_shaping calls."""
import os

import pytest

import py_surface
from megaverse_amd import MegaverseEnv
from megaverse_amd.megaverse_env import make_env_multitask
from megaverse_amd.rl import Wrapper

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def spy_on_shaping(env, calls):
    gym_set = env.env.set_reward_shaping

    def set_reward_shaping(env_idx, agent_idx, rs):
        calls.append((env_idx, agent_idx, {k: float(v) for k, v in rs.items()}))
        return gym_set(env_idx, agent_idx, rs)
    env.env.set_reward_shaping = set_reward_shaping


@pytest.mark.parametrize("name", sorted(py_surface.CASES))
def test_env_and_wrapper_equal_the_reference_classes(hip, name, monkeypatch):
    monkeypatch.setenv("BOXOBAN_LEVELS", os.path.join(ROOT, "tests", "golden", "boxoban"))
    rec, data = py_surface.load(name)
    spec = rec["spec"]
    if "multitask" in spec["scenario"]:
        env = make_env_multitask(spec["scenario"].casefold(), spec["task_idx"], spec["num_envs"], spec["agents"], 1, False, spec["params"])
    else:
        env = MegaverseEnv(spec["scenario"], spec["num_envs"], spec["agents"], 1, False, spec["params"])
    assert env.env.pixel_mode() == "exact"
    assert env.scenario_name == rec["scenario_name"] and [env.img_w, env.img_h, env.channels] == rec["img"]
    assert [s.n for s in env.action_space.spaces] == rec["action_space_sizes"]
    assert list(env.observation_space.shape) == rec["observation_space"][0] and str(env.observation_space.dtype) == rec["observation_space"][1]
    calls = []
    spy_on_shaping(env, calls)
    w = Wrapper(env, spec["increase_team_spirit"], spec["max_team_spirit_steps"])
    py_surface.replay(w, rec, data, calls, check_obs=True)
    w.close()
