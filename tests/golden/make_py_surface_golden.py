"""Golden vectors for the Python surface, produced by THE REFERENCE'S OWN CLASSES.  Runs in the build container only.

What runs: `/root/reference/megaverse/megaverse_env.py` (class MegaverseEnv, make_env_multitask) and
`/root/reference/megaverse_rl/megaverse_utils.py` (class Wrapper), imported from where they lie -- nothing of them is copied, and
neither file travels to the GPU box in any form; only the DATA this script writes does (tests/golden/py_surface_*.npz|json).

What is stood in for, and only because the image lacks it:
  * `gym`, `cv2`, `sample_factory.*`  -> base-class-only stubs in sys.modules (Env/Wrapper with `unwrapped`, Discrete/Tuple/Box that keep
    their arguments, the two Sample-Factory interfaces with the attributes megaverse_utils.py reads, a logger).  None of them computes
    anything that ends up in a fixture.
  * `megaverse.extension.megaverse.MegaverseGym` (the pybind class; Bullet/Magnum are absent) -> tests/oracle_lib.OracleGym, the CPU
    restatement with the same 17-method table, returning what pybind would: `get_last_rewards` a list of Python floats (std::vector<float>),
    `get_observation` an (h, w, 4) uint8 view.

So the fixtures pin the SURFACE (call order, list shapes, the done/infos protocol, CHW un-flipped frames, the Wrapper's episode statistics,
team-spirit annealing, shaping read-modify-write) to the reference's code; the simulation underneath is the oracle's (parity of physics and
pixels with Bullet/GL stays unpinned).

    python tests/golden/make_py_surface_golden.py          # writes tests/golden/py_surface_<case>.json + .npz
"""
import importlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
REFERENCE = "/root/reference"
sys.path.insert(0, TESTS)

from py_surface import CASES, obs_digest, scripted_actions  # noqa: E402  (tests/py_surface.py: the scripts, shared with the replaying tests)


def install_stubs():
    """base-class-only stand-ins for what the image lacks; see the module docstring"""
    from oracle_lib import OracleGym

    gym = types.ModuleType("gym")

    class Env:
        @property
        def unwrapped(self):
            return self

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

        @property
        def unwrapped(self):
            return self.env.unwrapped

        def close(self):                 # gym.Wrapper forwards close()
            return self.env.close()

    class _Space:
        def __init__(self, *a, **k):
            self.args, self.kwargs = a, k

    spaces = types.ModuleType("gym.spaces")
    spaces.Discrete = type("Discrete", (_Space,), {})
    spaces.Tuple = type("Tuple", (_Space,), {})
    spaces.Box = type("Box", (_Space,), {})
    gym.Env, gym.Wrapper, gym.spaces = Env, Wrapper, spaces
    sys.modules["gym"], sys.modules["gym.spaces"] = gym, spaces
    sys.modules["cv2"] = types.ModuleType("cv2")          # render() is not driven here

    sf = types.ModuleType("sample_factory")
    sf_envs = types.ModuleType("sample_factory.envs")
    sf_env_utils = types.ModuleType("sample_factory.envs.env_utils")
    sf_utils = types.ModuleType("sample_factory.utils")
    sf_utils_utils = types.ModuleType("sample_factory.utils.utils")

    class RewardShapingInterface:
        def __init__(self):
            pass

    class TrainingInfoInterface:
        def __init__(self):
            self.training_info = {}

        def set_training_info(self, training_info):
            self.training_info = training_info

    sf_env_utils.RewardShapingInterface, sf_env_utils.TrainingInfoInterface = RewardShapingInterface, TrainingInfoInterface
    import logging
    sf_utils_utils.log = logging.getLogger("sample_factory_stub")
    for name, mod in (("sample_factory", sf), ("sample_factory.envs", sf_envs), ("sample_factory.envs.env_utils", sf_env_utils),
                      ("sample_factory.utils", sf_utils), ("sample_factory.utils.utils", sf_utils_utils)):
        sys.modules[name] = mod

    class PybindShapedGym(OracleGym):
        """OracleGym returning what the pybind class returns (bindings/megaverse.cpp:128-137): a list of Python floats"""
        calls = []          # (method, args) of the calls whose ORDER is part of the surface

        def get_last_rewards(self):
            return [float(v) for v in OracleGym.get_last_rewards(self)]

        def set_reward_shaping(self, env_idx, agent_idx, rs):
            PybindShapedGym.calls.append(("set_reward_shaping", env_idx, agent_idx, {k: float(v) for k, v in rs.items()}))
            return OracleGym.set_reward_shaping(self, env_idx, agent_idx, rs)

    ext = types.ModuleType("megaverse.extension")
    ext_mv = types.ModuleType("megaverse.extension.megaverse")
    ext_mv.MegaverseGym = PybindShapedGym
    ext_mv.set_megaverse_log_level = lambda level: None
    sys.modules["megaverse.extension"], sys.modules["megaverse.extension.megaverse"] = ext, ext_mv
    return PybindShapedGym


# ---- scripted policies with a purpose: uniform random actions almost never earn a reward, so the fixtures pinned the reward PLUMBING (team-spirit
# annealing, shaping set mid-run, episode returns) on zeros.  These controllers are read off the oracle's state in THIS generator only -- walk to the nearest free box,
# pick it up, carry it where it scores, put it down; walk into the nearest diamond -- mixed with the random script (eps); the actions they chose are RECORDED in the
# fixture and the tests replay them blind (py_surface.replay: data["actions"]).  Conventions (env.cpp:89-122, agent.cpp:100-150): heads [strafe, walk, look, jump,
# interact, pitch]; forward = (m20, -m22); LookLeft turns forward clockwise in (x, z).
import math


def _steer(ag, tx, tz, act):
    px, pz = float(ag["pos"][0]), float(ag["pos"][2])
    _, _, m20, m22 = [float(v) for v in ag["basis"]]
    fx, fz = m20, -m22
    nrm = math.hypot(fx, fz) or 1.0
    fx, fz = fx / nrm, fz / nrm
    dx, dz = tx - px, tz - pz
    dist = math.hypot(dx, dz)
    ang = math.atan2(fx * dz - fz * dx, fx * dx + fz * dz)
    if abs(ang) > 0.12:
        act[2] = 1 if ang < 0 else 2
    if abs(ang) < 0.7 and dist > 1.0:
        act[1] = 1
    return dist, ang, (px + fx, pz + fz)   # ... and the cell one ahead: where a pick-up looks, where a carried box goes


def _nearest(ag, cands):
    px, pz = float(ag["pos"][0]), float(ag["pos"][2])
    return min(cands, key=lambda c: math.hypot(c[0] - px, c[1] - pz)) if cands else None


def policy_tower(s, a, st):
    ag, bz = s["agents"][a], s["bz"]
    act = [0, 0, 0, 0, 0, 0]
    if int(ag["carrying"]) < 0:
        t = _nearest(ag, [(o[0] + 0.5, o[2] + 0.5) for o in s["objects"][:int(s["num_objects"])] if o[3] == 0 and not (bz[0] <= o[0] < bz[1] and bz[2] <= o[2] < bz[3])])
        if t is None:
            return None
        dist, ang, _ = _steer(ag, t[0], t[1], act)
        if 0.6 < dist < 1.45 and abs(ang) < 0.25:
            act[4] = 1
    else:
        _, _, q = _steer(ag, (bz[0] + bz[1]) / 2.0, (bz[2] + bz[3]) / 2.0, act)
        if bz[0] <= math.floor(q[0]) < bz[1] and bz[2] <= math.floor(q[1]) < bz[3]:
            act[4] = 1
    return act


def policy_rearrange(s, a, st):
    RX, RY, RZ = 13, 2, 5   # the right pedestal's centre cell (scenario_rearrange.hpp:130-131)
    ag, n = s["agents"][a], int(s["num_items"])
    items, objs = s["items"][:n], s["objects"][:n]
    act = [0, 0, 0, 0, 0, 0]
    carrying = int(ag["carrying"])
    if carrying < 0:
        t = _nearest(ag, [(objs[k][0] + 0.5, objs[k][2] + 0.5) for k in range(n) if objs[k][3] == 0 and
                          (objs[k][0] - RX, objs[k][1] - RY, objs[k][2] - RZ) != tuple(int(v) for v in items[k][2:5])])
        if t is None:
            return None
        dist, ang, _ = _steer(ag, t[0], t[1], act)
        if 0.6 < dist < 1.5 and abs(ang) < 0.3:
            act[4] = 1
    else:
        it = items[carrying]
        _, _, q = _steer(ag, RX + int(it[2]) + 0.5, RZ + int(it[4]) + 0.5, act)
        if math.floor(q[0]) == RX + int(it[2]) and math.floor(q[1]) == RZ + int(it[4]):
            act[4] = 1
    if act[1] == 1 and math.hypot(float(ag["hv"][0]), float(ag["hv"][1])) < 0.5 and st % 3 == 0:
        act[3] = 1   # blocked: jump (a pedestal's edge is 0.5 high)
    return act


def policy_collect(s, a, st):
    ag = s["agents"][a]
    act = [0, 0, 0, 0, 0, 0]
    t = _nearest(ag, [(r[0] + 0.5, r[2] + 0.5) for r in s["rewards"][:int(s["num_rewards"])] if r[3] == 1])   # the +1 diamonds that are still there
    if t is None:
        return None
    dist, ang, _ = _steer(ag, t[0], t[1], act)
    if dist <= 1.0 and abs(ang) < 0.7:
        act[1] = 1
    if act[1] == 1 and math.hypot(float(ag["hv"][0]), float(ag["hv"][1])) < 0.5 and st % 3 == 0:
        act[3] = 1   # blocked by a terrace of the landscape: jump
    return act


POLICIES = {"tower": policy_tower, "rearrange": policy_rearrange, "collect": policy_collect}


def policy_actions(case, gym, st, n_envs, agents):
    """the case's action script for step st: the purposeful controller where it has something to do, the random script with probability eps and otherwise"""
    rnd = scripted_actions(case["seed"], st, n_envs * agents)
    if not case.get("policy"):
        return rnd
    gen = np.random.Generator(np.random.Philox(key=[case["seed"] + 1000, st]))
    out = []
    for e in range(n_envs):
        s = gym.snapshot(e)
        for a in range(agents):
            act = None if gen.random() < case.get("eps", 0.2) else POLICIES[case["policy"]](s, a, st)
            out.append(rnd[e * agents + a] if act is None else act)
    return out


def run_case(name, case, ref_env_mod, ref_utils_mod, gym_cls, out_dir):
    gym_cls.calls = []
    os.environ.setdefault("BOXOBAN_LEVELS", os.path.join(HERE, "boxoban"))
    if "multitask" in case["scenario"]:
        env = ref_env_mod.make_env_multitask(case["scenario"].casefold(), case["task_idx"], case["num_envs"], case["agents"], 1, False, case["params"])
    else:
        env = ref_env_mod.MegaverseEnv(case["scenario"], case["num_envs"], case["agents"], 1, False, case["params"])
    w = ref_utils_mod.Wrapper(env, case["increase_team_spirit"], case["max_team_spirit_steps"])
    n = env.num_agents
    rec = dict(case=name, spec={k: v for k, v in case.items() if k != "shaping_at"},
               shaping_at={str(k): [v[0], v[1]] for k, v in case["shaping_at"].items()},
               scenario_name=env.scenario_name, num_agents=w.num_agents, is_multiagent=w.is_multiagent, img=[env.img_w, env.img_h, env.channels],
               action_space_sizes=[s.args[0] for s in env.action_space.args[0]], observation_space=[list(env.observation_space.args[2]), str(np.dtype(env.observation_space.kwargs["dtype"]))],
               default_shaping=dict(w.get_default_reward_shaping()), steps=[])
    env.seed(case["seed"])
    obs, info = w.reset()
    assert info == {}
    rec["reset_obs"] = obs_digest(obs)
    rec["obs_shape"] = list(obs[0].shape)
    frames = {0: np.stack(obs)}
    rewards_all = np.zeros((case["steps"], n), np.float64)
    dones_all = np.zeros((case["steps"], n), np.bool_)
    episode_rewards_all = np.zeros((case["steps"], n), np.float64)
    actions_all = np.zeros((case["steps"], n, 6), np.int8)
    for st in range(case["steps"]):
        if st in case["shaping_at"]:
            actor, upd = case["shaping_at"][st]
            cur = w.get_current_reward_shaping(actor)
            cur.update(upd)
            w.set_reward_shaping(cur, actor)
        w.set_training_info({"approx_total_training_steps": st * case["training_steps_per_step"]})
        acts = policy_actions(case, env.env, st, case["num_envs"], case["agents"])   # (env.env: the gym under the reference's MegaverseEnv = the oracle)
        actions_all[st] = np.asarray(acts, np.int8)
        obs, rewards, terminated, truncated, infos = w.step(acts)
        assert len(obs) == len(rewards) == len(terminated) == len(truncated) == len(infos) == n
        rewards_all[st] = rewards
        dones_all[st] = terminated
        episode_rewards_all[st] = w.episode_rewards
        assert not any(truncated)
        rec["steps"].append(dict(obs=obs_digest(obs), infos=[dict(i) for i in infos],
                                 shaping_after=[w.get_current_reward_shaping(i) for i in range(n)] if (any(terminated) or st in case["shaping_at"]) else None))
        if (st + 1) % 100 == 0:
            frames[st + 1] = np.stack(obs)
    rec["gym_calls"] = [list(c) for c in gym_cls.calls]
    rec["episodes_finished"] = int(dones_all[:, ::case["agents"]].sum())
    rec["nonzero_rewards"] = int((rewards_all != 0).sum())
    assert rec["nonzero_rewards"] >= case.get("min_nonzero_rewards", 0), (name, rec["nonzero_rewards"])
    for a, b in case.get("reward_windows", []):   # rewards on both sides of the mid-run shaping changes
        assert (rewards_all[a:b] != 0).any(), (name, "no reward in steps", a, b)
    assert rec["episodes_finished"] >= 2 * case["num_envs"], (name, rec["episodes_finished"])
    w.close()
    with open(os.path.join(out_dir, f"py_surface_{name}.json"), "w") as f:
        json.dump(rec, f, indent=0, sort_keys=True)
    np.savez_compressed(os.path.join(out_dir, f"py_surface_{name}.npz"), rewards=rewards_all, dones=dones_all, episode_rewards=episode_rewards_all, actions=actions_all,
                        **{f"frames_{k}": v for k, v in frames.items()})
    print(f"{name}: {case['steps']} steps, {rec['episodes_finished']} episodes, {len(gym_cls.calls)} shaping calls, {rec['nonzero_rewards']} non-zero rewards, reward sum {rewards_all.sum():.4f}")


def main():
    if not os.path.isdir(REFERENCE):
        raise SystemExit("the reference tree is not here: this generator runs in the build container only")
    gym_cls = install_stubs()
    sys.path.insert(0, REFERENCE)
    ref_env_mod = importlib.import_module("megaverse.megaverse_env")
    ref_utils_mod = importlib.import_module("megaverse_rl.megaverse_utils")
    assert ref_env_mod.__file__.startswith(REFERENCE) and ref_utils_mod.__file__.startswith(REFERENCE)
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--case", action="append", help="only these cases (default: all)")
    args = ap.parse_args()
    for name, case in CASES.items():
        if not args.case or name in args.case:
            run_case(name, case, ref_env_mod, ref_utils_mod, gym_cls, args.out)


if __name__ == "__main__":
    main()
