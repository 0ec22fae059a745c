"""Learner-facing shim: the reference's Sample-Factory integration on top of the HIP simulator.

Reference: megaverse_rl/megaverse_utils.py:10-122 (MEGAVERSE_ENVS, class Wrapper, make_megaverse) and
megaverse_rl/megaverse_params.py:23-54 (the megaverse_* config fields).  Same behaviour: PBT reward-shaping interface,
per-episode extra stats (``z_<scenario>_true_objective``, ``z_<scenario>_reward``), optional team-spirit annealing, and
the gymnasium-style 5-tuple ``step``.  gym and sample_factory are optional here (the two interfaces the reference
inherits are duck-typed below); what is added is ``step_batched``: observations stay in the HBM slab the raster kernel
wrote (a CUDA uint8 tensor view, zero copies), which is what a PyTorch-ROCm learner on the same GPU should consume.
"""
from types import SimpleNamespace
from typing import Optional

from .megaverse_env import MegaverseEnv, make_env_multitask


class MegaverseSpec:
    def __init__(self, name):
        self.name = name


MEGAVERSE_ENVS = [MegaverseSpec(n) for n in (
    "TowerBuilding", "ObstaclesEasy", "ObstaclesHard", "Collect", "Sokoban", "HexMemory", "HexExplore", "Rearrange",
    "multitask_Obstacles", "multitask_megaverse8")]

# megaverse_params.py:23-54 defaults
DEFAULT_CFG = SimpleNamespace(megaverse_num_envs_per_instance=1, megaverse_num_agents_per_env=4, megaverse_num_simulation_threads=1,
                              megaverse_use_vulkan=False, megaverse_increase_team_spirit=False, megaverse_max_team_spirit_steps=1e9)


class Wrapper:
    """PBT reward shaping + multi-task summaries (megaverse_utils.py:30-93)."""

    def __init__(self, env, increase_team_spirit=False, max_team_spirit_steps=1e9):
        self.env = env
        self.unwrapped = env
        self.num_agents = env.num_agents
        self.is_multiagent = env.is_multiagent
        self.action_space, self.observation_space = env.action_space, env.observation_space
        import numpy as np
        self._returns = np.zeros(self.num_agents, np.float64)
        self.increase_team_spirit = increase_team_spirit
        self.max_team_spirit_steps = max_team_spirit_steps
        self.training_info = {}          # TrainingInfoInterface: the learner writes approx_total_training_steps here

    def set_training_info(self, training_info):
        self.training_info = training_info

    def get_default_reward_shaping(self):
        return self.env.get_default_reward_shaping()

    def get_current_reward_shaping(self, agent_idx: int):
        return self.env.get_current_reward_shaping(agent_idx)

    def set_reward_shaping(self, reward_shaping: dict, agent_idx: int):
        return self.env.set_reward_shaping(reward_shaping, agent_idx)

    def seed(self, seed=None):
        return self.env.seed(seed)

    def reset(self, **kwargs):
        self._returns[:] = 0.0
        return self.env.reset(), {}

    @property
    def episode_rewards(self):
        """per-agent return of the running episodes: the reference keeps a mutable Python list (megaverse_utils.py:41,58); here THE float64
        vector itself -- `w.episode_rewards[i] = 0` and `w.episode_rewards[i] += r` write through like they do there"""
        return self._returns

    @episode_rewards.setter
    def episode_rewards(self, values):
        import numpy as np
        self._returns[:] = np.asarray(values, dtype=np.float64)

    def _finish_episodes(self, rewards, dones, infos):
        """Vectorised episode statistics (what megaverse_utils.py:61-86 does agent by agent): add the step's rewards to the running
        returns, then -- only for the agents whose episode ended -- fill the learner-facing keys, anneal teamSpirit and zero the return.
        `infos` entries of finished agents must carry `true_reward` (MegaverseEnv.step / step_batched put it there)."""
        import numpy as np
        self._returns += np.asarray(rewards, dtype=np.float64)
        finished = np.flatnonzero(np.asarray(dones, dtype=bool))
        if finished.size == 0:
            return
        scenario = self.env.scenario_name.casefold()
        steps_so_far = self.training_info.get("approx_total_training_steps", 0)
        team_spirit = min(steps_so_far / self.max_team_spirit_steps, 1.0) if self.increase_team_spirit else None
        episode_returns = self._returns[finished].tolist()
        self._returns[finished] = 0.0
        for i, ret in zip(finished.tolist(), episode_returns):
            info = infos[i]
            info["true_objective"] = info["true_reward"]
            stats = info.setdefault("episode_extra_stats", {})
            stats.update({f"z_{scenario}_true_objective": info["true_reward"], f"z_{scenario}_reward": ret,
                          "z_approx_total_training_steps": steps_so_far})
            if team_spirit is not None:
                shaping = self.get_current_reward_shaping(i)
                shaping["teamSpirit"] = team_spirit
                self.set_reward_shaping(shaping, i)
                stats["teamSpirit"] = team_spirit

    def step(self, action):
        obs, rewards, dones, infos = self.env.step(action)
        self._finish_episodes(rewards, dones, infos)
        return obs, rewards, dones, [False] * len(dones), infos

    def step_batched(self, actions=None):
        """-> (obs CUDA uint8 (num_agents, 3, H, W) view of the HBM slab, rewards np.float32 [num_agents],
        terminated np.bool_ [num_agents], truncated, infos) with the same bookkeeping as step(); no per-agent Python work on
        steps where no episode ends"""
        import numpy as np
        obs, rewards, dones_env = self.env.step_batched(actions)
        dones = np.repeat(dones_env, self.env.num_agents_per_env)
        infos = [{} for _ in range(self.num_agents)]
        if dones_env.any():
            true_obj = self.env.env.get_true_objectives()
            for i in np.flatnonzero(dones).tolist():
                infos[i] = {"true_reward": float(true_obj[i])}
        self._finish_episodes(rewards, dones, infos)
        return obs, rewards, dones, np.zeros_like(dones), infos

    def render(self, *args, **kwargs):
        return self.env.render(*args, **kwargs)

    def close(self):
        self.env.close()


def make_megaverse(env_name, cfg=None, env_config=None, render_mode: Optional[str] = None, **kwargs):
    """Sample-Factory env factory with the reference's signature (megaverse_utils.py:96-122): `env_name` is a scenario or a
    `multitask_*` set (the worker index picks the task), `cfg` carries the megaverse_* options (megaverse_params.py:23-54).  Extra
    keyword arguments (img_w, img_h, device, params, ...) go to MegaverseEnv."""
    cfg = cfg or DEFAULT_CFG
    sim = dict(num_envs=cfg.megaverse_num_envs_per_instance, num_agents_per_env=cfg.megaverse_num_agents_per_env,
               num_simulation_threads=cfg.megaverse_num_simulation_threads, use_vulkan=cfg.megaverse_use_vulkan)
    name = env_name.casefold()
    if "multitask" in name:
        worker = (env_config or {}).get("worker_index", 0)
        env = make_env_multitask(name, worker, **sim)
    else:
        env = MegaverseEnv(scenario_name=name, **sim, **kwargs)
    return Wrapper(env, cfg.megaverse_increase_team_spirit, cfg.megaverse_max_team_spirit_steps)
