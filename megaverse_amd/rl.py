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
        self.episode_rewards = [0] * self.num_agents
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
        self.episode_rewards = [0] * self.num_agents
        return self.env.reset(), {}

    def _episode_bookkeeping(self, rewards, dones, infos):
        name = self.env.scenario_name.casefold()
        for i, info in enumerate(infos):
            self.episode_rewards[i] += rewards[i]
            if dones[i]:
                extra_stats = info.setdefault("episode_extra_stats", dict())
                info["true_objective"] = info["true_reward"]
                extra_stats[f"z_{name}_true_objective"] = info["true_reward"]
                extra_stats[f"z_{name}_reward"] = self.episode_rewards[i]
                approx_total_training_steps = self.training_info.get("approx_total_training_steps", 0)
                extra_stats["z_approx_total_training_steps"] = approx_total_training_steps
                self.episode_rewards[i] = 0
                if self.increase_team_spirit:
                    rew_shaping = self.get_current_reward_shaping(i)
                    rew_shaping["teamSpirit"] = min(approx_total_training_steps / self.max_team_spirit_steps, 1.0)
                    self.set_reward_shaping(rew_shaping, i)
                    extra_stats["teamSpirit"] = rew_shaping["teamSpirit"]

    def step(self, action):
        obs, rewards, dones, infos = self.env.step(action)
        self._episode_bookkeeping(rewards, dones, infos)
        return obs, rewards, dones, [False] * len(dones), infos

    def step_batched(self, actions=None):
        """-> (obs CUDA uint8 (num_agents, 3, H, W) view of the HBM slab, rewards np.float32 [num_agents],
        terminated np.bool_ [num_agents], truncated, infos) with the same bookkeeping as step()"""
        import numpy as np
        obs, rewards, dones_env = self.env.step_batched(actions)
        A = self.env.num_agents_per_env
        dones = np.repeat(dones_env, A)
        infos = [{} for _ in range(self.num_agents)]
        if dones_env.any():
            true_obj = self.env.env.get_true_objectives()
            for i in np.nonzero(dones)[0]:
                infos[i] = dict(true_reward=float(true_obj[i]))
        self._episode_bookkeeping(rewards, dones, infos)
        return obs, rewards, dones, np.zeros_like(dones), infos

    def render(self, *args, **kwargs):
        return self.env.render(*args, **kwargs)

    def close(self):
        self.env.close()


def make_megaverse(env_name, cfg=None, env_config=None, render_mode: Optional[str] = None, **kwargs):
    """megaverse_utils.py:96-122; extra keyword arguments (img_w, img_h, device, ...) go to MegaverseEnv"""
    cfg = cfg or DEFAULT_CFG
    scenario_name = env_name.casefold()
    if "multitask" in scenario_name:
        task_idx = env_config["worker_index"] if env_config is not None and "worker_index" in env_config else 0
        env = make_env_multitask(scenario_name, task_idx, num_envs=cfg.megaverse_num_envs_per_instance,
                                 num_agents_per_env=cfg.megaverse_num_agents_per_env,
                                 num_simulation_threads=cfg.megaverse_num_simulation_threads, use_vulkan=cfg.megaverse_use_vulkan)
    else:
        env = MegaverseEnv(scenario_name=scenario_name, num_envs=cfg.megaverse_num_envs_per_instance,
                           num_agents_per_env=cfg.megaverse_num_agents_per_env,
                           num_simulation_threads=cfg.megaverse_num_simulation_threads, use_vulkan=cfg.megaverse_use_vulkan, **kwargs)
    return Wrapper(env, cfg.megaverse_increase_team_spirit, cfg.megaverse_max_team_spirit_steps)
