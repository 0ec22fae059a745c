"""MegaverseEnv: the reference's Python surface on top of the HIP simulator.

Reference: megaverse/megaverse_env.py:42-201 (class MegaverseEnv(gym.Env)).  Same constructor
arguments, attributes (num_envs, num_agents_per_env, num_agents, action_space, observation_space,
is_multiagent) and methods (seed/reset/step/render/close + reward-shaping accessors), same return
conventions: ``reset() -> [obs]*num_agents`` with obs uint8 (3, H, W) *un-flipped* (rows bottom-up),
``step(actions) -> (obs, rewards, dones, infos)`` with ``infos[i] = {'true_reward': ...}`` on done.

Differences, all additive:
  * img_w/img_h are constructor keywords (reference hard-codes 128x72, megaverse_env.py:51-52);
  * ``step_batched`` / ``step_device`` / ``observations_tensor`` return one device tensor instead of O(num_agents)
    numpy views (the Python loops at megaverse_env.py:121-130,138-141 cap the reference well below
    the GPU's rate);
  * gym and cv2 are optional.
"""
import numpy as np

from . import spaces
from .extension import MegaverseGym, set_megaverse_log_level

MEGAVERSE8 = ['TowerBuilding', 'ObstaclesEasy', 'ObstaclesHard', 'Collect', 'Sokoban', 'HexMemory', 'HexExplore', 'Rearrange']
OBSTACLES_MULTITASK = ['ObstaclesWalls', 'ObstaclesSteps', 'ObstaclesLava', 'ObstaclesEasy', 'ObstaclesHard']
# what libmegaverse_hip.so can construct (mv_create): every scenario of the reference's multi-task sets
SUPPORTED_SCENARIOS = ['TowerBuilding', 'ObstaclesEasy', 'ObstaclesMedium', 'ObstaclesHard', 'ObstaclesWalls', 'ObstaclesSteps', 'ObstaclesLava',
                       'Collect', 'Sokoban', 'HexMemory', 'HexExplore', 'Rearrange', 'Empty']
_warned_unsupported = False


def make_env_multitask(multitask_name, task_idx, num_envs, num_agents_per_env, num_simulation_threads, use_vulkan=False, params=None):
    """reference: megaverse_env.py:27-39"""
    assert 'multitask' in multitask_name
    if multitask_name.endswith('megaverse8'):
        tasks = MEGAVERSE8
    elif multitask_name.endswith('obstacles'):
        tasks = OBSTACLES_MULTITASK
    else:
        raise NotImplementedError()
    supported = [t for t in tasks if t in SUPPORTED_SCENARIOS]
    if len(supported) != len(tasks):   # say it once, up front, instead of failing in 2 of every 8 workers at construction
        global _warned_unsupported
        if not _warned_unsupported:
            import warnings
            warnings.warn(f"{multitask_name}: {sorted(set(tasks) - set(supported))} are not available in this build; "
                          f"the multi-task job is dealt over {supported}")
            _warned_unsupported = True
        tasks = supported
    scenario = tasks[task_idx % len(tasks)]
    return MegaverseEnv(scenario, num_envs, num_agents_per_env, num_simulation_threads, use_vulkan, params)


class MegaverseEnv:
    def __init__(self, scenario_name, num_envs, num_agents_per_env, num_simulation_threads=1, use_vulkan=False, params=None,
                 img_w=128, img_h=72, device=0, env_offset=0, total_envs=0):
        scenario_name = scenario_name.casefold()
        self.scenario_name = scenario_name
        self.is_multiagent = True
        set_megaverse_log_level(2)

        self.img_w, self.img_h, self.channels = int(img_w), int(img_h), 3
        self.use_vulkan = use_vulkan
        self.num_agents = num_envs * num_agents_per_env
        self.num_envs = num_envs
        self.num_agents_per_env = num_agents_per_env
        self.device = device

        float_params = {}
        if params is not None:
            for k, v in params.items():
                if isinstance(v, float):
                    float_params[k] = v
                else:
                    raise Exception('Params of type %r not supported', type(v))

        self.env = MegaverseGym(self.scenario_name, self.img_w, self.img_h, num_envs, num_agents_per_env, num_simulation_threads,
                                use_vulkan, float_params, device=device, env_offset=env_offset, total_envs=total_envs)
        self.default_shaping_scheme = self.env.get_reward_shaping(0, 0)
        self.action_space = self.generate_action_space(self.env.action_space_sizes())
        self.observation_space = spaces.Box(0, 255, (self.channels, self.img_h, self.img_w), dtype=np.uint8)
        self._obs_tensor = None
        self._host_obs = None
        self._dev_out = None

    @staticmethod
    def generate_action_space(action_space_sizes):
        return spaces.Tuple([spaces.Discrete(sz) for sz in action_space_sizes])

    def seed(self, seed=None):
        if seed is None:
            return
        assert isinstance(seed, int), 'Expect seed to be an integer'
        self.env.seed(seed)

    # ---- reference-shaped (list of per-agent numpy arrays) ----
    def observations(self):
        """list of (3, H, W) uint8, one per agent (megaverse_env.py:121-130)"""
        frames = self.observations_numpy()
        return [frames[i] for i in range(self.num_agents)]

    def observations_numpy(self):
        """(num_agents, 3, H, W) uint8 on the host: ONE D2H copy of the RGBA slab into a pinned buffer (PCIe: ~64 MB per step at 1024 x 128 x 128 -- the
        reference's getObservation is a view of host memory, megaverse.cpp:139-143, here the frames live in HBM), returned as a transposed view of it like
        the reference's np.transpose(obs[:, :, :3], (2, 0, 1)) (megaverse_env.py:121-130); valid until the next call"""
        torch = self._torch()
        slab = self.observations_tensor(rgba=True)
        if self._host_obs is None:
            self._host_obs = torch.empty(slab.shape, dtype=torch.uint8, pin_memory=True)
        self._host_obs.copy_(slab, non_blocking=True)
        torch.cuda.current_stream(slab.device).synchronize()
        return self._host_obs.numpy()[..., :3].transpose(0, 3, 1, 2)

    def reset(self):
        self.env.reset()
        return self.observations()

    def step(self, actions):
        self.env.set_actions_batched(np.asarray(actions, dtype=np.int32).reshape(self.num_agents, -1))
        self.env.step()
        dones_env = self.env.get_dones().astype(bool)
        A = self.num_agents_per_env
        dones = np.repeat(dones_env, A).tolist()          # (megaverse_env.py:149-150: the env's done, once per agent)
        infos = [{} for _ in range(self.num_agents)]
        if dones_env.any():                                # true_reward of the agents whose episode just ended (megaverse_env.py:152-156)
            true_obj = self.env.get_true_objectives()
            for env_i in np.nonzero(dones_env)[0]:
                for j in range(A):
                    infos[env_i * A + j] = dict(true_reward=float(true_obj[env_i * A + j]))
        rewards = self.env.get_last_rewards()
        return self.observations(), rewards, dones, infos

    # ---- batched device path ----
    def _torch(self):
        import torch
        return torch

    def observations_tensor(self, rgba=False):
        """uint8 CUDA tensor viewing the HBM observation slab written by the raster kernel:
        (num_agents, 3, H, W) (a permuted view, no copy) or (num_agents, H, W, 4) if rgba."""
        torch = self._torch()
        if self._obs_tensor is None:
            self._obs_tensor = torch.empty((self.num_agents, self.img_h, self.img_w, 4), dtype=torch.uint8, device=f'cuda:{self.device}')
            self.env.set_obs_buffer(self._obs_tensor.data_ptr())
            self.env.render()
        self.env.synchronize()
        if rgba:
            return self._obs_tensor
        return self._obs_tensor[..., :3].permute(0, 3, 1, 2)

    def step_batched(self, actions=None):
        """actions: int32 [num_agents, 6] (numpy, or a CUDA torch tensor) or None (keep what was set).
        Returns (obs uint8 CUDA view (num_agents,3,H,W), rewards float32 np [num_agents], dones bool np [num_envs])."""
        if self._obs_tensor is None:
            self.observations_tensor()   # (allocates: before the action buffer is handed over, not between hand-over and step)
        held = None
        if actions is not None:
            if hasattr(actions, 'data_ptr'):
                # Lifetime rule of mv_set_actions_device: the buffer is READ BY THE NEXT STEP KERNEL, in the order of the gym's stream -- it must stay
                # alive and unchanged until that step has been enqueued (a reset or a host-side action setter in between reads it at once instead).
                # int32, contiguous, [num_agents, 6]; `held` keeps a converted copy alive until step() has returned (the caching allocator does not
                # hand its memory to anyone on another stream before the stream's work is done).
                import torch
                held = actions.to(dtype=torch.int32).contiguous()
                self.env.set_actions_device(held.data_ptr())
            else:
                self.env.set_actions_batched(actions)
        self.env.step()
        del held
        return self.observations_tensor(), self.env.get_rewards_array(), self.env.get_dones().astype(bool)

    def step_device(self, actions=None):
        """step_batched without a host synchronisation: (obs uint8 (num_agents, 3, H, W), rewards float32 [num_agents], dones uint8 [num_envs]), all three
        CUDA tensors the step writes into (mv_set_output_ring with one entry), valid in the order of the gym's stream -- torch's current stream when the
        env's first observation was asked for -- until the next step.  What a learner whose policy runs on the device wants (the reference has no
        counterpart: its outputs are host arrays, megaverse_env.py:121-162); `infos` / true rewards stay available through env.get_true_objectives()."""
        torch = self._torch()
        if self._obs_tensor is None:
            self.observations_tensor()
        if self._dev_out is None:
            dev = self._obs_tensor.device
            self._dev_out = (torch.zeros(self.num_agents, dtype=torch.float32, device=dev), torch.zeros(self.num_envs, dtype=torch.uint8, device=dev))
            self.env.set_output_ring(1, self._obs_tensor.data_ptr(), self._dev_out[0].data_ptr(), self._dev_out[1].data_ptr())
        held = None
        if actions is not None:
            if hasattr(actions, 'data_ptr'):   # (the lifetime rule of step_batched)
                held = actions.to(dtype=torch.int32).contiguous()
                self.env.set_actions_device(held.data_ptr())
            else:
                self.env.set_actions_batched(actions)
        self.env.step()
        del held
        return self._obs_tensor[..., :3].permute(0, 3, 1, 2), self._dev_out[0], self._dev_out[1]

    # ---- rendering (megaverse_env.py:164-184): returns the tiled BGR image, shows it if cv2 exists ----
    def convert_obs(self, obs):
        if not self.use_vulkan:
            obs = obs[::-1]
        return np.ascontiguousarray(obs[:, :, [2, 1, 0]])

    def render(self, mode='human'):
        self.env.draw_overview()
        self.env.draw_hires()
        rows = []
        for env_i in range(self.num_envs):
            obs = [self.convert_obs(self.env.get_hires_observation(env_i, i)) for i in range(self.num_agents_per_env)]
            rows.append(np.concatenate(obs, axis=1))
        obs_final = np.concatenate(rows, axis=0)
        if mode == 'human':
            try:
                import cv2  # type: ignore
                cv2.imshow(f'agent_{id(self)}', obs_final)
                cv2.waitKey(1)
            except Exception:  # noqa: BLE001 - headless image
                pass
        return obs_final

    def get_default_reward_shaping(self):
        return self.default_shaping_scheme

    def get_current_reward_shaping(self, actor_idx: int):
        return self.env.get_reward_shaping(actor_idx // self.num_agents_per_env, actor_idx % self.num_agents_per_env)

    def set_reward_shaping(self, reward_shaping: dict, actor_idx: int):
        return self.env.set_reward_shaping(actor_idx // self.num_agents_per_env, actor_idx % self.num_agents_per_env, reward_shaping)

    def close(self):
        if self.env:
            self.env.close()
