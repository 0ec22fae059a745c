"""Multi-task batches: several scenarios stepped side by side on one GPU.

Reference: megaverse/megaverse_env.py:11-39.  The reference runs a multi-task job as one MegaverseGym per scenario
(``make_env_multitask`` picks ``tasks[task_idx % len(tasks)]`` per worker); BASELINE.json configs[4] deals the scenarios
round-robin by env index.  ``MultiTaskGym`` owns one HIP gym per scenario -- own state, own episode feeder -- and steps them
as ONE group (``mv_group``): one step launch whose workgroups run their own scenario's tick, at most two observation launches
(the short-list and the long-list raster variant), one pair of streams, pipelined like a single gym; three launches per tick
instead of sixteen.  All of them write into ONE observation slab, scenario-major: frames of scenario k are rows
[k * n_k * A, (k + 1) * n_k * A).  (``MV_MULTITASK_UNION=0``: the round-2 scheme, one stream per scenario and one launch pair
per gym, kept for comparison.)

Global env index i  <->  (scenario i % S, local env i // S).  Seeds and the benchmark's random actions are drawn per
GLOBAL env index (mv_config.env_stride), so the job is bit-identical to S separately seeded single-scenario jobs that
share one master stream -- and every sub-gym is bit-exact against the oracle by the single-scenario parity tests.
"""
import os

import numpy as np

from .extension import GymGroup, MegaverseGym

# megaverse_env.py:18-21 of the reference: the eight scenarios of its multi-task benchmark, all available on the HIP path
# (Sokoban reads Boxoban level files: $BOXOBAN_LEVELS, as in the reference)
MEGAVERSE8 = ["TowerBuilding", "ObstaclesEasy", "ObstaclesHard", "Collect", "Sokoban", "HexMemory", "HexExplore", "Rearrange"]
MEGAVERSE_IN_SCOPE = MEGAVERSE8   # (older name)


class MultiTaskGym:
    def __init__(self, scenarios, w, h, num_envs, num_agents_per_env, num_simulation_threads=4, float_params=None, device=0,
                 env_offset=0, total_envs=0):
        S = len(scenarios)
        if num_envs % S:
            raise ValueError("num_envs must be a multiple of the number of scenarios")
        self.scenarios = list(scenarios)
        self.w, self.h, self.num_envs, self.num_agents_per_env = int(w), int(h), int(num_envs), int(num_agents_per_env)
        self.per_task = num_envs // S
        total = total_envs if total_envs > 0 else num_envs
        self.gyms = [MegaverseGym(name, w, h, self.per_task, num_agents_per_env, num_simulation_threads, False, float_params or {},
                                  device=device, env_offset=env_offset + k, total_envs=total, env_stride=S)
                     for k, name in enumerate(self.scenarios)]
        self.union = os.environ.get("MV_MULTITASK_UNION", "1") != "0" and len(self.gyms) <= 8
        if not self.union:
            for g in self.gyms:   # the sub-gyms overlap each other, one stream each: a second (simulation) stream per gym only
                g.set_pipelining(False)   # oversubscribes the hardware queues (measured: 8 gyms, 64 x 64: 3.5 M vs 6.0 M obs/s)
        self._group = None
        self._streams = None
        self._obs = None
        self._handles = None
        self._sample = None
        self.ring_obs = self.ring_rewards = self.ring_dones = None

    # ---- plumbing: torch owns the slab and the streams
    def attach(self, torch_device):
        import torch
        A = self.num_agents_per_env
        return self.attach_tensor(torch.empty((self.num_envs * A, self.h, self.w, 4), dtype=torch.uint8, device=torch_device))

    def attach_tensor(self, obs):
        """render into the caller's [num_envs * A, h, w, 4] uint8 slab (one stream per scenario, created on first use)"""
        import torch
        A, n = self.num_agents_per_env, self.per_task
        assert tuple(obs.shape) == (self.num_envs * A, self.h, self.w, 4) and obs.dtype == torch.uint8 and obs.is_contiguous()
        if self.union:
            if self._group is None:   # one stream for all of them: torch's current one
                cur = torch.cuda.current_stream(obs.device).cuda_stream
                for g in self.gyms:
                    g.set_stream(cur)
                self._group = GymGroup(self.gyms)
        elif self._streams is None:
            self._streams = [torch.cuda.Stream(device=obs.device) for _ in self.gyms]
            for k, g in enumerate(self.gyms):
                g.set_stream(self._streams[k].cuda_stream)
        self._obs = obs
        frame_bytes = self.h * self.w * 4
        for k, g in enumerate(self.gyms):
            g.set_obs_buffer(obs.data_ptr() + k * n * A * frame_bytes)
        return obs

    def set_output_ring(self, count):
        """Rollout rings, one set per scenario (mv_set_output_ring): tick t of sub-gym k leaves its observations in ``ring_obs[k][t % count]``
        ([count, n_k * A, h, w, 4] uint8), its rewards in ``ring_rewards[k][t % count]`` and its dones in ``ring_dones[k][t % count]``.  With rings at
        least as deep as a call (of 2 ... 8 ticks; up to 1024 envs in the group), ``step_n`` is TWO launches for all scenarios and all of its ticks (one union
        step launch, one union observation launch); otherwise two launches per tick.  count = 0: back to the shared slab.  -> (ring_obs, ring_rewards, ring_dones), lists of CUDA tensors."""
        import torch
        if count <= 0:
            for g in self.gyms:
                g.set_output_ring(0)
            self.ring_obs = self.ring_rewards = self.ring_dones = None
            return None
        dev = self._obs.device if self._obs is not None else torch.device("cuda", self.gyms[0].device if hasattr(self.gyms[0], "device") else 0)
        A, n = self.num_agents_per_env, self.per_task
        self.ring_obs = [torch.zeros((count, n * A, self.h, self.w, 4), dtype=torch.uint8, device=dev) for _ in self.gyms]
        self.ring_rewards = [torch.zeros((count, n * A), dtype=torch.float32, device=dev) for _ in self.gyms]
        self.ring_dones = [torch.zeros((count, n), dtype=torch.uint8, device=dev) for _ in self.gyms]
        torch.cuda.synchronize(dev)
        for k, g in enumerate(self.gyms):
            g.set_output_ring(count, self.ring_obs[k].data_ptr(), self.ring_rewards[k].data_ptr(), self.ring_dones[k].data_ptr())
        return self.ring_obs, self.ring_rewards, self.ring_dones

    def recommended_ticks_per_call(self):
        """the k to ask step_n for: what every member recommends (mv_recommended_ticks_per_call), at most 8 -- the two-launch group call's limit"""
        return max(1, min([8] + [g.recommended_ticks_per_call() for g in self.gyms]))

    def recommended_pass_overlap(self):
        return False

    def set_pixel_mode(self, mode):
        for g in self.gyms:
            g.set_pixel_mode(mode)

    def locate(self, env_idx):
        """global env index -> (sub-gym, local env index)"""
        S = len(self.gyms)
        return self.gyms[env_idx % S], env_idx // S

    def frame_row(self, env_idx, agent_idx=0):
        """row of the shared observation slab that holds (global env, agent)"""
        S, A = len(self.gyms), self.num_agents_per_env
        return ((env_idx % S) * self.per_task + env_idx // S) * A + agent_idx

    # ---- MegaverseGym surface, env indices are global
    def num_agents(self):
        return self.num_envs * self.num_agents_per_env

    def seed(self, seed):
        for g in self.gyms:
            g.seed(seed)

    def reset(self):
        for g in self.gyms:
            g.reset()

    def set_actions(self, env_idx, agent_idx, actions):
        g, j = self.locate(env_idx)
        g.set_actions(j, agent_idx, actions)

    def sample_random_actions(self, seed, step_index):
        self._sample = (int(seed) & 0xFFFFFFFF, int(step_index) & 0xFFFFFFFF)   # drawn inside the next step (one C call for all sub-gyms)

    def _ensure_group(self):
        if self.union and self._group is None:
            self._group = GymGroup(self.gyms)
        return self._group

    def step_n(self, k, policy="multidiscrete", seed=0, first_step_index=0):
        """k open-loop ticks of every scenario with one call (union launches; mv_group_step); without the union (MV_MULTITASK_UNION=0, or more
        scenarios than a group holds): k single steps of every sub-gym"""
        if not self.union:
            if policy != "multidiscrete":
                raise ValueError("MultiTaskGym.step_n without union launches supports the 'multidiscrete' policy only")
            for j in range(int(k)):
                self.sample_random_actions(seed, int(first_step_index) + j)
                self.step()
            return
        self._ensure_group().step(k, True, policy, seed, first_step_index)

    def step(self):
        import ctypes as C
        if self.union:
            seed, idx = self._sample if self._sample else (0, 0)
            self._ensure_group().step(1, True, "multidiscrete" if self._sample else "none", seed, idx)
            self._sample = None
            return
        if self._handles is None:
            self._handles = (C.c_void_p * len(self.gyms))(*[g._g for g in self.gyms])
        lib = self.gyms[0]._lib
        seed, idx = self._sample if self._sample else (0, 0)
        rc = lib.mv_step_many(self._handles, len(self.gyms), 1, 1 if self._sample else 0, seed, idx)
        self._sample = None   # (every sub-gym was stepped, whatever one of them reports: nothing to retry)
        if rc < 0:
            raise RuntimeError(lib.mv_last_error().decode())
        if rc > 0:
            import warnings
            warnings.warn(lib.mv_last_error().decode(), RuntimeWarning, stacklevel=2)

    def synchronize(self):
        for g in self.gyms:
            g.synchronize()

    def is_done(self, env_idx):
        g, j = self.locate(env_idx)
        return g.is_done(j)

    def get_observation(self, env_idx, agent_idx):
        g, j = self.locate(env_idx)
        return g.get_observation(j, agent_idx)

    def get_last_rewards(self):
        """env-major over GLOBAL env indices, like MegaverseGym::getLastRewards (megaverse.cpp:128-137)"""
        S, A = len(self.gyms), self.num_agents_per_env
        out = np.empty((self.per_task, S, A), np.float32)
        for k, g in enumerate(self.gyms):
            out[:, k, :] = g.get_rewards_array().reshape(self.per_task, A)
        return out.reshape(-1)

    def true_objective(self, env_idx, agent_idx):
        g, j = self.locate(env_idx)
        return g.true_objective(j, agent_idx)

    def profile_begin(self, n):
        for g in self.gyms:
            g.profile_begin(n)

    def profile_end(self):
        return [g.profile_end() for g in self.gyms]

    def close(self):
        if self._group is not None:
            self._group.close()
            self._group = None
        for g in self.gyms:
            g.close()
