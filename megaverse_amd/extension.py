"""ctypes binding of libmegaverse_hip.so exposing the reference's ``MegaverseGym`` method table.

Reference: class MegaverseGym, src/libs/bindings/megaverse.cpp:34-292 (pybind11 module
``megaverse.extension.megaverse``).  Method names, arity and argument meaning are the same so that
``MegaverseEnv`` (megaverse_env.py here, megaverse/megaverse_env.py there) reads the same.  There is
no CPU fallback: if the shared library or a HIP device is missing, construction raises.
"""
import ctypes as C
import os
import warnings

import numpy as np

from . import build as _build

_LIB = None


class _Config(C.Structure):
    _fields_ = [
        ("scenario", C.c_char_p), ("obs_width", C.c_int32), ("obs_height", C.c_int32), ("num_envs", C.c_int32),
        ("num_agents_per_env", C.c_int32), ("num_simulation_threads", C.c_int32), ("use_vulkan", C.c_int32),
        ("device", C.c_int32), ("param_keys", C.POINTER(C.c_char_p)), ("param_vals", C.POINTER(C.c_float)),
        ("num_params", C.c_int32), ("env_offset", C.c_int32), ("total_envs", C.c_int32), ("env_stride", C.c_int32),
    ]


# every symbol include/megaverse_hip.h declares: (name, restype, argtypes)
_P, _I, _U, _F = C.c_void_p, C.c_int32, C.c_uint32, C.c_float
SYMBOLS = [
    ("mv_last_error", C.c_char_p, []), ("mv_device_count", C.c_int, []), ("mv_abi_version", C.c_int, []),
    ("mv_create", C.c_int, [C.POINTER(_Config), C.POINTER(_P)]),
    ("mv_close", C.c_int, [_P]), ("mv_destroy", C.c_int, [_P]),
    ("mv_num_agents", C.c_int, [_P]), ("mv_action_space_sizes", C.c_int, [_P]),
    ("mv_seed", C.c_int, [_P, _I]), ("mv_reset", C.c_int, [_P]),
    ("mv_set_actions", C.c_int, [_P, _I, _I, _P, _I]),
    ("mv_set_actions_batched", C.c_int, [_P, _P]), ("mv_set_actions_device", C.c_int, [_P, _P]),
    ("mv_sample_random_actions", C.c_int, [_P, _U, _U]),
    ("mv_step_many", C.c_int, [_P, _I, _I, _I, _U, _U]),
    ("mv_group_create", C.c_int, [_P, _I, C.POINTER(_P)]), ("mv_group_step", C.c_int, [_P, _I, _I, _I, _U, _U]), ("mv_group_destroy", C.c_int, [_P]),
    ("mv_step", C.c_int, [_P]), ("mv_step_no_render", C.c_int, [_P]), ("mv_render", C.c_int, [_P]),
    ("mv_step_n", C.c_int, [_P, _I, _I, _U, _U]), ("mv_set_sample_policy", C.c_int, [_P, _I]),
    ("mv_set_output_ring", C.c_int, [_P, _I, _P, _P, _P]),
    ("mv_set_pass_overlap", C.c_int, [_P, _I]),
    ("mv_recommended_ticks_per_call", C.c_int, [_P]), ("mv_recommended_pass_overlap", C.c_int, [_P]), ("mv_arena_bytes", C.c_int64, [_P]),
    ("mv_host_generator_threads", C.c_int, [_P]),
    ("mv_is_done", C.c_int, [_P, _I]), ("mv_get_dones", C.c_int, [_P, _P]),
    ("mv_get_last_rewards", C.c_int, [_P, _P]),
    ("mv_true_objective", C.c_int, [_P, _I, _I, C.POINTER(_F)]), ("mv_get_true_objectives", C.c_int, [_P, _P]),
    ("mv_get_observation", C.c_int, [_P, _I, _I, _P]),
    ("mv_obs_device_ptr", _P, [_P]), ("mv_rewards_device_ptr", _P, [_P]), ("mv_dones_device_ptr", _P, [_P]),
    ("mv_true_objectives_device_ptr", _P, [_P]),
    ("mv_set_obs_buffer", C.c_int, [_P, _P]), ("mv_set_stream", C.c_int, [_P, _P]),
    ("mv_set_pixel_mode", C.c_int, [_P, _I]), ("mv_get_pixel_mode", C.c_int, [_P]),
    ("mv_set_render_resolution", C.c_int, [_P, _I, _I]), ("mv_draw_hires", C.c_int, [_P]),
    ("mv_get_hires_observation", C.c_int, [_P, _I, _I, _P]), ("mv_draw_overview", C.c_int, [_P]),
    ("mv_num_reward_shaping_keys", C.c_int, [_P]), ("mv_reward_shaping_key", C.c_char_p, [_P, _I]),
    ("mv_get_reward_shaping", C.c_int, [_P, _I, _I, C.c_char_p, C.POINTER(_F)]),
    ("mv_set_reward_shaping", C.c_int, [_P, _I, _I, C.c_char_p, _F]),
    ("mv_synchronize", C.c_int, [_P]),
    ("mv_profile_begin", C.c_int, [_P, _I]), ("mv_profile_end", C.c_int, [_P, _P, _P]),
    ("mv_set_pipelining", C.c_int, [_P, _I]), ("mv_get_pipelining", C.c_int, [_P]),
    ("mv_debug_set_agent_pos", C.c_int, [_P, _I, _I, _F, _F, _F]),
    ("mv_debug_set_agent_yaw", C.c_int, [_P, _I, _I, _F, _F]), ("mv_debug_set_agent_velocity", C.c_int, [_P, _I, _I, _F, _F, _F]),
    ("mv_debug_snapshot_size", C.c_int, [_P]), ("mv_debug_snapshot", C.c_int, [_P, _I, _P]),
    ("mv_debug_rng", C.c_int, [_I, _U, _I, _P, _P, _I, _P]),
    ("mv_debug_math", C.c_int, [_I, _I, _P, _P, _I, _P]),
    ("mv_debug_generate_episode", C.c_int, [C.c_char_p, _I, _I, _I, _F, _P, _I]),
    ("mv_debug_feeder_selftest", C.c_int, [C.c_char_p, _I, _I, _I, _I]),
    ("mv_debug_generate_sokoban", C.c_int, [_I, _I, _I, _F, _P, _I]),
    ("mv_debug_collect_draw_host", C.c_int, [_I, _I, _I, _F, _P, _I]),
    ("mv_debug_collect_draw_device", C.c_int, [_I, _I, _P, _I, _I, _F, _P, C.c_int64, _P]),
]


def library_path():
    return _build.LIB


def load_library():
    """dlopen libmegaverse_hip.so (building it first if a source is newer) and type every symbol."""
    global _LIB
    if _LIB is None:
        path = _build.LIB
        if _build.is_stale() and os.path.exists("/opt/rocm/bin/hipcc"):
            _build.build()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)")
        # PyTorch wheels bundle their own HIP/HSA runtime under the system runtime's SONAME.  If torch is
        # imported AFTER this library, the process ends up with two runtimes and the second one to
        # initialise reports "no ROCm-capable device".  Importing torch first makes both share one.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(path)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)   # AttributeError here == the C ABI lost a symbol
            fn.restype, fn.argtypes = res, args
        _LIB = lib
    return _LIB


class GymGroup:
    """mv_group: up to eight gyms of one job stepped with union launches (one step launch, at most two observation launches per tick).
    The gyms must share device, observation size, agents per env and stream (set_stream first)."""

    def __init__(self, gyms):
        self._lib = load_library()
        self.gyms = list(gyms)
        handles = (_P * len(self.gyms))(*[g._g for g in self.gyms])
        h = _P()
        if self._lib.mv_group_create(handles, len(self.gyms), C.byref(h)) != 0:
            raise RuntimeError("mv_group_create: " + self._lib.mv_last_error().decode())
        self._h = h

    def step(self, k=1, render=True, policy="none", seed=0, first_step_index=0):
        rc = self._lib.mv_group_step(self._h, int(k), 1 if render else 0, int(MegaverseGym.POLICIES.get(policy, policy)), int(seed) & 0xFFFFFFFF,
                                     int(first_step_index) & 0xFFFFFFFF)
        if rc < 0:
            raise RuntimeError(self._lib.mv_last_error().decode())
        if rc > 0:
            warnings.warn(self._lib.mv_last_error().decode(), RuntimeWarning, stacklevel=2)

    def close(self):
        if self._h is not None:
            self._lib.mv_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_log_level = 2


def set_megaverse_log_level(level):
    """reference: setMegaverseLogLevel, megaverse.cpp:29-32.  The HIP library does not log."""
    global _log_level
    _log_level = int(level)


class MegaverseGym:
    """Same constructor and methods as the reference's pybind class (megaverse.cpp:267-292)."""

    def __init__(self, scenario, w, h, num_envs, num_agents_per_env, num_simulation_threads, use_vulkan, float_params,
                 device=0, env_offset=0, total_envs=0, env_stride=1):
        self._lib = load_library()
        fp = dict(float_params or {})
        keys = (C.c_char_p * max(1, len(fp)))(*[k.encode() for k in fp])
        vals = (C.c_float * max(1, len(fp)))(*[float(v) for v in fp.values()])
        self._keep = (keys, vals)
        cfg = _Config(scenario.encode(), int(w), int(h), int(num_envs), int(num_agents_per_env), int(num_simulation_threads),
                      int(bool(use_vulkan)), int(device), keys, vals, len(fp), int(env_offset), int(total_envs), int(env_stride))
        handle = _P()
        self._g = None
        if self._lib.mv_create(C.byref(cfg), C.byref(handle)) != 0:
            raise RuntimeError("mv_create: " + self._lib.mv_last_error().decode())
        self._g = handle
        self.w, self.h, self.num_envs, self.num_agents_per_env, self.device = int(w), int(h), int(num_envs), int(num_agents_per_env), int(device)
        self.render_w, self.render_h = 768, 432

    def _ck(self, rc):
        if rc < 0:
            raise RuntimeError(self._lib.mv_last_error().decode())
        return rc

    def _ckw(self, rc):
        """stepping calls: 1 = done, with a warning (a capacity limit was hit and is reported once; include/megaverse_hip.h)"""
        if rc < 0:
            raise RuntimeError(self._lib.mv_last_error().decode())
        if rc > 0:
            warnings.warn(self._lib.mv_last_error().decode(), RuntimeWarning, stacklevel=3)
        return rc

    # ---- the reference's method table ----
    def num_agents(self):
        return self.num_agents_per_env

    def action_space_sizes(self):
        out = (C.c_int32 * 6)()
        self._lib.mv_action_space_sizes(out)
        return list(out)

    def seed(self, seed):
        self._ck(self._lib.mv_seed(self._g, int(seed)))

    def reset(self):
        self._ckw(self._lib.mv_reset(self._g))

    def set_actions(self, env_idx, agent_idx, actions):
        arr = (C.c_int32 * len(actions))(*[int(a) for a in actions])
        self._ck(self._lib.mv_set_actions(self._g, int(env_idx), int(agent_idx), arr, len(actions)))

    def step(self):
        self._ckw(self._lib.mv_step(self._g))

    def is_done(self, env_idx):
        return bool(self._ck(self._lib.mv_is_done(self._g, int(env_idx))))

    def get_observation(self, env_idx, agent_idx):
        out = np.empty((self.h, self.w, 4), np.uint8)
        self._ck(self._lib.mv_get_observation(self._g, int(env_idx), int(agent_idx), out.ctypes.data))
        return out

    def get_last_rewards(self):
        out = np.empty(self.num_envs * self.num_agents_per_env, np.float32)
        self._ck(self._lib.mv_get_last_rewards(self._g, out.ctypes.data))
        return out.tolist()

    def true_objective(self, env_idx, agent_idx):
        v = _F()
        self._ck(self._lib.mv_true_objective(self._g, int(env_idx), int(agent_idx), C.byref(v)))
        return float(v.value)

    def set_render_resolution(self, w, h):
        self._ck(self._lib.mv_set_render_resolution(self._g, int(w), int(h)))
        self.render_w, self.render_h = int(w), int(h)

    def draw_hires(self):
        self._ck(self._lib.mv_draw_hires(self._g))

    def draw_overview(self):
        self._ck(self._lib.mv_draw_overview(self._g))

    def get_hires_observation(self, env_idx, agent_idx):
        out = np.empty((self.render_h, self.render_w, 4), np.uint8)
        self._ck(self._lib.mv_get_hires_observation(self._g, int(env_idx), int(agent_idx), out.ctypes.data))
        return out

    def get_reward_shaping(self, env_idx, agent_idx):
        out = {}
        for i in range(self._lib.mv_num_reward_shaping_keys(self._g)):
            key = self._lib.mv_reward_shaping_key(self._g, i)
            v = _F()
            self._ck(self._lib.mv_get_reward_shaping(self._g, int(env_idx), int(agent_idx), key, C.byref(v)))
            out[key.decode()] = float(v.value)
        return out

    def set_reward_shaping(self, env_idx, agent_idx, reward_shaping):
        # reference semantics: the map is REPLACED (scenario.hpp:215); keys this scenario does not
        # know are an error here instead of a later std::out_of_range
        for k, v in reward_shaping.items():
            self._ck(self._lib.mv_set_reward_shaping(self._g, int(env_idx), int(agent_idx), k.encode(), float(v)))

    def close(self):
        if self._g is not None:
            self._lib.mv_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- batched extensions (no per-agent Python loop; SURVEY.md 3.2 hot loop iii) ----
    def set_actions_batched(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.num_envs * self.num_agents_per_env, 6)
        self._ck(self._lib.mv_set_actions_batched(self._g, a.ctypes.data))

    def set_actions_device(self, device_ptr):
        """int32 [num_agents, 6] multi-discrete actions in device memory.  Nothing is launched here: the NEXT step kernel reads the buffer (in the
        order of the gym's stream), so the caller keeps it alive and unchanged until that step() call has returned; mv_reset and the host-side
        action setters read a pending buffer at once instead (last writer wins)."""
        self._ck(self._lib.mv_set_actions_device(self._g, _P(int(device_ptr))))

    def sample_random_actions(self, seed, step_index):
        self._ck(self._lib.mv_sample_random_actions(self._g, int(seed) & 0xFFFFFFFF, int(step_index) & 0xFFFFFFFF))

    def step_no_render(self):
        self._ckw(self._lib.mv_step_no_render(self._g))

    POLICIES = {"none": 0, "multidiscrete": 1, "single-bit": 2}

    def step_n(self, k, policy="multidiscrete", seed=0, first_step_index=0):
        """k open-loop ticks (each stepped and rendered) with one call; tick j draws its actions from (policy, seed, first_step_index + j)"""
        self._ckw(self._lib.mv_step_n(self._g, int(k), int(self.POLICIES.get(policy, policy)), int(seed) & 0xFFFFFFFF, int(first_step_index) & 0xFFFFFFFF))

    def set_sample_policy(self, policy):
        """'multidiscrete' (action_space.sample()) or 'single-bit' (the reference's megaverse_test_app policy) for sample_random_actions"""
        self._ck(self._lib.mv_set_sample_policy(self._g, int(self.POLICIES.get(policy, policy))))

    def set_output_ring(self, count, obs_ptr=0, rewards_ptr=0, dones_ptr=0):
        """tick t leaves its outputs in entry t % count of the given device rings (0 = keep that output in its single array)"""
        self._ck(self._lib.mv_set_output_ring(self._g, int(count), _P(int(obs_ptr) or None), _P(int(rewards_ptr) or None), _P(int(dones_ptr) or None)))

    def set_pass_overlap(self, on=True):
        """with a ring at least two calls deep, the observation passes of consecutive step_n calls overlap (include/megaverse_hip.h: mv_set_pass_overlap);
        an entry must then be consumed before the next stepping call after the one that produced it"""
        self._ck(self._lib.mv_set_pass_overlap(self._g, int(bool(on))))

    def recommended_ticks_per_call(self):
        """the k to ask step_n for (the measured rules of include/megaverse_hip.h: 16 for 1024..2047 frames per tick, else 8; 1 for few-tick episodes)"""
        return int(self._lib.mv_recommended_ticks_per_call(self._g))

    def recommended_pass_overlap(self):
        return bool(self._lib.mv_recommended_pass_overlap(self._g))

    def host_generator_threads(self):
        """threads of the host-side episode feeder; 0 = the episodes are drawn on the device (TowerBuilding; Collect with the device generator)"""
        return int(self._lib.mv_host_generator_threads(self._g))

    def arena_bytes(self):
        return int(self._lib.mv_arena_bytes(self._g))

    def render(self):
        self._ck(self._lib.mv_render(self._g))

    def synchronize(self):
        self._ck(self._lib.mv_synchronize(self._g))

    def get_dones(self):
        out = np.empty(self.num_envs, np.uint8)
        self._ck(self._lib.mv_get_dones(self._g, out.ctypes.data))
        return out

    def get_rewards_array(self):
        out = np.empty(self.num_envs * self.num_agents_per_env, np.float32)
        self._ck(self._lib.mv_get_last_rewards(self._g, out.ctypes.data))
        return out

    def get_true_objectives(self):
        out = np.empty(self.num_envs * self.num_agents_per_env, np.float32)
        self._ck(self._lib.mv_get_true_objectives(self._g, out.ctypes.data))
        return out

    def obs_device_ptr(self):
        return int(self._lib.mv_obs_device_ptr(self._g) or 0)

    def rewards_device_ptr(self):
        return int(self._lib.mv_rewards_device_ptr(self._g) or 0)

    def dones_device_ptr(self):
        return int(self._lib.mv_dones_device_ptr(self._g) or 0)

    def true_objectives_device_ptr(self):
        return int(self._lib.mv_true_objectives_device_ptr(self._g) or 0)

    def set_obs_buffer(self, device_ptr):
        self._ck(self._lib.mv_set_obs_buffer(self._g, _P(int(device_ptr))))

    def set_pixel_mode(self, mode):
        """'fast' (default: hardware rcp/rsqrt, pixels within DESIGN.md's tolerance) or 'exact' (bit-identical to the oracle)"""
        m = {"exact": 0, "fast": 1}.get(mode, mode)
        self._ck(self._lib.mv_set_pixel_mode(self._g, int(m)))

    def pixel_mode(self):
        return "fast" if self._lib.mv_get_pixel_mode(self._g) == 1 else "exact"

    def set_pipelining(self, on):
        """one-step-ahead pipelining of the step kernels against the observation pass (include/megaverse_hip.h); on by default"""
        self._ck(self._lib.mv_set_pipelining(self._g, 1 if on else 0))

    def pipelining(self):
        return self._lib.mv_get_pipelining(self._g) == 1

    def set_stream(self, hip_stream):
        self._ck(self._lib.mv_set_stream(self._g, _P(int(hip_stream))))

    def profile_begin(self, max_steps):
        self._ck(self._lib.mv_profile_begin(self._g, int(max_steps)))

    def profile_end(self):
        """-> {'step': (avg_ms, n), 'reset': (...), 'setup': (...), 'raster': (...)} measured with HIP events on the gym's stream"""
        ms = (C.c_float * 4)()
        cnt = (C.c_int32 * 4)()
        self._ck(self._lib.mv_profile_end(self._g, ms, cnt))
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(("step", "reset", "setup", "raster"))}

    def debug_set_agent_pos(self, env_idx, agent_idx, x, y, z):
        self._ck(self._lib.mv_debug_set_agent_pos(self._g, int(env_idx), int(agent_idx), float(x), float(y), float(z)))

    def debug_set_agent_yaw(self, env_idx, agent_idx, c, s):
        self._ck(self._lib.mv_debug_set_agent_yaw(self._g, int(env_idx), int(agent_idx), float(c), float(s)))

    def debug_set_agent_velocity(self, env_idx, agent_idx, hvx, hvz, vvel):
        self._ck(self._lib.mv_debug_set_agent_velocity(self._g, int(env_idx), int(agent_idx), float(hvx), float(hvz), float(vvel)))

    def debug_snapshot_bytes(self, env_idx):
        n = self._lib.mv_debug_snapshot_size(self._g)
        buf = np.zeros(n, np.uint8)
        self._ck(self._lib.mv_debug_snapshot(self._g, int(env_idx), buf.ctypes.data))
        return buf
