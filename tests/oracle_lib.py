"""ctypes binding of the CPU oracle (oracle/libmv_oracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

MAX_BOXES, MAX_OBJECTS, MAX_AGENTS, CHUNK = 1024, 80, 8, 32 * 16 * 32
MAX_TERRAIN, MAX_REWARDS, MAX_SHAPING, HM_DIM = 16, 96, 8, 42

SNAP_AGENT = np.dtype([
    ("pos", "<f4", 3), ("basis", "<f4", 4), ("pitch", "<f4"), ("hv", "<f4", 2), ("vvel", "<f4"), ("voffset", "<f4"),
    ("step_offset", "<f4"), ("jump_speed", "<f4"), ("was_jumping", "<i4"), ("carrying", "<i4"), ("picked_up", "<i4"),
    ("visited_zone", "<i4"), ("spawn", "<i4", 3), ("last_reward", "<f4"), ("total_reward", "<f4"), ("shaping", "<f4", MAX_SHAPING),
])
HEX_MAX_BOXES, HEX_MAX_OBJS = 2048, 128
HEX_REC = np.dtype([("a", "<f4", 3), ("meta", "<i4"), ("b", "<f4", 3), ("color", "<i4")])
SNAP = np.dtype([
    ("scenario", "<i4"), ("L", "<i4"), ("H", "<i4"), ("W", "<i4"), ("bz", "<i4", 4), ("layout_color", "<i4"), ("wall_color", "<i4"),
    ("draw_walls", "<i4"), ("num_objects", "<i4"), ("num_boxes", "<i4"), ("num_frames", "<i4"), ("done", "<i4"),
    ("highest_tower", "<i4"), ("num_agents", "<i4"), ("num_terrain", "<i4"), ("num_rewards", "<i4"), ("num_platforms", "<i4"),
    ("solved", "<i4"), ("episode_sec", "<f4"), ("episode_len", "<f4"), ("bz_reward", "<f4"),
    ("bar_half_width", "<f4"), ("boxes", "<i4", (MAX_BOXES, 8)), ("terrain", "<i4", (MAX_TERRAIN, 8)),
    ("objects", "i1", (MAX_OBJECTS, 4)), ("rewards", "i1", (MAX_REWARDS, 4)),
    ("agents", SNAP_AGENT, MAX_AGENTS), ("chunk", "u1", CHUNK), ("heightmap", "i1", HM_DIM * HM_DIM),
    ("num_items", "<i4"), ("items", "<i4", (8, 5)), ("soko", "u1", 32 * 32),
    ("hex_num_boxes", "<i4"), ("hex_num_objs", "<i4"), ("hex_target", "<f4", 3),
    ("hex_boxes", HEX_REC, HEX_MAX_BOXES), ("hex_objs", HEX_REC, HEX_MAX_OBJS),
])


def build_oracle():
    so = os.path.join(ORACLE_DIR, "libmv_oracle.so")
    src = os.path.join(ORACLE_DIR, "mv_oracle.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build_oracle())
        L.mvo_create.restype = C.c_void_p
        L.mvo_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p),
                                 C.POINTER(C.c_float), C.c_int]
        for name in ("mvo_close", "mvo_reset", "mvo_step", "mvo_step_norender", "mvo_render"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = None
        L.mvo_seed.argtypes = [C.c_void_p, C.c_int]
        L.mvo_set_actions.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
        L.mvo_set_action_mask.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.mvo_is_done.argtypes = [C.c_void_p, C.c_int]
        L.mvo_get_dones.argtypes = [C.c_void_p, C.c_void_p]
        L.mvo_render_env.argtypes = [C.c_void_p, C.c_int]
        L.mvo_set_raster.argtypes = [C.c_void_p, C.c_int]
        L.mvo_get_last_rewards.argtypes = [C.c_void_p, C.c_void_p]
        L.mvo_true_objective.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mvo_true_objective.restype = C.c_float
        L.mvo_get_observation.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mvo_get_observation.restype = C.c_void_p
        L.mvo_get_reward_shaping.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int)]
        L.mvo_get_reward_shaping.restype = C.c_float
        L.mvo_set_reward_shaping.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_float]
        L.mvo_debug_set_agent_pos.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
        L.mvo_debug_set_agent_yaw.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
        L.mvo_debug_set_agent_velocity.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
        L.mvo_set_action_masks.argtypes = [C.c_void_p, C.c_void_p]
        L.mvo_snapshot_size.argtypes = [C.c_void_p]
        L.mvo_snapshot.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.mvo_mt19937_nth.argtypes = [C.c_uint32, C.c_int]
        L.mvo_mt19937_nth.restype = C.c_uint32
        L.mvo_rand_range_seq.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.mvo_frand_seq.argtypes = [C.c_uint32, C.c_int, C.c_void_p]
        L.mvo_shuffle_iota.argtypes = [C.c_uint32, C.c_int, C.c_void_p]
        L.mvo_action_mask.argtypes = [C.POINTER(C.c_int), C.c_int]
        L.mvo_get_coords.argtypes = [C.c_void_p, C.c_void_p]
        L.mvo_building_reward_coeff.argtypes = [C.c_float]
        L.mvo_building_reward_coeff.restype = C.c_float
        L.mvo_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.mvo_perlin_octave2_01.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def ref_lib():
    """oracle/_ref/libmv_ref_util.so: the reference's own util.hpp compiled in place (or None)."""
    p = os.path.join(ORACLE_DIR, "_ref", "libmv_ref_util.so")
    if not os.path.exists(p):
        return None
    L = C.CDLL(p)
    L.mvref_rand_range_seq.argtypes = [C.c_uint, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.mvref_frand_seq.argtypes = [C.c_uint, C.c_int, C.c_void_p]
    L.mvref_random_bool_seq.argtypes = [C.c_uint, C.c_int, C.c_void_p]
    L.mvref_env_seeds.argtypes = [C.c_int, C.c_int, C.c_void_p]
    if hasattr(L, "mvref_perlin_octave2_01"):
        L.mvref_perlin_octave2_01.argtypes = [C.c_uint, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return L


class OracleGym:
    """Same method table as MegaverseGym (bindings/megaverse.cpp:267-292), backed by the oracle."""

    def __init__(self, scenario, w, h, num_envs, num_agents_per_env, num_simulation_threads=1, use_vulkan=False,
                 float_params=None):
        self.L = lib()
        fp = float_params or {}
        keys = (C.c_char_p * max(1, len(fp)))(*[k.encode() for k in fp])
        vals = (C.c_float * max(1, len(fp)))(*[float(v) for v in fp.values()])
        self.w, self.h, self.num_envs, self.num_agents_per_env = w, h, num_envs, num_agents_per_env
        self.g = self.L.mvo_create(scenario.encode(), w, h, num_envs, num_agents_per_env, num_simulation_threads, keys,
                                   vals, len(fp))
        if not self.g:
            raise RuntimeError("mvo_create failed")

    def num_agents(self): return self.num_agents_per_env
    def action_space_sizes(self): return [3, 3, 3, 2, 2, 3]
    def seed(self, s): self.L.mvo_seed(self.g, int(s))
    def reset(self): self.L.mvo_reset(self.g)

    def set_actions(self, env_idx, agent_idx, actions):
        arr = (C.c_int * len(actions))(*[int(a) for a in actions])
        self.L.mvo_set_actions(self.g, env_idx, agent_idx, arr, len(actions))

    def set_action_mask(self, env_idx, agent_idx, mask): self.L.mvo_set_action_mask(self.g, env_idx, agent_idx, int(mask))

    def set_action_masks(self, masks):
        m = np.ascontiguousarray(masks, dtype=np.int32).reshape(self.num_envs * self.num_agents_per_env)
        self.L.mvo_set_action_masks(self.g, m.ctypes.data)

    def step(self): self.L.mvo_step(self.g)
    def step_norender(self): self.L.mvo_step_norender(self.g)
    def render(self): self.L.mvo_render(self.g)
    def is_done(self, env_idx): return bool(self.L.mvo_is_done(self.g, env_idx))

    def get_dones(self):
        out = np.empty(self.num_envs, np.uint8)
        self.L.mvo_get_dones(self.g, out.ctypes.data)
        return out

    def render_env(self, env_idx): self.L.mvo_render_env(self.g, int(env_idx))
    def set_raster(self, tiled): self.L.mvo_set_raster(self.g, 1 if tiled else 0)

    def get_last_rewards(self):
        out = np.zeros(self.num_envs * self.num_agents_per_env, np.float32)
        self.L.mvo_get_last_rewards(self.g, out.ctypes.data)
        return out

    def true_objective(self, env_idx, agent_idx): return float(self.L.mvo_true_objective(self.g, env_idx, agent_idx))

    def get_observation(self, env_idx, agent_idx):
        p = self.L.mvo_get_observation(self.g, env_idx, agent_idx)
        buf = (C.c_uint8 * (self.h * self.w * 4)).from_address(p)
        return np.frombuffer(buf, np.uint8).reshape(self.h, self.w, 4)

    def get_reward_shaping(self, env_idx, agent_idx):
        out = {}
        for k in ("teamSpirit", "towerPickedUpObject", "towerVisitedBuildingZoneWithObject", "towerBuildingReward",
                  "obstaclesAgentAtExit", "obstaclesAllAgentsAtExit", "obstaclesExtraReward", "obstaclesAgentCarriedObjectToExit",
                  "memoryCollectGood", "memoryCollectBad", "exploreSolved",
                  "collectSingleGood", "collectSingleBad", "collectAll", "collectAbyss",
                  "rearrangeOneMoreObjectCorrectPosition", "rearrangeAllObjectsCorrectPosition",
                  "sokobanBoxOnTarget", "sokobanBoxLeavesTarget", "sokobanAllBoxesOnTarget"):
            f = C.c_int(0)
            v = self.L.mvo_get_reward_shaping(self.g, env_idx, agent_idx, k.encode(), C.byref(f))
            if f.value:
                out[k] = float(v)
        return out

    def set_reward_shaping(self, env_idx, agent_idx, rs):
        for k, v in rs.items():
            self.L.mvo_set_reward_shaping(self.g, env_idx, agent_idx, k.encode(), float(v))

    def debug_set_agent_pos(self, env_idx, agent_idx, x, y, z): self.L.mvo_debug_set_agent_pos(self.g, env_idx, agent_idx, x, y, z)
    def debug_set_agent_yaw(self, env_idx, agent_idx, c, s): self.L.mvo_debug_set_agent_yaw(self.g, env_idx, agent_idx, c, s)
    def debug_set_agent_velocity(self, env_idx, agent_idx, hvx, hvz, vvel): self.L.mvo_debug_set_agent_velocity(self.g, env_idx, agent_idx, hvx, hvz, vvel)

    def snapshot(self, env_idx):
        assert self.L.mvo_snapshot_size(self.g) == SNAP.itemsize, (self.L.mvo_snapshot_size(self.g), SNAP.itemsize)
        buf = np.zeros(1, SNAP)
        self.L.mvo_snapshot(self.g, env_idx, buf.ctypes.data)
        return buf[0]

    def close(self):
        if self.g:
            self.L.mvo_close(self.g)
            self.g = None
