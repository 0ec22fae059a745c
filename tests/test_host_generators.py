"""CPU parity of the product's HOST-side episode generators (mv_gen_obstacles.cpp, mv_gen_collect.cpp, reached through the
C ABI test hook mv_debug_generate_episode -- no GPU involved) against the oracle's Env::reset restatement: same
master seed -> same per-env seeds -> byte-identical layout slabs, terrain, movable boxes, diamonds, spawn cells,
spawn rotations and episode length, over several consecutive episodes of the same env stream."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from megaverse_amd import extension as ext

MAX_AGENTS, MAX_OBJECTS = 8, 80
LAYOUT_BOX = np.dtype([("min", "<i4", 3), ("type", "<i4"), ("max", "<i4", 3), ("slot", "<i4")])
TERRAIN_BOX = np.dtype([("min", "<i4", 3), ("type", "<i4"), ("max", "<i4", 3), ("pad", "<i4")])
OBJ = np.dtype([("x", "i1"), ("y", "i1"), ("z", "i1"), ("state", "i1")])
EPISODE_BLOB = np.dtype([
    ("seq", "<i4"), ("num_boxes", "<i4"), ("num_terrain", "<i4"), ("num_objects", "<i4"), ("num_rewards", "<i4"),
    ("num_platforms", "<i4"), ("layout_color", "<i4"), ("wall_color", "<i4"), ("draw_walls", "<i4"), ("dim", "<i4", 3),
    ("org", "<i4", 3), ("episode_len", "<f4"), ("spawn", "<i4", (MAX_AGENTS, 3)), ("yaw_frand", "<f4", MAX_AGENTS),
    ("boxes", LAYOUT_BOX, 128), ("terrain", TERRAIN_BOX, 16), ("objects", OBJ, MAX_OBJECTS), ("rewards", OBJ, 16),
], align=False)
COLLECT_BLOB = np.dtype([
    ("seq", "<i4"), ("num_boxes", "<i4"), ("num_objects", "<i4"), ("num_rewards", "<i4"), ("num_positive", "<i4"),
    ("layout_color", "<i4"), ("wall_color", "<i4"), ("dim", "<i4", 3), ("episode_len", "<f4"), ("pad", "<i4"),
    ("spawn", "<i4", (MAX_AGENTS, 3)), ("yaw_frand", "<f4", MAX_AGENTS), ("objects", OBJ, MAX_OBJECTS),
    ("rewards", OBJ, 96), ("heightmap", "i1", 1792), ("boxes", LAYOUT_BOX, 1024),
], align=False)


ITEM = np.dtype([("shape", "<i4"), ("color", "<i4"), ("off", "<i4", 3), ("pad", "<i4", 3)])
REARRANGE_BLOB = np.dtype([
    ("seq", "<i4"), ("num_boxes", "<i4"), ("num_items", "<i4"), ("max_matching", "<i4"), ("draw_walls", "<i4"), ("dim", "<i4", 3),
    ("episode_len", "<f4"), ("pad", "<i4", 3), ("spawn", "<i4", (MAX_AGENTS, 3)), ("yaw_frand", "<f4", MAX_AGENTS),
    ("boxes", LAYOUT_BOX, 16), ("items", ITEM, 8), ("objects", OBJ, 8),
], align=False)


HEX_BLOB = np.dtype([
    ("seq", "<i4"), ("num_boxes", "<i4"), ("num_colliders", "<i4"), ("num_objs", "<i4"), ("num_good", "<i4"), ("episode_len", "<f4"),
    ("target", "<f4", 2), ("spawn", "<f4", (MAX_AGENTS, 3)), ("yaw", "<f4", MAX_AGENTS),
    ("objs", oracle_lib.HEX_REC, oracle_lib.HEX_MAX_OBJS), ("boxes", oracle_lib.HEX_REC, oracle_lib.HEX_MAX_BOXES),
], align=False)


def generate(scenario, agents, env_seed, n, base_len=60.0):
    lib = ext.load_library()
    size = lib.mv_debug_generate_episode(scenario.encode(), agents, env_seed, n, base_len, None, 0)
    dt = COLLECT_BLOB if scenario.lower() == "collect" else REARRANGE_BLOB if scenario.lower() == "rearrange" else HEX_BLOB if scenario.lower().startswith("hex") else EPISODE_BLOB
    assert size == dt.itemsize, (size, dt.itemsize)
    buf = np.zeros(1, dt)
    assert lib.mv_debug_generate_episode(scenario.encode(), agents, env_seed, n, base_len, buf.ctypes.data, size) == size
    return buf[0]


def env_seeds(master, n):
    lo, hi = np.zeros(n, np.int32), np.full(n, 1 << 30, np.int32)
    out = np.empty(n, np.int32)
    oracle_lib.lib().mvo_rand_range_seq(master, lo.ctypes.data, hi.ctypes.data, n, out.ctypes.data)
    return out


def check_common(blob, snap, agents):
    nb = int(snap["num_boxes"])
    assert int(blob["num_boxes"]) == nb
    b = blob["boxes"][:nb]
    got = np.concatenate([b["min"], b["max"], b["type"][:, None], b["slot"][:, None]], axis=1)
    assert np.array_equal(got, snap["boxes"][:nb])
    no = int(snap["num_objects"])
    assert int(blob["num_objects"]) == no
    o = blob["objects"][:no]
    assert np.array_equal(np.stack([o["x"], o["y"], o["z"], o["state"]], 1), snap["objects"][:no])
    nr = int(snap["num_rewards"])
    assert int(blob["num_rewards"]) == nr
    r = blob["rewards"][:nr]
    assert np.array_equal(np.stack([r["x"], r["y"], r["z"], r["state"]], 1), snap["rewards"][:nr])
    assert np.float32(blob["episode_len"]).tobytes() == np.float32(snap["episode_len"]).tobytes()
    assert int(blob["layout_color"]) == int(snap["layout_color"]) and int(blob["wall_color"]) == int(snap["wall_color"])
    for k in range(agents):
        a = snap["agents"][k]
        assert np.array_equal(blob["spawn"][k], a["spawn"])
        # spawn rotation: yaw = frand * pi * 2 (scenario_default.hpp:87).  The bit-exact basis is checked on the GPU
        # (test_*_parity_gpu.py); here: the same draw, up to the rounding of the device's rotation-matrix helper
        ang = float(np.float32(np.float32(blob["yaw_frand"][k]) * np.float32(3.14159274)) * np.float32(2))
        assert abs(np.cos(ang) - float(a["basis"][0])) < 2e-6 and abs(np.sin(ang) - float(a["basis"][1])) < 2e-6


@pytest.mark.parametrize("scenario", ["ObstaclesEasy", "ObstaclesMedium", "ObstaclesHard", "ObstaclesWalls", "ObstaclesSteps", "ObstaclesLava"])
@pytest.mark.parametrize("agents", [1, 4])
def test_obstacles_generator_matches_oracle(scenario, agents):
    n_env, master = 12, 100 + agents
    og = oracle_lib.OracleGym(scenario, 32, 32, n_env, agents, 2)
    og.seed(master)
    seeds = env_seeds(master, n_env)
    for episode in (1, 2, 3):
        og.reset()
        for e in range(n_env):
            blob, snap = generate(scenario, agents, int(seeds[e]), episode), og.snapshot(e)
            check_common(blob, snap, agents)
            nt = int(snap["num_terrain"])
            assert int(blob["num_terrain"]) == nt and int(blob["num_platforms"]) == int(snap["num_platforms"])
            t = blob["terrain"][:nt]
            assert np.array_equal(np.concatenate([t["min"], t["max"], t["type"][:, None]], 1), snap["terrain"][:nt, :7])
            assert int(blob["draw_walls"]) == int(snap["draw_walls"])
    og.close()


@pytest.mark.parametrize("agents", [1, 3, 8])
def test_collect_generator_matches_oracle(agents):
    n_env, master = 40, 7 + agents
    og = oracle_lib.OracleGym("Collect", 32, 32, n_env, agents, 2)
    og.seed(master)
    seeds = env_seeds(master, n_env)
    for episode in (1, 2):
        og.reset()
        for e in range(n_env):
            blob, snap = generate("Collect", agents, int(seeds[e]), episode), og.snapshot(e)
            check_common(blob, snap, agents)
            assert np.array_equal(blob["heightmap"][:42 * 42], snap["heightmap"])
            assert int(blob["num_positive"]) == int(snap["num_platforms"])
            assert [int(v) for v in blob["dim"]] == [int(snap["L"]), int(snap["H"]), int(snap["W"])]
    og.close()


@pytest.mark.parametrize("agents", [1, 2, 8])
def test_rearrange_generator_matches_oracle(agents):
    n_env, master = 48, 31 + agents
    og = oracle_lib.OracleGym("Rearrange", 32, 32, n_env, agents, 2)
    og.seed(master)
    seeds = env_seeds(master, n_env)
    walls = set()
    for episode in (1, 2, 3):
        og.reset()
        for e in range(n_env):
            blob, snap = generate("Rearrange", agents, int(seeds[e]), episode), og.snapshot(e)
            nb = int(snap["num_boxes"])
            assert int(blob["num_boxes"]) == nb == 5
            b = blob["boxes"][:nb]
            got = np.concatenate([b["min"], b["max"], b["type"][:, None], b["slot"][:, None]], axis=1)
            assert np.array_equal(got, snap["boxes"][:nb]), (got.tolist(), snap["boxes"][:nb].tolist())
            ni = int(snap["num_items"])
            assert int(blob["num_items"]) == ni == int(snap["num_objects"]) and 1 <= ni <= 7
            it = blob["items"][:ni]
            assert np.array_equal(np.concatenate([it["shape"][:, None], it["color"][:, None], it["off"]], 1), snap["items"][:ni])
            o = blob["objects"][:ni]
            assert np.array_equal(np.stack([o["x"], o["y"], o["z"], o["state"]], 1), snap["objects"][:ni])
            assert int(blob["max_matching"]) == int(snap["num_platforms"]) and int(blob["draw_walls"]) == int(snap["draw_walls"])
            assert [int(v) for v in blob["dim"]] == [int(snap["L"]), int(snap["H"]), int(snap["W"])] and float(blob["episode_len"]) == 60.0
            walls.add(int(blob["draw_walls"]))
            for k in range(agents):
                a = snap["agents"][k]
                assert np.array_equal(blob["spawn"][k], a["spawn"])
                ang = float(np.float32(np.float32(blob["yaw_frand"][k]) * np.float32(3.14159274)) * np.float32(2))
                assert abs(np.cos(ang) - float(a["basis"][0])) < 2e-6 and abs(np.sin(ang) - float(a["basis"][1])) < 2e-6
    assert walls == {0, 1}   # both merge classes (walls drawn / not drawn) were exercised
    og.close()


def test_episode_length_param_is_honoured():
    b = generate("Collect", 1, 5, 1, base_len=10.0)
    assert float(b["episode_len"]) == 10.0 + 2.0 * int(b["num_rewards"])
    b = generate("ObstaclesEasy", 1, 5, 1, base_len=500.0)
    assert float(b["episode_len"]) == 500.0


@pytest.mark.parametrize("scenario,threads", [("ObstaclesHard", 1), ("ObstaclesHard", 6), ("Collect", 4), ("Rearrange", 3), ("HexMemory", 5), ("HexExplore", 2)])
def test_background_feeder_delivers_each_envs_stream_in_order(scenario, threads):
    """the worker pool (mv_feeder.cpp) against straight sequential generation, 5 episodes x 24 envs, no device involved"""
    lib = ext.load_library()
    rc = lib.mv_debug_feeder_selftest(scenario.encode(), 24, 2, threads, 5)
    assert rc == 0, lib.mv_last_error().decode()


@pytest.mark.parametrize("scenario", ["HexExplore", "HexMemory"])
@pytest.mark.parametrize("agents", [1, 3, 8])
def test_hex_generator_matches_oracle(scenario, agents):
    """mv_gen_hex.cpp against the oracle: the maze's boxes (floor, walls, edgings, landmarks; colliders first), the collectables, the
    agents' starting positions and rotations, the reward cell, the episode length -- byte for byte over three episodes of 24 env streams"""
    n_env, master = 24, 55 + agents
    og = oracle_lib.OracleGym(scenario, 32, 32, n_env, agents, 2)
    og.seed(master)
    seeds = env_seeds(master, n_env)
    sizes = set()
    for episode in (1, 2, 3):
        og.reset()
        for e in range(n_env):
            blob, snap = generate(scenario, agents, int(seeds[e]), episode), og.snapshot(e)
            nb, no = int(snap["hex_num_boxes"]), int(snap["hex_num_objs"])
            assert int(blob["num_boxes"]) == nb and int(blob["num_objs"]) == no
            assert blob["boxes"][:nb].tobytes() == snap["hex_boxes"][:nb].tobytes()
            assert blob["objs"][:no].tobytes() == snap["hex_objs"][:no].tobytes()
            ncol = int(((snap["hex_boxes"]["meta"][:nb] >> 4) & 1).sum())
            assert int(blob["num_colliders"]) == ncol and np.all((blob["boxes"]["meta"][:ncol] >> 4) & 1)
            assert int(blob["num_good"]) == int(snap["num_platforms"])
            assert np.float32(blob["episode_len"]).tobytes() == np.float32(snap["episode_len"]).tobytes()
            assert np.array_equal(blob["target"], snap["hex_target"][[0, 2]])
            for k in range(agents):
                a = snap["agents"][k]
                want = blob["spawn"][k] + np.array([0.5, np.float32(0.0) + np.float32(1.75), 0.5], np.float32)
                assert want.astype(np.float32).tobytes() == a["pos"].tobytes(), (k, want, a["pos"])
                ang = float(blob["yaw"][k])
                assert abs(np.cos(ang) - float(a["basis"][0])) < 2e-6 and abs(np.sin(ang) - float(a["basis"][1])) < 2e-6
            sizes.add(nb)
    assert len(sizes) > 10   # mazes of different sizes were drawn
    og.close()
