"""CPU tests of the Hex scenarios' building blocks in oracle/ (SURVEY.md 8f-4).  The honeycomb maze generator (src/libs/mazes in the
reference tree: vendored, dependency-free) is PINNED: the restatement is compared with the library itself, compiled in place into
oracle/_ref/libmv_ref_mazes.so with a seeded subclass of its Kruskal (the reference seeds it from std::random_device)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib

REF = os.path.join(oracle_lib.ORACLE_DIR, "_ref", "libmv_ref_mazes.so")


def maze(fn, size, seed):
    cells = C.c_int(0)
    n = fn(size, seed, C.byref(cells), None, None, None, None, None)
    counts = np.zeros(cells.value, np.int32); to = np.zeros(n, np.int32); xy = np.zeros((n, 4), np.float64)
    centers = np.zeros((cells.value, 2), np.float64); bounds = np.zeros(4, np.float64)
    assert fn(size, seed, C.byref(cells), counts.ctypes.data, to.ctypes.data, xy.ctypes.data, centers.ctypes.data, bounds.ctypes.data) == n
    return cells.value, counts, to, xy, centers, bounds


def bind(lib, name):
    fn = getattr(lib, name)
    fn.argtypes = [C.c_int, C.c_uint, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    fn.restype = C.c_int
    return fn


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libmv_ref_mazes.so is built from /root/reference (not present here)")
@pytest.mark.parametrize("size", [2, 3, 5, 8, 10])
def test_honeycomb_maze_matches_the_reference_library(size):
    ref, mine = bind(C.CDLL(REF), "mvref_hex_maze"), bind(oracle_lib.lib(), "mvo_hex_maze")
    for seed in (0, 1, 12345, 2 ** 31 + 7):
        a, b = maze(ref, size, seed), maze(mine, size, seed)
        assert a[0] == b[0] == 3 * size * (size - 1) + 1
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])     # borders left per cell and their neighbours, list order included
        assert np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])     # cell centres and bounds, bit for bit
        # border end points: the restatement uses a literal table for the seven angles' cos / sin (a compiler is free to call sincos()
        # for cos(): one ulp of a double in a cancelling sum), the library whatever its build calls
        assert np.abs(a[3] - b[3]).max() < 1e-15


@pytest.mark.parametrize("size", [2, 4, 7])
def test_honeycomb_maze_is_a_spanning_tree(size):
    cells, counts, to, xy, centers, bounds = maze(bind(oracle_lib.lib(), "mvo_hex_maze"), size, 99)
    # borders left between cells = inner edges - (cells - 1) removed by the spanning tree, each listed on both sides
    inner = int((to >= 0).sum()) // 2
    total_inner = (6 * cells - 6 * (2 * size - 1)) // 2       # every cell has 6 neighbours except along the rim
    assert inner == total_inner - (cells - 1)
    # passages (removed borders) connect everything: union-find over pairs of adjacent cells that do NOT share a remaining border
    start = np.concatenate([[0], np.cumsum(counts)])
    walls = {(i, int(t)) for i in range(cells) for t in to[start[i]:start[i + 1]] if t >= 0}
    parent = list(range(cells))
    def root(u):
        while parent[u] != u:
            parent[u] = parent[parent[u]]; u = parent[u]
        return u
    d = np.linalg.norm(centers[:, None, :] - centers[None, :, :], axis=-1)
    for i in range(cells):
        for j in range(i + 1, cells):
            if abs(d[i, j] - np.sqrt(3)) < 1e-9 and (i, j) not in walls:
                parent[root(i)] = root(j)
    assert len({root(i) for i in range(cells)}) == 1
    assert np.allclose(bounds, [-np.sqrt(3) * (size - 0.5), -(1.5 * size - 0.5), np.sqrt(3) * (size - 0.5), 1.5 * size - 0.5])


# ---- the scenarios (scenario_hex_explore.cpp, scenario_hex_memory.cpp, component_hexagonal_maze.cpp) -------------------------------
ROT = np.array([[0.8660254, 0.5], [0.8660254, -0.5], [0.0, 1.0]], np.float32)   # (cos, sin) of the three wall orientations


def hex_boxes(s):
    b = s["hex_boxes"][: s["hex_num_boxes"]]
    return b, (b["meta"] & 15) - 1, (b["meta"] >> 4) & 1


def hex_objs(s):
    o = s["hex_objs"][: s["hex_num_objs"]]
    m = o["meta"]
    return o, m & 15, (m >> 4) & 1, (m >> 8) & 1, ((m >> 12) & 255) - 128, ((m >> 20) & 255) - 128


def drive(g, rng, steps, check=None):
    N, A = g.num_envs, g.num_agents_per_env
    total = np.zeros(N * A, np.float64)
    for st in range(steps):
        for e in range(N):
            for a in range(A):
                g.set_action_mask(e, a, (int(rng.integers(0, 2048)) & ~(1 << 4)) | (1 << 3))
        g.step_norender()
        total += g.get_last_rewards()
        if check:
            check(st)
    return total


@pytest.mark.parametrize("scenario", ["HexExplore", "HexMemory"])
def test_hex_layout_is_what_the_maze_component_builds(scenario):
    g = oracle_lib.OracleGym(scenario, 32, 18, 6, 2)
    g.seed(11); g.reset()
    for e in range(6):
        s = g.snapshot(e)
        b, frame, collide = hex_boxes(s)
        assert frame[0] == -1 and collide[0] == 1                                    # the floor comes first, in the world frame
        assert np.all(frame[1:] >= 0)
        walls = b[1:][collide[1:] == 1]
        assert np.all(walls["color"] == 0x3a7fa6)                                    # DARK_BLUE
        assert np.allclose(walls["b"][:, 2] - walls["a"][:, 2], 0.3, atol=1e-4)        # 2 * 0.15 thick
        assert np.allclose(walls["a"][:, 1], 0.0, atol=1e-5)                          # stand on the floor: centre y == half height
        h = walls["b"][0, 1]
        assert 2 * 0.85 - 1e-5 <= h <= 2 * 1.4 + 1e-5 and np.allclose(walls["b"][:, 1], h)
        assert np.allclose(walls["b"][:, 0] - walls["a"][:, 0], 3.5, atol=1e-3)       # a border of a unit hexagon, times the maze scale
        # colliding boxes come first; one edging per wall (not colliding, 0.24 * wallHeight / 2 high), landmarks in between
        ncol = int(collide.sum())
        assert np.all(collide[:ncol] == 1) and np.all(collide[ncol:] == 0)
        edg = b[ncol:][np.isclose(b[ncol:]["b"][:, 1], 0.12 * h, atol=1e-5) & np.isclose(b[ncol:]["a"][:, 1], 0.0, atol=1e-6)]
        assert len(edg) == len(walls) and np.array_equal(np.sort(frame[1:ncol]), np.sort(frame[ncol:][np.isin(np.arange(len(b) - ncol), np.nonzero(np.isclose(b[ncol:]["b"][:, 1], 0.12 * h, atol=1e-5) & np.isclose(b[ncol:]["a"][:, 1], 0.0, atol=1e-6))[0])]))
        # the number of walls: rim borders + inner borders left by the spanning tree, minus the randomly omitted ones
        size = next(n for n in range(2, 9) if (b[0]["b"][0] - b[0]["a"][0]) / 4 < 3.5 * (0.5 * 3 ** 0.5 * (2 * n - 1) + 1e-3))
        cells = 3 * size * (size - 1) + 1
        rim, inner = 6 * (2 * size - 1), (6 * cells - 6 * (2 * size - 1)) // 2 - (cells - 1)
        assert rim <= len(walls) <= rim + inner
        o, shape, good, alive, vx, vz = hex_objs(s)
        if scenario == "HexExplore":
            assert len(o) == 1 and shape[0] == 1 and o["color"][0] == 0xd468ee and np.allclose(o["a"][0, 1], 1.2)
            assert abs(s["episode_len"] - 60.0) < 1e-6
        else:
            ngood = int((good[1:] == 1).sum())
            assert alive[0] == 0 and np.all(alive[1:] == 1) and ngood == s["num_platforms"] and len(o) - 1 - ngood in (0, ngood)
            assert abs(s["episode_len"] - (60.0 + 3.0 * ngood)) < 1e-4
            assert np.all(shape[1:][good[1:] == 1] == shape[0]) and np.all(o["color"][1:][good[1:] == 1] == o["color"][0])
            bad = good[1:] == 0
            if bad.any():
                assert (shape[1:][bad][0], o["color"][1:][bad][0]) != (shape[0], o["color"][0])
    g.close()


def wall_distance(s, p):
    """distance of point p (capsule centre) to the nearest colliding wall box grown by the capsule's half height"""
    b, frame, collide = hex_boxes(s)
    best = 1e9
    for k in range(3):
        w = b[(frame == k) & (collide == 1)]
        if not len(w):
            continue
        c, sn = ROT[k]
        q = np.array([c * p[0] - sn * p[2], p[1], sn * p[0] + c * p[2]], np.float64)
        lo, hi = w["a"].astype(np.float64), w["b"].astype(np.float64)
        lo[:, 1] -= 0.525; hi[:, 1] += 0.525
        d = np.linalg.norm(q - np.clip(q, lo, hi), axis=1)
        best = min(best, d.min())
    return best


@pytest.mark.parametrize("scenario", ["HexExplore", "HexMemory"])
def test_hex_agents_walk_on_the_floor_and_never_enter_a_wall(scenario):
    g = oracle_lib.OracleGym(scenario, 32, 18, 4, 2)
    g.seed(5); g.reset()
    rng = np.random.default_rng(1)
    worst = [1e9]

    def check(st):
        if st % 7:
            return
        for e in range(4):
            s = g.snapshot(e)
            for a in range(2):
                p = s["agents"]["pos"][a]
                assert p[1] > 0.5, (st, e, a, p)                                      # capsule centre above the floor (half height 0.525 + radius)
                worst[0] = min(worst[0], wall_distance(s, p))
    drive(g, rng, 600, check)
    assert worst[0] > 0.25 - 0.05, worst[0]                                           # capsule radius 0.25, penetration allowance 0.041
    g.close()


def test_hex_explore_is_solved_by_reaching_the_reward_object():
    g = oracle_lib.OracleGym("HexExplore", 32, 18, 2, 2)
    g.seed(3); g.reset()
    s = g.snapshot(0)
    t = s["hex_target"]
    assert np.linalg.norm(s["agents"]["pos"][0][[0, 2]] - t[[0, 2]]) > 3.0                # agents start in a far cell
    g.debug_set_agent_pos(0, 1, float(t[0]) + 0.4, 1.0, float(t[2]))
    g.step_norender()
    r = g.get_last_rewards().reshape(2, 2)
    assert np.allclose(r[0], [0.0, 5.0]) and np.allclose(r[1], 0.0)                      # rewardTeam with teamSpirit 0: the finder only
    s = g.snapshot(0)
    assert s["solved"] == 1 and abs(s["episode_len"] - s["episode_sec"] - (0.3 - 1 / 15)) < 1e-3   # doneWithTimer, then += dt
    o, shape, good, alive, vx, vz = hex_objs(s)
    assert alive[0] == 0 and o["a"][0, 1] > 1000.0
    steps = 0
    while not g.is_done(0):
        g.step_norender(); steps += 1
        assert steps < 10 and np.all(g.get_last_rewards() == 0)                        # solved once
    assert g.true_objective(0, 0) == 1.0 and g.true_objective(0, 1) == 1.0
    assert g.snapshot(0)["solved"] == 0                                                 # the next episode
    g.close()


def test_hex_memory_collects_good_and_bad_objects_and_finishes_when_all_good_ones_are_taken():
    g = oracle_lib.OracleGym("HexMemory", 32, 18, 1, 2)
    g.seed(9); g.reset()
    s = g.snapshot(0)
    o, shape, good, alive, vx, vz = hex_objs(s)
    assert np.allclose(np.linalg.norm(s["agents"]["pos"][:2][:, [0, 2]] - 0.5, axis=1), 1.5, atol=1e-5)   # a circle around (0.5, 0.5)
    bad = [i for i in range(1, len(o)) if not good[i]]
    goods = [i for i in range(1, len(o)) if good[i]]
    expect_sum = 0.0
    order = [bad[0]] + goods if bad else goods
    for n, i in enumerate(order):
        raw_y = 0.5                                                                     # grid coordinate of the object: y voxel 0
        g.debug_set_agent_pos(0, n % 2, float(o["a"][i, 0]), 0.8, float(o["a"][i, 2]))
        g.step_norender()
        r = g.get_last_rewards()
        s2 = g.snapshot(0)
        o2, _, _, alive2, _, _ = hex_objs(s2)
        assert alive2[i] == 0 and o2["a"][i, 1] > 99.0, (n, i, r)
        assert r[n % 2] == (1.0 if good[i] else -1.0)
        expect_sum += r.sum()
    assert s2["highest_tower"] == len(goods) and s2["solved"] == 0                       # noticed at the start of the next step
    g.step_norender()
    s3 = g.snapshot(0)
    assert s3["solved"] == 1 and abs(s3["episode_len"] - s3["episode_sec"] - (0.3 - 1 / 15)) < 1e-3
    steps = 0
    while not g.is_done(0):
        g.step_norender(); steps += 1
        assert steps < 10
    assert g.true_objective(0, 0) == 1.0
    g.close()


@pytest.mark.parametrize("scenario", ["HexExplore", "HexMemory"])
def test_hex_episodes_are_a_function_of_the_seed(scenario):
    def run(seed):
        g = oracle_lib.OracleGym(scenario, 32, 18, 3, 1)
        g.seed(seed); g.reset()
        total = drive(g, np.random.default_rng(4), 200)
        g.render()
        out = [g.snapshot(e).tobytes() for e in range(3)], total, np.stack([g.get_observation(e, 0).copy() for e in range(3)])
        g.close()
        return out
    a, b, c = run(21), run(21), run(22)
    assert a[0] == b[0] and np.array_equal(a[2], b[2]) and np.array_equal(a[1], b[1])
    assert a[0] != c[0]
    assert (a[2][..., :3] > 0).any()                                                     # something was drawn
