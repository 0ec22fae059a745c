"""CPU tests of the Sokoban restatement in oracle/ (second half): spec-derived invariants of
scenario_sokoban.{hpp,cpp} on synthetic level files in the public Boxoban text format (tests/golden/boxoban)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import oracle_lib
from megaverse_amd.rollout import action_masks, sample_actions

LEVEL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "boxoban")


@pytest.fixture(autouse=True)
def boxoban_env(monkeypatch):
    monkeypatch.setenv("BOXOBAN_LEVELS", LEVEL_DIR)


def parse_levels():
    """every level a SokobanScenario can ever pick: all but the LAST level of each file (reloadLevels stores a level when it meets
    the next ';' line, scenario_sokoban.cpp:92-99)"""
    out = []
    for path in sorted(glob.glob(os.path.join(LEVEL_DIR, "unfiltered", "train", "*.txt"))):
        levels, cur = [], None
        for line in [l for l in open(path).read().split("\n") if l]:
            if line.startswith(";"):
                if cur is not None:
                    levels.append(cur)
                cur = []
            else:
                cur.append(line)
        out.extend(levels)          # `cur` (the file's last level) is dropped, like the reference does
    return out


def signature(rows):
    walls = {(x, z) for x, r in enumerate(rows) for z, c in enumerate(r) if c == "#"}
    goals = {(x, z) for x, r in enumerate(rows) for z, c in enumerate(r) if c in ".+"}      # '*' puts a box down but no goal (:163-168)
    boxes = sorted((x, z) for x, r in enumerate(rows) for z, c in enumerate(r) if c in "$*")
    player = [(x, z) for x, r in enumerate(rows) for z, c in enumerate(r) if c in "@+"]
    return frozenset(walls), frozenset(goals), tuple(boxes), tuple(player)


def test_reset_builds_one_of_the_file_levels_and_never_repeats_before_the_list_is_empty():
    known = {signature(r): i for i, r in enumerate(parse_levels())}
    assert len(known) >= 10
    n, A = 6, 2
    g = oracle_lib.OracleGym("Sokoban", 32, 32, n, A, 2)
    g.seed(9)
    seen = [[] for _ in range(n)]
    for episode in range(5):
        g.reset()
        for e in range(n):
            s = g.snapshot(e)
            assert s["scenario"] == 4 and float(s["episode_len"]) == 80.0 and int(s["H"]) == 3
            grid = s["soko"].reshape(32, 32)
            walls = frozenset((int(x), int(z)) for x, z in zip(*np.nonzero(grid == 1)))
            goals = frozenset((int(x), int(z)) for x, z in zip(*np.nonzero(grid == 2)))
            no = int(s["num_objects"])
            boxes = tuple(sorted((int(o[0]), int(o[2])) for o in s["objects"][:no]))
            assert all(int(o[1]) == 1 and int(o[3]) == 0 for o in s["objects"][:no])
            match = [i for sig, i in known.items() if sig[0] == walls and sig[1] == goals and sig[2] == boxes]
            assert len(match) == 1, (episode, e)
            seen[e].append(match[0])
            px, pz = [sig for sig, i in known.items() if i == match[0]][0][3][0]
            for k in range(A):     # agents start on the player's cell, staggered by half cells and lifted by 0.6 per agent (:151-156)
                x, y, z = (float(v) for v in s["agents"][k]["pos"])
                assert x == (px + (k % 2) * 0.5) * 2 + 0.5 and z == (pz + (k % 4 > 1) * 0.5) * 2 + 0.5
                assert abs(y - (2 + 0.6 * k + 1.75)) < 1e-6
            nb = int(s["num_boxes"])
            b = s["boxes"][:nb]
            vol = ((b[:, 3] - b[:, 0]) * (b[:, 4] - b[:, 1]) * (b[:, 5] - b[:, 2])).sum()
            assert vol == 100 + 2 * len(walls)      # 10 x 10 floor cells + two voxels per wall cell
    for e in range(n):
        assert len(set(seen[e])) == len(seen[e])    # 5 episodes < 6 usable levels per file: no repeats within one load
    g.close()


def test_default_reward_shaping_and_episode_length_param():
    g = oracle_lib.OracleGym("Sokoban", 16, 16, 1, 1)
    for k, v in {"teamSpirit": 0.0, "sokobanBoxOnTarget": 1.0, "sokobanBoxLeavesTarget": -1.0, "sokobanAllBoxesOnTarget": 10.0}.items():
        found = C.c_int(0)
        assert g.L.mvo_get_reward_shaping(g.g, 0, 0, k.encode(), C.byref(found)) == v and found.value == 1
    g.close()
    g = oracle_lib.OracleGym("Sokoban", 16, 16, 1, 1, 1, False, {"episodeLengthSec": 20.0})
    g.seed(1); g.reset()
    assert float(g.snapshot(0)["episode_len"]) == 20.0
    g.close()


def test_boxes_move_one_cell_at_a_time_never_into_walls_or_each_other_and_rewards_follow_the_goals():
    n, A = 8, 1
    g = oracle_lib.OracleGym("Sokoban", 16, 16, n, A, 4)
    g.seed(4); g.reset()
    prev = [g.snapshot(e) for e in range(n)]
    moved = 0
    for st in range(1250):
        m = action_masks(sample_actions(6, st, n * A))
        for e in range(n):
            g.set_action_mask(e, 0, int(m[e]))
        g.step_norender()
        rew = g.get_last_rewards()
        for e in range(n):
            s = g.snapshot(e)
            if g.is_done(e) or int(s["num_frames"]) < int(prev[e]["num_frames"]):
                prev[e] = s
                continue
            no = int(s["num_objects"])
            grid = s["soko"].reshape(32, 32)
            cells = [tuple(int(v) for v in o[:3]) for o in s["objects"][:no]]
            before = [tuple(int(v) for v in o[:3]) for o in prev[e]["objects"][:no]]
            assert len(set(cells)) == no
            changed = [i for i in range(no) if cells[i] != before[i]]
            assert len(changed) <= 1
            delta = 0
            for i in changed:
                moved += 1
                d = np.abs(np.array(cells[i]) - np.array(before[i]))
                assert d.sum() == 1
                (x, y, z), (bx, by, bz) = cells[i], before[i]
                assert not (y == 1 and grid[x, z] == 1)                                    # never into a wall cell
                delta = int(y == 1 and grid[x, z] == 2) - int(by == 1 and grid[bx, bz] == 2)
            on_goal = sum(1 for (x, y, z) in cells if y == 1 and grid[x, z] == 2)
            if delta == 1:
                assert rew[e] == (11.0 if on_goal == no and prev[e]["solved"] == 0 and int(s["highest_tower"]) == no else 1.0)
            elif delta == -1:
                assert rew[e] == -1.0
            else:
                assert rew[e] == 0.0
            prev[e] = s
    assert moved > 10
    g.close()


def test_product_host_generator_matches_the_oracle():
    """mv_gen_sokoban.cpp (through the host-only hook mv_debug_generate_sokoban) against the oracle: same master seed -> same
    level sequence per env (file pick, shuffle, pops), same merged slabs, cells, boxes, spawn positions and rotations"""
    from megaverse_amd import extension as ext
    from test_host_generators import LAYOUT_BOX, OBJ, env_seeds
    blob_t = np.dtype([("seq", "<i4"), ("num_boxes", "<i4"), ("num_objects", "<i4"), ("dim", "<i4", 3), ("floor_color", "<i4"),
                       ("episode_len", "<f4"), ("spawn", "<f4", (8, 3)), ("yaw_frand", "<f4", 8), ("boxes", LAYOUT_BOX, 128),
                       ("objects", OBJ, 80), ("cells", "u1", 1024)], align=False)
    lib = ext.load_library()
    assert lib.mv_debug_generate_sokoban(1, 0, 1, 80.0, None, 0) == blob_t.itemsize
    n_env, A, master, episodes = 5, 3, 123, 8      # 8 > 6 usable levels per file: crosses a reload
    og = oracle_lib.OracleGym("Sokoban", 16, 16, n_env, A, 2)
    og.seed(master)
    seeds = env_seeds(master, n_env)
    blobs = []
    for e in range(n_env):
        buf = np.zeros(episodes, blob_t)
        assert lib.mv_debug_generate_sokoban(A, int(seeds[e]), episodes, 80.0, buf.ctypes.data, buf.nbytes) == episodes, lib.mv_last_error()
        blobs.append(buf)
    for ep in range(episodes):
        og.reset()
        for e in range(n_env):
            b, s = blobs[e][ep], og.snapshot(e)
            nb = int(s["num_boxes"])
            assert int(b["num_boxes"]) == nb
            bb = b["boxes"][:nb]
            assert np.array_equal(np.concatenate([bb["min"], bb["max"], bb["type"][:, None], bb["slot"][:, None]], 1), s["boxes"][:nb])
            no = int(s["num_objects"])
            assert int(b["num_objects"]) == no
            o = b["objects"][:no]
            assert np.array_equal(np.stack([o["x"], o["y"], o["z"], o["state"]], 1), s["objects"][:no])
            assert np.array_equal(b["cells"], s["soko"]) and int(b["floor_color"]) == int(s["layout_color"])
            assert [int(v) for v in b["dim"]] == [int(s["L"]), int(s["H"]), int(s["W"])] and float(b["episode_len"]) == 80.0
            for k in range(A):
                want = np.array([b["spawn"][k][0] + np.float32(0.5), b["spawn"][k][1] + np.float32(0.0) + np.float32(1.75), b["spawn"][k][2] + np.float32(0.5)], np.float32)
                assert np.array_equal(want, s["agents"][k]["pos"]), (ep, e, k)
                ang = float(np.float32(np.float32(b["yaw_frand"][k]) * np.float32(3.14159274)) * np.float32(2))
                assert abs(np.cos(ang) - float(s["agents"][k]["basis"][0])) < 2e-6
    og.close()
