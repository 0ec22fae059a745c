"""Size-independent invariants of the restated simulator (oracle, CPU): whatever the random policy does, agents stay inside the
world, never end a tick inside solid voxels, movable boxes always rest on something, carried boxes are carried by exactly one
agent, and the per-agent reward bookkeeping adds up.  The same invariants hold on the HIP path by bit-exact parity."""
import numpy as np
import pytest

import oracle_lib
from megaverse_amd.rollout import action_masks, sample_actions

CAP_BOTTOM = 0.525 + 0.33   # capsule centre to its lowest point


def rollout(scenario, n, A, steps, seed, every):
    g = oracle_lib.OracleGym(scenario, 16, 16, n, A, 4)
    g.seed(seed); g.reset()
    total = np.zeros(n * A, np.float64)
    for st in range(steps):
        m = action_masks(sample_actions(seed + 7, st, n * A))
        for e in range(n):
            for a in range(A):
                g.set_action_mask(e, a, int(m[e * A + a]))
        g.step_norender()
        r = g.get_last_rewards()
        done = np.repeat([g.is_done(e) for e in range(n)], A)
        total = np.where(done, 0.0, total + r)      # a done env starts a new episode (and reports 0 on that tick)
        if st % every == 0:
            yield st, g, total
    g.close()


@pytest.mark.parametrize("scenario,A", [("TowerBuilding", 3), ("Rearrange", 2)])
def test_room_scenarios_keep_agents_inside_and_boxes_supported(scenario, A):
    n = 10
    for st, g, total in rollout(scenario, n, A, 700, 3, 35):
        for e in range(n):
            s = g.snapshot(e)
            L, H, W = int(s["L"]), int(s["H"]), int(s["W"])
            chunk = s["chunk"].reshape(16, 32, 32)    # [y][z][x]
            for a in range(A):
                x, y, z = (float(v) for v in s["agents"][a]["pos"])
                assert 1.0 + 0.33 - 0.05 <= x <= L - 1 - 0.33 + 0.05 and 1.0 + 0.33 - 0.05 <= z <= W - 1 - 0.33 + 0.05, (st, e, a, x, z)
                assert y - CAP_BOTTOM >= 1.0 - 0.06, (st, e, a, y)          # never below the floor's top (allowed penetration 0.04)
                cell = (int(np.floor(x)), int(np.floor(y + 0.05)), int(np.floor(z)))
                assert not (chunk[cell[1], cell[2], cell[0]] & 1), (st, e, a, cell)   # centre never inside a solid voxel
                assert abs(float(s["agents"][a]["total_reward"]) - total[e * A + a]) < 1e-4
            no = int(s["num_objects"])
            objs = s["objects"][:no]
            carried = [int(o[3]) for o in objs if o[3] > 0]
            assert len(carried) == len(set(carried)) and all(1 <= c <= A for c in carried)       # one box per agent at most
            for a in range(A):
                c = int(s["agents"][a]["carrying"])
                assert (c >= 0) == ((a + 1) in carried) and (c < 0 or int(objs[c][3]) == a + 1)
            placed = {(int(o[0]), int(o[1]), int(o[2])) for o in objs if o[3] == 0}
            for (x, y, z) in placed:
                below_solid = bool(chunk[y - 1, z, x] & 1)
                assert below_solid or (x, y - 1, z) in placed, (st, e, (x, y, z))               # rests on the world or on a box
                assert bool(chunk[y, z, x] & 4) and not (chunk[y, z, x] & 1)                     # the grid knows it; not inside a wall


def test_obstacles_agents_never_end_a_tick_inside_the_level_geometry():
    n, A = 10, 2
    for st, g, total in rollout("ObstaclesHard", n, A, 600, 5, 30):
        for e in range(n):
            s = g.snapshot(e)
            nb = int(s["num_boxes"])
            boxes = s["boxes"][:nb]
            for a in range(A):
                p = np.array([float(v) for v in s["agents"][a]["pos"]])
                if all(abs(v % 1.0 - 0.5) < 1e-6 for v in p):
                    continue   # teleported to a voxel centre this tick (lava / fall: FallDetectionComponent::resetAgent): the capsule
                               # overlaps the floor until the next tick's depenetration, exactly like the reference
                for b in boxes:
                    if not (b[6] & 1):
                        continue
                    lo, hi = b[0:3].astype(float), b[3:6].astype(float)
                    q = np.clip(p, lo - [0, 0.525, 0], hi + [0, 0.525, 0])     # closest point of the box grown by the capsule's half height
                    assert np.linalg.norm(p - q) >= 0.33 - 0.1, (st, e, a, p, b[:6])   # ccd allowance 0.04 + max penetration depth 0.041, never more
                assert abs(float(s["agents"][a]["total_reward"]) - total[e * A + a]) < 1e-4


def test_collect_rewards_are_integers_of_the_shaping_table_and_diamonds_only_disappear():
    n, A = 10, 2
    seen = {}
    for st, g, total in rollout("Collect", n, A, 900, 11, 25):
        for e in range(n):
            s = g.snapshot(e)
            nr = int(s["num_rewards"])
            state = tuple(int(v) for v in s["rewards"][:nr, 3])
            prev = seen.get(e)
            if prev is not None and prev[0] < int(s["num_frames"]) and len(prev[1]) == nr:
                assert all(b == a or b == 0 for a, b in zip(prev[1], state)), (st, e)        # collected diamonds never come back
            seen[e] = (int(s["num_frames"]), state)
            assert int(s["solved"]) in (0, 1) and (int(s["solved"]) == 0 or int(s["highest_tower"]) >= int(s["num_platforms"]))


@pytest.mark.parametrize("scenario,A,W,H", [("TowerBuilding", 2, 64, 64), ("ObstaclesHard", 1, 64, 36), ("Collect", 2, 48, 48), ("Rearrange", 2, 64, 64),
                                            ("Sokoban", 1, 64, 64), ("HexMemory", 2, 40, 24), ("HexExplore", 1, 40, 24), ("Empty", 2, 33, 17)])
def test_tiled_raster_equals_brute_force(scenario, A, W, H, monkeypatch):
    """the tile-culled software raster bench.py times as the CPU baseline's raster leg draws the image of the brute-force checker, byte for byte:
    the culling rectangles are conservative, the per-pixel arithmetic and the draw order are the same code"""
    import os
    monkeypatch.setenv("BOXOBAN_LEVELS", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "boxoban"))
    from megaverse_amd.rollout import action_masks, sample_actions
    N = 4
    g = oracle_lib.OracleGym(scenario, W, H, N, A, 2, False, {})
    g.seed(9); g.reset()
    for rnd in range(3):
        for st in range(15):
            g.set_action_masks(action_masks(sample_actions(3, 15 * rnd + st, N * A)))
            g.step_norender()
        g.set_raster(False); g.render()
        brute = np.stack([g.get_observation(e, a).copy() for e in range(N) for a in range(A)])
        g.set_raster(True); g.render()
        tiled = np.stack([g.get_observation(e, a).copy() for e in range(N) for a in range(A)])
        assert brute[..., :3].max() > 0
        assert np.array_equal(brute, tiled), f"{scenario} round {rnd}: {int((brute != tiled).any(axis=-1).sum())} pixels differ"
    g.close()
