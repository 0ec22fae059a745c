"""CPU tests of the Rearrange restatement in oracle/ (SURVEY.md §8f rank 1): spec-derived invariants of
scenario_rearrange.{hpp,cpp}."""
import ctypes as C

import numpy as np

import oracle_lib
from megaverse_amd.rollout import action_masks, sample_actions

SHAPES = {0, 1, 2, 4}   # Box, Capsule, Sphere, Cylinder (DrawableType)
OBJECT_COLORS = {0xffdd3c, 0x3bb372, 0x50c878, 0x2eb5d0, 0xadd8e6, 0x3a7fa6, 0xffb400, 0xb3b3b3, 0x555555, 0xffffff, 0xff0000,
                 0xffa770, 0xd468ee, 0xffe6e6}


def test_reset_invariants():
    n, A = 64, 3
    g = oracle_lib.OracleGym("Rearrange", 32, 32, n, A, 4)
    g.seed(17); g.reset()
    sizes = set()
    for e in range(n):
        s = g.snapshot(e)
        assert s["scenario"] == 3 and (int(s["L"]), int(s["W"])) == (19, 14) and 4 <= int(s["H"]) <= 6
        ni = int(s["num_items"])
        sizes.add(ni)
        assert 1 <= ni <= 7 and int(s["num_objects"]) == ni and float(s["episode_len"]) == 60.0
        items = s["items"][:ni]
        offs = [tuple(int(v) for v in it[2:]) for it in items]
        assert offs[0] == (0, 0, 0) and len(set(offs)) == ni
        for it, (x, y, z) in zip(items, offs):
            assert int(it[0]) in SHAPES and int(it[1]) in OBJECT_COLORS
            assert abs(x) <= 1 and abs(z) <= 1 and 0 <= y <= 1 and (y == 0 or (x, 0, z) in offs)   # stacked items stand on another one
        objs = [tuple(int(v) for v in o[:3]) for o in s["objects"][:ni]]
        assert len(set(objs)) == ni                                     # displaced items never share a cell
        for (x, y, z) in objs:
            assert abs(x - 13) <= 2 and abs(z - 5) <= 2 and 2 <= y <= 3
        matching = sum(1 for i in range(ni) if any(int(items[k][0]) == int(items[i][0]) and int(items[k][1]) == int(items[i][1]) and
                                                   offs[k] == (objs[i][0] - 13, objs[i][1] - 2, objs[i][2] - 5) for k in range(ni)))
        assert int(s["num_platforms"]) == matching and s["solved"] == 0
        for k in range(A):
            sx, sy, sz = (int(v) for v in s["agents"][k]["spawn"])
            assert (sx, sy, sz) == (0, 0, 0) or (sy == 2 and 2 <= sx < 18 and 2 <= sz < 13 and not (abs(sx - 5) < 2 and abs(sz - 5) < 2)
                                                 and not (abs(sx - 13) < 2 and abs(sz - 5) < 2))
    assert len(sizes) >= 4
    g.close()


def test_default_reward_shaping():
    g = oracle_lib.OracleGym("Rearrange", 16, 16, 1, 1)
    for k, v in {"teamSpirit": 0.0, "rearrangeOneMoreObjectCorrectPosition": 1.0, "rearrangeAllObjectsCorrectPosition": 10.0}.items():
        found = C.c_int(0)
        assert g.L.mvo_get_reward_shaping(g.g, 0, 0, k.encode(), C.byref(found)) == v and found.value == 1
    g.close()


def test_agents_settle_on_the_raised_floor_and_rollouts_are_reproducible():
    def run():
        n, A = 8, 2
        g = oracle_lib.OracleGym("Rearrange", 48, 32, n, A, 4)
        g.seed(5); g.reset()
        for st in range(30):   # no actions: everybody drops onto the raised floor (top at y = 1.5) or a pedestal step
            g.step_norender()
        ys = [float(g.snapshot(e)["agents"][a]["pos"][1]) for e in range(n) for a in range(A)]
        assert all(1.5 + 0.8 < y < 2.0 + 0.9 for y in ys), ys   # capsule bottom rests allowedCcdPenetration (0.04) inside the surface
        rew = []
        for st in range(950):
            m = action_masks(sample_actions(9, st, n * A))
            for e in range(n):
                for a in range(A):
                    g.set_action_mask(e, a, int(m[e * A + a]))
            g.step_norender()
            rew.append(g.get_last_rewards().copy())
        done = [g.is_done(e) for e in range(n)]
        g.render()
        f = np.stack([g.get_observation(e, 0) for e in range(n)])
        snaps = [g.snapshot(e).tobytes() for e in range(n)]
        g.close()
        return np.stack(rew), f, snaps, done
    r1, f1, s1, d1 = run()
    r2, f2, s2, d2 = run()
    assert np.array_equal(r1, r2) and np.array_equal(f1, f2) and s1 == s2 and d1 == d2
    assert f1[..., 3].min() == 255 and f1[..., :3].max() > 0
