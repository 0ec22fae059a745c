"""CPU tests of the Collect restatement in oracle/ (SURVEY.md §8 row Y): the Perlin noise is pinned against the
reference's vendored perlin_noise.hpp compiled in place (oracle/_ref); the rest are spec-derived invariants of
scenario_collect.cpp."""
import numpy as np
import pytest

import oracle_lib
from megaverse_amd.rollout import action_masks, sample_actions


@pytest.mark.parametrize("seed,octaves", [(0, 1), (1, 3), (123456789, 9), (999999999, 5), (4294967295, 2)])
def test_perlin_matches_reference_header(seed, octaves):
    ref = oracle_lib.ref_lib()
    if ref is None or not hasattr(ref, "mvref_perlin_octave2_01"):
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(seed & 0xffff)
    n = 4000
    xs = rng.uniform(0, 12, n); ys = rng.uniform(0, 12, n)
    xs[:64] = np.arange(64) / (42 / 9.9); ys[:64] = np.arange(64)[::-1] / (42 / 0.1)   # the x / fx, z / fz grid of :86-88
    a, b = np.empty(n), np.empty(n)
    oracle_lib.lib().mvo_perlin_octave2_01(seed, xs.ctypes.data, ys.ctypes.data, n, octaves, a.ctypes.data)
    ref.mvref_perlin_octave2_01(seed, xs.ctypes.data, ys.ctypes.data, n, octaves, b.ctypes.data)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    assert a.min() >= 0.0 and a.max() <= 1.0


def _gym(n=24, agents=2, seed=5, **kw):
    g = oracle_lib.OracleGym("Collect", 32, 32, n, agents, 4, False, kw or None)
    g.seed(seed); g.reset()
    return g


def test_reset_invariants():
    n, A = 48, 3
    g = _gym(n, A, 11)
    for e in range(n):
        s = g.snapshot(e)
        L, W = int(s["L"]), int(s["W"])
        assert s["scenario"] == 2 and 8 <= L < 42 and 8 <= W < 42          # scenario_collect.cpp:65-66
        hm = s["heightmap"].reshape(42, 42)
        assert (hm[:L, :W] >= 0).all() and (hm[L:, :] == -1).all() and (hm[:, W:] == -1).all()
        assert (hm[0, :W] == 0).all() and (hm[L - 1, :W] == 0).all() and (hm[:L, 0] == 0).all() and (hm[:L, W - 1] == 0).all()
        assert hm.max() <= 14                                                # intensity < 18, noise - ground <= 0.8
        nb = int(s["num_boxes"])
        boxes = s["boxes"][:nb]
        vol = ((boxes[:, 3] - boxes[:, 0]) * (boxes[:, 4] - boxes[:, 1]) * (boxes[:, 5] - boxes[:, 2])).sum()
        assert vol == (hm[:L, :W].astype(int) + 1).sum()                     # the merged slabs tile the solid voxels exactly
        nr, no = int(s["num_rewards"]), int(s["num_objects"])
        assert 1 <= nr <= int(round(0.05 * W * L)) + 1 and no == max(3, int(L * W * 0.04))
        assert s["episode_len"] == np.float32(60.0 + 2.0 * nr)               # scenario_collect.hpp:55-59
        cells = set()
        for k in range(A):
            cells.add(tuple(int(v) for v in s["agents"][k]["spawn"]))
        for r in s["rewards"][:nr]:
            assert r[3] in (1, 2)
            cells.add((int(r[0]), int(r[1]), int(r[2])))
        for o in s["objects"][:no]:
            cells.add((int(o[0]), int(o[1]), int(o[2])))
        assert len(cells) == A + nr + no                                     # all drawn from one shuffled list
        for (x, y, z) in cells:
            assert 1 <= x < L - 1 and 1 <= z < W - 1 and y == max(1, hm[x, z] + 1)
        assert int(s["num_platforms"]) == int((s["rewards"][:nr, 3] == 1).sum())   # numPositiveRewards
        sh = g.L.mvo_get_reward_shaping
    g.close()


def test_default_reward_shaping():
    import ctypes as C
    g = _gym(1, 1)
    want = {"teamSpirit": 0.0, "collectSingleGood": 1.0, "collectSingleBad": -1.0, "collectAll": 5.0, "collectAbyss": -0.5}
    for k, v in want.items():
        found = C.c_int(0)
        assert g.L.mvo_get_reward_shaping(g.g, 0, 0, k.encode(), C.byref(found)) == v and found.value == 1
    g.close()


def test_same_seed_same_rollout_and_rewards_happen():
    def run(seed):
        n, A = 12, 2
        g = _gym(n, A, seed)
        tot, falls = [], 0
        for st in range(700):
            m = action_masks(sample_actions(3, st, n * A))
            for e in range(n):
                for a in range(A):
                    g.set_action_mask(e, a, int(m[e * A + a]))
            g.step_norender()
            tot.append(g.get_last_rewards().copy())
        g.render()
        f = np.stack([g.get_observation(e, 0) for e in range(n)])
        snaps = [g.snapshot(e).tobytes() for e in range(n)]
        g.close()
        return np.stack(tot), f, snaps
    r1, f1, s1 = run(9)
    r2, f2, s2 = run(9)
    assert np.array_equal(r1, r2) and np.array_equal(f1, f2) and s1 == s2
    assert (r1 > 0).any() and (r1 < 0).any()          # diamonds collected, bad diamonds / falls punished
    assert f1[..., 3].min() == 255 and f1[..., :3].max() > 0
