"""CPU: pins the oracle (test infrastructure) against everything the reference gives us for this path:
 * the reference's own RNG helpers compiled in place (oracle/_ref, util.hpp:25-56),
 * the C++ standard's mt19937 known answer,
 * the reference's own unit-test vector for getCoords (src/test/src/voxel_grid_tests.cpp:25),
 * spec-derived known answers (SURVEY.md 8c "fixtures the build must create itself"),
 * behavioural invariants of the restated controller, and the committed golden rollouts.
Physics trajectories and pixels are "parity unpinned" with respect to the real reference (Bullet /
Magnum are not vendored); those tests are restatement-relative and say so."""
import ctypes as C
import itertools
import os

import numpy as np
import pytest

import oracle_lib
from megaverse_amd.rollout import action_masks, sample_actions

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_mt19937_standard_known_answer():
    # [rand.predef]/3: the 10000th consecutive invocation of a default-constructed mt19937 produces 4123659995
    assert oracle_lib.lib().mvo_mt19937_nth(5489, 10000) == 4123659995


@pytest.fixture(scope="module")
def ref():
    r = oracle_lib.ref_lib()
    if r is None:
        pytest.skip("oracle/_ref not built (reference tree absent and no prebuilt .so)")
    return r


@pytest.mark.parametrize("seed", [0, 1, 42, 12345, 2**31 - 1, 4294967295])
def test_rand_range_matches_reference_util_hpp(ref, seed):
    rng = np.random.default_rng(seed)
    n = 5000
    lo = rng.integers(-50, 50, n).astype(np.int32)
    hi = (lo + rng.integers(1, 2000, n)).astype(np.int32)
    hi[:50] = lo[:50] + 1                       # range of one value still consumes a draw
    lo[50:60], hi[50:60] = 0, 1 << 30           # the seeding range (megaverse.cpp:66, env.cpp:61)
    a, b = np.empty(n, np.int32), np.empty(n, np.int32)
    oracle_lib.lib().mvo_rand_range_seq(seed, lo.ctypes.data, hi.ctypes.data, n, a.ctypes.data)
    ref.mvref_rand_range_seq(seed, lo.ctypes.data, hi.ctypes.data, n, b.ctypes.data)
    assert np.array_equal(a, b)
    assert np.all(a >= lo) and np.all(a < hi)


@pytest.mark.parametrize("seed", [0, 7, 42, 99999])
def test_frand_matches_reference_util_hpp(ref, seed):
    n = 20000
    a, b = np.empty(n, np.float32), np.empty(n, np.float32)
    oracle_lib.lib().mvo_frand_seq(seed, n, a.ctypes.data)
    ref.mvref_frand_seq(seed, n, b.ctypes.data)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert a.min() >= 0.0 and a.max() < 1.0


def test_env_seed_rule_matches_reference(ref):
    # MegaverseGym::seed: master mt19937(seed) -> randRange(0, 1<<30) per env (megaverse.cpp:60-69)
    n = 64
    lo, hi = np.zeros(n, np.int32), np.full(n, 1 << 30, np.int32)
    a, b = np.empty(n, np.int32), np.empty(n, np.int32)
    oracle_lib.lib().mvo_rand_range_seq(42, lo.ctypes.data, hi.ctypes.data, n, a.ctypes.data)
    ref.mvref_env_seeds(42, n, b.ctypes.data)
    assert np.array_equal(a, b)


def test_triangular_number_matches_reference_math_utils_hpp(ref):
    """boxes an Obstacles wall / step / gap needs (platforms.hpp requiresMovableBoxesToTraverse): the oracle's helper against the
    reference's own util/math_utils.hpp:7-10 compiled in place"""
    L = oracle_lib.lib()
    for n in range(0, 40):
        assert L.mvo_triangular_number(n) == ref.mvref_triangular_number(n) == n * (n + 1) // 2


def test_action_mask_table():
    # bindings/megaverse.cpp:100-116 + enum Action env.hpp:22-42
    L = oracle_lib.lib()
    def mask(a):
        arr = (C.c_int * 6)(*a)
        return L.mvo_action_mask(arr, 6)
    assert mask([0, 0, 0, 0, 0, 0]) == 0
    assert mask([1, 0, 0, 0, 0, 0]) == 1 << 1      # Left
    assert mask([2, 0, 0, 0, 0, 0]) == 1 << 2      # Right
    assert mask([0, 1, 0, 0, 0, 0]) == 1 << 3      # Forward
    assert mask([0, 2, 0, 0, 0, 0]) == 1 << 4      # Backward
    assert mask([0, 0, 1, 0, 0, 0]) == 1 << 5      # LookLeft
    assert mask([0, 0, 2, 0, 0, 0]) == 1 << 6      # LookRight
    assert mask([0, 0, 0, 1, 0, 0]) == 1 << 7      # Jump
    assert mask([0, 0, 0, 0, 1, 0]) == 1 << 8      # Interact
    assert mask([0, 0, 0, 0, 0, 1]) == 1 << 9      # LookDown
    assert mask([0, 0, 0, 0, 0, 2]) == 1 << 10     # LookUp
    combos = np.array(list(itertools.product(range(3), range(3), range(3), range(2), range(2), range(3))), np.int32)
    assert len(combos) == 324
    assert np.array_equal(action_masks(combos), np.array([mask(c) for c in combos], np.int32))
    assert np.all(action_masks(combos) & 1 == 0)   # bit 0 is never a valid action


def test_get_coords_reference_unit_test_vector():
    # src/test/src/voxel_grid_tests.cpp:25  getCoords({1.5,2.3,3.2}) == {1,2,3}
    v = np.array([1.5, 2.3, 3.2], np.float32)
    out = np.zeros(3, np.int32)
    oracle_lib.lib().mvo_get_coords(v.ctypes.data, out.ctypes.data)
    assert out.tolist() == [1, 2, 3]
    v = np.array([-0.25, -1.0, 0.999], np.float32)   # floor, not truncation (voxel_grid.hpp:18-21)
    oracle_lib.lib().mvo_get_coords(v.ctypes.data, out.ctypes.data)
    assert out.tolist() == [-1, -1, 0]


def test_building_reward_coefficients():
    # scenario_tower_building.cpp:246-251: 0.05 h + min(0.05 * 2^h, 20)
    for h in range(1, 11):
        want = np.float32(np.float32(h) * np.float32(0.05)) + np.float32(min(np.float32(0.05) * np.float32(2 ** h), np.float32(20.0)))
        got = oracle_lib.lib().mvo_building_reward_coeff(float(h))
        assert np.float32(got) == np.float32(want), (h, got, want)
    assert oracle_lib.lib().mvo_building_reward_coeff(9.0) == pytest.approx(0.45 + 20.0)


def test_sincos_polynomial_accuracy():
    L = oracle_lib.lib()
    s, c = C.c_float(), C.c_float()
    xs = np.linspace(-7, 7, 20001).astype(np.float32)
    err = 0.0
    for x in xs[::7]:
        L.mvo_sincos(float(x), C.byref(s), C.byref(c))
        err = max(err, abs(s.value - np.sin(np.float64(x))), abs(c.value - np.cos(np.float64(x))))
    assert err < 5e-7


def test_default_reward_shaping_and_isolation():
    # scenario_tower_building.hpp:44-52 ; megaverse/tests/test_env.py:123-140
    g = oracle_lib.OracleGym("TowerBuilding", 32, 32, 3, 2, 2)
    default = g.get_reward_shaping(0, 0)
    assert default == pytest.approx({"teamSpirit": 0.1, "towerPickedUpObject": 0.1, "towerVisitedBuildingZoneWithObject": 0.1,
                                     "towerBuildingReward": 1.0})
    g.set_reward_shaping(1, 1, {k: v * 3 for k, v in default.items()})
    assert g.get_reward_shaping(0, 0) == default and g.get_reward_shaping(1, 0) == default
    assert g.get_reward_shaping(1, 1) != default
    g.close()


@pytest.mark.parametrize("seed", [1, 42, 2024])
def test_generation_invariants(seed):
    # SURVEY.md appendix A.4 (scenario_tower_building.cpp:19-89, platforms.hpp:167-190)
    N, A = 24, 3
    g = oracle_lib.OracleGym("TowerBuilding", 16, 16, N, A, 1)
    g.seed(seed)
    g.reset()
    for e in range(N):
        s = g.snapshot(e)
        L, H, W = int(s["L"]), int(s["H"]), int(s["W"])
        assert 12 <= L < 30 and 12 <= W < 25 and H in (5, 6)
        bz = s["bz"]
        assert 1 <= bz[0] and bz[1] <= L - 1 and 1 <= bz[2] and bz[3] <= W - 1
        assert 3 <= bz[1] - bz[0] < 9 and 3 <= bz[3] - bz[2] < 9
        n = int(s["num_objects"])
        assert 4 <= n <= 73
        assert s["episode_len"] == np.float32(60.0 + 4.0 * n)           # :263-266
        objs = s["objects"][:n]
        assert len({tuple(o[:3]) for o in objs}) == n                    # one object per voxel
        assert np.all(objs[:, 1] >= 1) and np.all(objs[:, 1] <= 2) and np.all(objs[:, 3] == 0)
        chunk = s["chunk"].reshape(16, 32, 32)                           # [y][z][x]
        assert np.all(chunk[0, :W, :L] & 1)                              # floor + wall bottoms solid
        assert np.all(chunk[:H, 0, :L] & 1) and np.all(chunk[:H, W - 1, :L] & 1)
        assert np.all(chunk[:H, :W, 0] & 1) and np.all(chunk[:H, :W, L - 1] & 1)
        assert not np.any(chunk[1:, 1:W - 1, 1:L - 1] & 1)               # interior is air
        assert int((chunk & 4 != 0).sum()) == n
        for o in objs:
            assert chunk[o[1], o[2], o[0]] & 4
        spawns = [tuple(s["agents"][k]["spawn"]) for k in range(A)]
        assert len(set(spawns)) == A                                     # disjoint slices of one shuffled list (:41-59)
        for sp in spawns:
            assert 1 <= sp[0] <= L - 2 and sp[1] == 2 and 1 <= sp[2] <= W - 2
            assert (sp[0], 2, sp[2]) not in {tuple(o[:3]) for o in objs}
        boxes = s["boxes"][: int(s["num_boxes"])]
        assert int(s["num_boxes"]) == 5
        vol = sum(int(np.prod(b[3:6] - b[0:3])) for b in boxes)
        assert vol == int((chunk & 1 != 0).sum())                        # merged boxes tile the solid voxels exactly
    g.close()


def test_seed_determinism_first_observation():
    # megaverse/tests/test_env.py:42-55 test_seeds
    obs = []
    for _ in range(2):
        g = oracle_lib.OracleGym("TowerBuilding", 64, 36, 1, 1, 1)
        g.seed(42)
        g.reset()
        obs.append(g.get_observation(0, 0).copy())
        g.close()
    assert np.array_equal(obs[0], obs[1])
    g = oracle_lib.OracleGym("TowerBuilding", 64, 36, 1, 1, 1)
    g.seed(43)
    g.reset()
    assert not np.array_equal(obs[0], g.get_observation(0, 0))
    assert obs[0][..., 3].min() == 255 and obs[0][..., :3].max() > 0     # alpha 255, something is visible
    g.close()


def _fresh(seed=42, A=1, params=None, N=1):
    g = oracle_lib.OracleGym("TowerBuilding", 16, 16, N, A, 1, False, params)
    g.seed(seed)
    g.reset()
    return g


def test_free_fall_lands_and_rests():
    # spawn origin = voxel + (0.5, 1.75, 0.5) at y=2 (agent.cpp:42-46); resting height on the y=1
    # floor top is centre = 1 + 0.525 + 0.33 - 0.04 (allowed CCD penetration), never deeper than 0.041
    g = _fresh()
    ys = []
    for _ in range(40):
        g.step_norender()
        ys.append(float(g.snapshot(0)["agents"][0]["pos"][1]))
    assert ys[0] < 3.75 and max(abs(ys[-1] - ys[-2]), abs(ys[-2] - ys[-3])) < 1e-5   # came to rest (last-ulp jitter of step up/down)
    rest = ys[-1]
    on_floor, on_box = abs(rest - 1.815) < 2e-3, abs(rest - (1.5 - 0.05 + 0.4485 + 0.855 - 0.04)) < 2e-3
    assert on_floor or on_box, rest
    a = g.snapshot(0)["agents"][0]
    assert a["vvel"] == 0 and a["voffset"] == 0 and np.all(a["hv"] == 0)
    g.close()


def test_jump_apex_and_max_walk_speed():
    # jump impulse 6.2 (agent.cpp:160), gravity 13.72 (.hpp:169): apex ~ 6.2^2 / (2*13.72) = 1.40 at dt=1/15
    g = _fresh()
    for _ in range(40):
        g.step_norender()
    y0 = float(g.snapshot(0)["agents"][0]["pos"][1])
    g.set_action_mask(0, 0, 1 << 7)
    peak = y0
    for _ in range(20):
        g.step_norender()
        peak = max(peak, float(g.snapshot(0)["agents"][0]["pos"][1]))
    assert 1.0 < peak - y0 < 1.45, peak - y0
    # walking: speed saturates at 4.5 (kinematic_character_controller.hpp:173) until a wall stops it
    speeds = []
    for _ in range(12):
        g.set_action_mask(0, 0, 1 << 3)
        g.step_norender()
        speeds.append(float(np.hypot(*g.snapshot(0)["agents"][0]["hv"])))
    assert max(speeds) <= 4.5 + 1e-4
    g.close()


def test_rollout_invariants_never_inside_solids():
    N, A = 8, 2
    g = _fresh(seed=7, A=A, N=N)
    for st in range(600):
        masks = action_masks(sample_actions(99, st, N * A))
        for e in range(N):
            for a in range(A):
                g.set_action_mask(e, a, int(masks[e * A + a]))
        g.step_norender()
        if st % 20 == 0:
            for e in range(N):
                s = g.snapshot(e)
                L, W = int(s["L"]), int(s["W"])
                for a in range(A):
                    p = s["agents"][a]["pos"]
                    lim = 0.33 - 0.041 - 1e-3                            # capsule radius minus tolerated penetration
                    assert 1 + lim <= p[0] <= L - 1 - lim and 1 + lim <= p[2] <= W - 1 - lim, (st, e, a, p)
                    assert p[1] >= 1 + 0.525 + 0.33 - 0.041 - 1e-3, (st, e, a, p)
                    assert float(np.hypot(*s["agents"][a]["hv"])) <= 4.5 * 1.5
                carried = [int(s["agents"][a]["carrying"]) for a in range(A) if s["agents"][a]["carrying"] >= 0]
                assert len(set(carried)) == len(carried)
                for a in range(A):
                    c = int(s["agents"][a]["carrying"])
                    if c >= 0:
                        assert s["objects"][c][3] == 1 + a
                n = int(s["num_objects"])
                placed = s["objects"][:n][s["objects"][:n, 3] == 0]
                chunk = s["chunk"].reshape(16, 32, 32)
                assert int((chunk & 4 != 0).sum()) == len(placed)
    g.close()


def test_done_step_semantics():
    # vector_env.cpp:93-105 + SURVEY appendix A.1: on the done step the env is already reset, the
    # reported reward is 0, true_objective was captured before the reset
    g = _fresh(params={"episodeLengthSec": -1000.0})                     # episode length < 0: done on every step
    for _ in range(5):
        g.step_norender()
        assert g.is_done(0)
        s = g.snapshot(0)
        assert s["num_frames"] == 0 and s["episode_sec"] == 0 and s["done"] == 0
        assert g.get_last_rewards()[0] == 0.0
        assert g.true_objective(0, 0) == 0.0
    g.close()


@pytest.mark.parametrize("name", ["tower_a1", "tower_config0", "tower_a4", "tower_short_episodes", "obstacles_hard_a2", "obstacles_easy_a1", "collect_a2", "rearrange_a4"])
def test_oracle_reproduces_golden(name):
    """restatement-relative: the committed vectors were generated by this oracle (tests/golden/make_golden.py)"""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    N, A, steps, every, W, H = (int(z[k]) for k in ("N", "A", "steps", "trace_every", "W", "H"))
    params = dict(zip(z["param_keys"].tolist(), z["param_vals"].tolist())) if "param_keys" in z else None
    g = oracle_lib.OracleGym(str(z["scenario"]) if "scenario" in z else "TowerBuilding", W, H, N, A, 1, False, params)
    g.seed(int(z["seed"]))
    g.reset()
    for e in range(N):
        s = g.snapshot(e)
        assert np.array_equal(s["objects"][: int(s["num_objects"])], z[f"reset_{e}_objects"])
        assert np.array_equal(s["boxes"][: int(s["num_boxes"])], z[f"reset_{e}_boxes"])
        assert s["episode_len"] == z[f"reset_{e}_episode_len"]
    assert np.array_equal(np.stack([g.get_observation(e, a) for e in range(min(N, 4)) for a in range(A)]), z["reset_obs"])
    trace = []
    for st in range(steps):
        masks = action_masks(sample_actions(int(z["action_seed"]), st, N * A))
        for e in range(N):
            for a in range(A):
                g.set_action_mask(e, a, int(masks[e * A + a]))
        g.step_norender()
        assert np.array_equal(g.get_last_rewards().view(np.uint32), z["rewards"][st].view(np.uint32)), st
        assert [int(g.is_done(e)) for e in range(N)] == z["dones"][st].tolist(), st
        if (st + 1) % every == 0:
            trace.append(np.stack([np.concatenate([np.asarray(g.snapshot(e)["agents"][a]["pos"]) for a in range(A)]) for e in range(N)]))
    assert np.array_equal(np.stack(trace).view(np.uint32), z["trace_pos"].view(np.uint32))
    g.render()
    assert np.array_equal(np.stack([g.get_observation(e, a) for e in range(min(N, 4)) for a in range(A)]), z["final_obs"])
    g.close()


def test_level_file_tokeniser_matches_the_reference_split_string():
    """SokobanScenario::reloadLevels splits a level file with splitString(content, "\\n") (scenario_sokoban.cpp:90,
    util/src/string_utils.cpp:10-25: strtok_r, empty pieces never appear); the oracle's split_tokens against the reference function
    compiled in place"""
    import ctypes as C
    ref = oracle_lib.ref_lib()
    if ref is None or not hasattr(ref, "mvref_split_string"):
        pytest.skip("oracle/_ref/libmv_ref_util.so is built from /root/reference (not present here)")
    ref.mvref_split_string.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    mine = oracle_lib.lib().mvo_split_tokens
    mine.argtypes = [C.c_char_p, C.c_char, C.c_char_p, C.c_int]
    texts = ["; 0\n##########\n#  @ $ . #\n##########\n\n; 1\n####\n#@$.#\n", "\n\n\nabc\n\n\ndef\n", "no newline at all", "", "\n", "a\r\nb\r\n",
             "; 3\n" + "#" * 40 + "\n" * 7 + "tail"]
    for t in texts:
        a, b = C.create_string_buffer(4096), C.create_string_buffer(4096)
        na = ref.mvref_split_string(t.encode(), b"\n", a, 4096)
        nb = mine(t.encode(), b"\n", b, 4096)
        assert na == nb and a.value == b.value, (t, na, nb, a.value, b.value)
