"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Bar: bit-exact for every piece of discrete AND fp32 state, rewards, dones, and RGBA8 pixels
(the library is built with -ffp-contract=off so that this is achievable; DESIGN.md "numerics").
At BASELINE.json's full size the oracle is compared directly in tests/test_full_size_oracle_gpu.py (its physics costs ~5 ms per 1024-env tick);
test_full_size_properties_1024_envs below adds size-independent properties of the product alone."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib
from hip_util import diff_snapshots, hip_snapshot, make_pair, set_same_actions
from megaverse_amd.extension import MegaverseGym
from megaverse_amd.rollout import action_masks, sample_actions

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_fp32_ops_are_ieee_exact_on_device(hip):
    lib = hip.load_library()
    rng = np.random.default_rng(0)
    n = 1 << 16
    a = (rng.standard_normal(n) * 10 ** rng.uniform(-6, 6, n)).astype(np.float32)
    b = (rng.standard_normal(n) * 10 ** rng.uniform(-6, 6, n)).astype(np.float32)
    out = np.empty(n, np.float32)
    assert lib.mv_debug_math(0, 0, a.ctypes.data, b.ctypes.data, n, out.ctypes.data) == 0
    assert np.array_equal(out.view(np.uint32), (a / b).view(np.uint32)), "fp32 divide is not correctly rounded"
    aa = np.abs(a)
    lib.mv_debug_math(0, 1, aa.ctypes.data, None, n, out.ctypes.data)
    assert np.array_equal(out.view(np.uint32), np.sqrt(aa).view(np.uint32)), "sqrtf is not correctly rounded"
    lib.mv_debug_math(0, 3, a.ctypes.data, b.ctypes.data, n, out.ctypes.data)
    assert np.array_equal(out.view(np.uint32), ((a * b).astype(np.float32) + a).view(np.uint32)), "a*b+a was contracted to fma"
    x = rng.uniform(-7, 7, 4096).astype(np.float32)
    o2 = np.empty(2 * len(x), np.float32)
    lib.mv_debug_math(0, 2, x.ctypes.data, None, len(x), o2.ctypes.data)
    s, c = C.c_float(), C.c_float()
    for i in range(0, len(x), 5):
        oracle_lib.lib().mvo_sincos(float(x[i]), C.byref(s), C.byref(c))
        assert np.float32(s.value).tobytes() == o2[2 * i].tobytes() and np.float32(c.value).tobytes() == o2[2 * i + 1].tobytes()


@pytest.mark.parametrize("A,seed", [(1, 42), (2, 7), (4, 2024), (8, 1)])
def test_reset_parity(hip, A, seed):
    N = 48
    og, hg = make_pair(N, A, 32, 32, seed=seed)
    for e in range(N):
        d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
        assert not d, (e, d[:5])
    og.close(); hg.close()


def test_many_resets_parity(hip):
    """2 400 generated episodes against the oracle's.  The kernel computes the used prefix of the shuffled spawn list directly (mv_rng.h:
    shuffle_prefix_u16) and falls back to the full std::shuffle when one of the ~320 draws is not accepted at once (Lemire's rejection: about one
    episode in a hundred): both paths come up here, the oracle always runs the plain shuffle."""
    N, A = 600, 2
    og, hg = make_pair(N, A, 16, 16, seed=5)
    for rnd in range(4):
        if rnd:
            og.reset(); hg.reset()
        for e in range(N):
            d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
            assert not d, (rnd, e, d[:5])
    og.close(); hg.close()


@pytest.mark.parametrize("W,H", [(128, 128), (128, 72), (64, 64), (48, 20)])
def test_pixel_parity_after_reset_and_rollout(hip, W, H):
    N, A = 6, 2
    og, hg = make_pair(N, A, W, H, seed=11)
    def frames(g):
        return np.stack([g.get_observation(e, a) for e in range(N) for a in range(A)])
    assert np.array_equal(frames(og), frames(hg))
    for st in range(150):
        set_same_actions(og, hg, N, A, 5, st)
        og.step_norender(); hg.step_no_render()
    og.render(); hg.render()
    fo, fh = frames(og), frames(hg)
    assert fo.shape == (N * A, H, W, 4)
    assert np.array_equal(fo, fh), f"{int((fo != fh).sum())} differing bytes"
    assert fo[..., 3].min() == 255
    og.close(); hg.close()


def test_hires_render_parity(hip):
    og = oracle_lib.OracleGym("TowerBuilding", 768, 432, 1, 1, 1)
    hg = MegaverseGym("TowerBuilding", 128, 72, 1, 1, 1, False, {})
    og.seed(3); hg.seed(3); og.reset(); hg.reset()
    hg.draw_overview()
    hg.draw_hires()                                   # default 768x432, megaverse.cpp:261
    assert np.array_equal(og.get_observation(0, 0), hg.get_hires_observation(0, 0))
    og.close(); hg.close()


@pytest.mark.parametrize("A,N,steps", [(1, 24, 2500), (4, 8, 900)])
def test_rollout_parity(hip, A, N, steps):
    og, hg = make_pair(N, A, 32, 32, seed=42)
    events = 0
    for st in range(steps):
        set_same_actions(og, hg, N, A, 1234, st)
        og.step_norender(); hg.step_no_render()
        ro, rh = og.get_last_rewards(), hg.get_rewards_array()
        assert np.array_equal(ro.view(np.uint32), rh.view(np.uint32)), (st, ro, rh)
        events += int((ro != 0).sum())
        if st % 100 == 99 or st == steps - 1:
            assert [og.is_done(e) for e in range(N)] == hg.get_dones().astype(bool).tolist()
            for e in range(N):
                d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
                assert not d, (st, e, d[:5])
    assert events > 0, "the rollout never produced a reward: the interact path was not exercised"
    og.close(); hg.close()


@pytest.mark.parametrize("A,N,steps,redo", [(8, 16, 400, False), (4, 8, 300, True), (3, 8, 300, False)])
def test_multi_agent_controllers_shared_out(hip, monkeypatch, A, N, steps, redo):
    """Several agents per env: the kernel runs the controllers of agents that cannot meet on different waves, near ones in sequence, and checks
    afterwards (mv_tick_tower.h); the oracle runs them in a row.  Eight agents in one room: clusters all the time; MV_DEBUG_FORCE_REDO: the
    "check failed" path -- restore the agents, step them in a row -- on every tick."""
    if redo:
        monkeypatch.setenv("MV_DEBUG_FORCE_REDO", "1")
    og, hg = make_pair(N, A, 16, 16, seed=77)
    for st in range(steps):
        set_same_actions(og, hg, N, A, 4321, st)
        og.step_norender(); hg.step_no_render()
        ro, rh = og.get_last_rewards(), hg.get_rewards_array()
        assert np.array_equal(ro.view(np.uint32), rh.view(np.uint32)), (st, ro, rh)
        if st % 50 == 49 or st == steps - 1:
            for e in range(N):
                d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
                assert not d, (st, e, d[:5])
    og.close(); hg.close()


def test_auto_reset_parity_with_short_episodes(hip):
    # negative base episode length -> many envs finish on every tick: stresses done/true_objective/reset ordering
    N, A = 12, 2
    og, hg = make_pair(N, A, 32, 32, seed=5, params={"episodeLengthSec": -220.0})
    ndone = 0
    for st in range(300):
        set_same_actions(og, hg, N, A, 77, st)
        og.step(); hg.step()
        do = np.array([og.is_done(e) for e in range(N)])
        assert np.array_equal(do, hg.get_dones().astype(bool)), st
        ndone += int(do.sum())
        assert np.array_equal(og.get_last_rewards().view(np.uint32), hg.get_rewards_array().view(np.uint32))
        for e in np.nonzero(do)[0]:
            for a in range(A):
                assert og.true_objective(int(e), a) == hg.true_objective(int(e), a)
                assert hg.get_rewards_array()[e * A + a] == 0.0          # rewards on a done step are 0
        if st % 25 == 0:
            for e in range(N):
                d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
                assert not d, (st, e, d[:5])
                for a in range(A):
                    assert np.array_equal(og.get_observation(e, a), hg.get_observation(e, a))
    assert ndone > 100
    og.close(); hg.close()


@pytest.mark.parametrize("name", ["tower_a1", "tower_config0", "tower_a4", "tower_short_episodes", "obstacles_hard_a2", "obstacles_easy_a1", "collect_a2", "rearrange_a4",
                                  "sokoban_a2", "hex_memory_a2", "hex_explore_a3"])
def test_hip_reproduces_committed_golden(hip, name, monkeypatch):
    monkeypatch.setenv("BOXOBAN_LEVELS", os.path.join(GOLDEN, "boxoban"))
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    N, A, steps, every, W, H = (int(z[k]) for k in ("N", "A", "steps", "trace_every", "W", "H"))
    params = dict(zip(z["param_keys"].tolist(), [float(v) for v in z["param_vals"]])) if "param_keys" in z else {}
    hg = MegaverseGym(str(z["scenario"]) if "scenario" in z else "TowerBuilding", W, H, N, A, 1, False, params)
    hg.seed(int(z["seed"]))
    hg.reset()
    for e in range(N):
        s = hip_snapshot(hg, e)
        assert np.array_equal(s["objects"][: int(s["num_objects"])], z[f"reset_{e}_objects"])
        assert np.array_equal(s["boxes"][: int(s["num_boxes"])], z[f"reset_{e}_boxes"])
        if f"reset_{e}_hex_boxes" in z:
            assert s["hex_boxes"][: int(s["hex_num_boxes"])].tobytes() == z[f"reset_{e}_hex_boxes"].tobytes()
            assert s["hex_objs"][: int(s["hex_num_objs"])].tobytes() == z[f"reset_{e}_hex_objs"].tobytes()
    assert np.array_equal(np.stack([hg.get_observation(e, a) for e in range(min(N, 4)) for a in range(A)]), z["reset_obs"])
    trace = []
    for st in range(steps):
        hg.set_actions_batched(sample_actions(int(z["action_seed"]), st, N * A))
        hg.step_no_render()
        assert np.array_equal(hg.get_rewards_array().view(np.uint32), z["rewards"][st].view(np.uint32)), st
        assert np.array_equal(hg.get_dones(), z["dones"][st]), st
        if (st + 1) % every == 0:
            trace.append(np.stack([np.concatenate([np.asarray(hip_snapshot(hg, e)["agents"][a]["pos"]) for a in range(A)]) for e in range(N)]))
    assert np.array_equal(np.stack(trace).view(np.uint32), z["trace_pos"].view(np.uint32))
    hg.render()
    assert np.array_equal(np.stack([hg.get_observation(e, a) for e in range(min(N, 4)) for a in range(A)]), z["final_obs"])
    hg.close()


def test_device_random_policy_equals_host_stream(hip):
    # mv_sample_random_actions must write exactly megaverse_amd.rollout.sample_actions' stream
    N, A = 16, 2
    g1 = MegaverseGym("TowerBuilding", 16, 16, N, A, 1, False, {}); g1.seed(9); g1.reset()
    g2 = MegaverseGym("TowerBuilding", 16, 16, N, A, 1, False, {}); g2.seed(9); g2.reset()
    for st in range(300):
        g1.sample_random_actions(1234, st)
        g2.set_actions_batched(sample_actions(1234, st, N * A))
        g1.step_no_render(); g2.step_no_render()
    for e in range(N):
        assert not diff_snapshots(hip_snapshot(g1, e), hip_snapshot(g2, e), A)
    g1.close(); g2.close()


def test_per_agent_set_actions_equals_batched(hip):
    N, A = 3, 2
    g1 = MegaverseGym("TowerBuilding", 16, 16, N, A, 1, False, {}); g1.seed(4); g1.reset()
    g2 = MegaverseGym("TowerBuilding", 16, 16, N, A, 1, False, {}); g2.seed(4); g2.reset()
    for st in range(120):
        acts = sample_actions(3, st, N * A)
        for e in range(N):
            for a in range(A):
                g1.set_actions(e, a, acts[e * A + a].tolist())
        g2.set_actions_batched(acts)
        g1.step(); g2.step()
        if st % 7 == 0:      # a tick with no set_actions call is an idle tick (actions are cleared, env.cpp:141-142)
            g1.step(); g2.step()
    for e in range(N):
        assert not diff_snapshots(hip_snapshot(g1, e), hip_snapshot(g2, e), A)
    g1.close(); g2.close()


def test_sharded_gyms_equal_one_big_gym(hip):
    # env sharding with job-wide seeds (bench.py --gpus N, one process per GPU): the union of the
    # shards must be the single-process world, bit for bit
    total, A = 24, 1
    big = MegaverseGym("TowerBuilding", 32, 32, total, A, 1, False, {}); big.seed(42); big.reset()
    shards = [MegaverseGym("TowerBuilding", 32, 32, 8, A, 1, False, {}, env_offset=8 * r, total_envs=total) for r in range(3)]
    for s in shards:
        s.seed(42); s.reset()
    for st in range(200):
        big.sample_random_actions(1234, st); big.step()
        for s in shards:
            s.sample_random_actions(1234, st); s.step()
    for r, s in enumerate(shards):
        for e in range(8):
            assert not diff_snapshots(hip_snapshot(big, 8 * r + e), hip_snapshot(s, e), A)
            assert np.array_equal(big.get_observation(8 * r + e, 0), s.get_observation(e, 0))
    big.close()
    for s in shards:
        s.close()


def test_full_size_properties_1024_envs(hip):
    """BASELINE.json configs[1] size.  Properties that do not need the oracle: determinism, object
    table <-> voxel chunk consistency, agents stay inside the room, reward/done conventions."""
    N, A, W, H = 1024, 1, 128, 128
    def run():
        g = MegaverseGym("TowerBuilding", W, H, N, A, 1, False, {"episodeLengthSec": -100.0})
        g.seed(42); g.reset()
        rew, ndone = 0.0, 0
        for st in range(400):
            g.sample_random_actions(1234, st); g.step()
            if st % 40 == 0:
                d = g.get_dones(); r = g.get_rewards_array()
                assert np.all(r[d.astype(bool)] == 0.0)
                rew += float(r.sum()); ndone += int(d.sum())
        frames = np.stack([g.get_observation(e, 0) for e in (0, 1, 511, 1023)])
        snaps = [hip_snapshot(g, e).copy() for e in range(0, N, 16)]
        g.close()
        return rew, ndone, frames, snaps
    r1, d1, f1, s1 = run()
    r2, d2, f2, s2 = run()
    assert r1 == r2 and d1 == d2 and np.array_equal(f1, f2)
    assert all(a.tobytes() == b.tobytes() for a, b in zip(s1, s2)), "two identical runs diverged"
    assert d1 > 0 and f1[..., :3].max() > 0 and f1[..., 3].min() == 255
    for s in s1:
        L, Wd, n = int(s["L"]), int(s["W"]), int(s["num_objects"])
        p = s["agents"][0]["pos"]
        lim = 0.33 - 0.041 - 1e-3
        assert 1 + lim <= p[0] <= L - 1 - lim and 1 + lim <= p[2] <= Wd - 1 - lim and p[1] >= 1.8
        objs = s["objects"][:n]
        chunk = s["chunk"].reshape(16, 32, 32)
        placed = objs[objs[:, 3] == 0]
        assert int((chunk & 4 != 0).sum()) == len(placed)
        for o in placed:
            assert chunk[o[1], o[2], o[0]] & 4
        c = int(s["agents"][0]["carrying"])
        assert (c < 0 and (objs[:, 3] != 0).sum() == 0) or (objs[c, 3] == 1 and (objs[:, 3] != 0).sum() == 1)
