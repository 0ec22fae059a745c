"""Collect's episodes drawn ON THE DEVICE (megaverse_amd/csrc/mv_collect_draw.h, collect_draw_kernel; SURVEY.md 8 row f3's remainder): the kernel against the
host generator, record by record, and a gym whose feeder runs in device mode (MV_COLLECT_DEVICE_GEN=1) against the oracle -- the same tests the host-fed gym
passes: resets, rollouts with natural auto-resets, episodes of a few ticks (every env consuming a landscape per step), forced resets and a re-seed in
mid-run, batched calls with the host far ahead."""
import numpy as np
import pytest

from hip_util import diff_snapshots, hip_snapshot, make_pair, set_same_actions
from megaverse_amd import extension as ext
from test_collect_draw import assert_same_episode
from test_host_generators import COLLECT_BLOB, generate

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


def draw_device(agents, seeds, n, base_len=60.0):
    lib = ext.load_library()
    seeds = np.asarray(seeds, np.int32)
    out = np.zeros(len(seeds), COLLECT_BLOB)
    ms = np.zeros(1, np.float32)
    rc = lib.mv_debug_collect_draw_device(0, agents, seeds.ctypes.data, len(seeds), n, base_len, out.ctypes.data, out.nbytes, ms.ctypes.data)
    assert rc == 0, ext.last_error() if hasattr(ext, "last_error") else rc
    return out, float(ms[0])


@pytest.mark.parametrize("agents,n", [(1, 1), (2, 3), (8, 2)])
def test_kernel_draws_the_host_generators_episodes(hip, agents, n):
    rng = np.random.default_rng(99 + agents)
    seeds = [0, 1, 42, (1 << 30) - 1] + [int(s) for s in rng.integers(0, 1 << 30, 252)]
    got, ms = draw_device(agents, seeds, n)
    print(f"collect_draw_kernel: {len(seeds)} episodes per launch, {ms:.2f} ms per launch")
    for i, s in enumerate(seeds):
        assert got[i]["seq"] == n
        assert_same_episode(got[i], generate("Collect", agents, s, n), agents, (s, n))


def test_a_full_batch_of_episodes_in_one_launch(hip):
    # 1024 envs' worth in one launch (what a forced reset asks for): the rate the device generator sustains, and a sample of the records
    seeds = np.arange(1024, dtype=np.int32) * 7919 + 13
    got, ms = draw_device(1, seeds, 2)
    print(f"collect_draw_kernel: 1024 episodes per launch, {ms:.2f} ms per launch = {1024 / ms:.0f} k episodes/s")
    for i in range(0, 1024, 37):
        assert_same_episode(got[i], generate("Collect", 1, int(seeds[i]), 2), 1, int(seeds[i]))


@pytest.fixture
def device_gen(monkeypatch):
    monkeypatch.setenv("MV_COLLECT_DEVICE_GEN", "1")


@pytest.mark.parametrize("A,seed", [(1, 3), (5, 15)])
def test_gym_reset_parity(hip, device_gen, A, seed):
    N = 32
    og, hg = make_pair(N, A, 32, 32, seed=seed, scenario="Collect")
    for e in range(N):
        d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
        assert not d, (e, d[:5])
    assert np.array_equal(og.get_observation(3, 0), hg.get_observation(3, 0))
    og.close(); hg.close()


@pytest.mark.parametrize("base,steps,min_done", [(6.0, 900, 10), (-24.0, 160, 15), (-500.0, 100, 900)])
def test_gym_rollouts_with_auto_resets(hip, device_gen, base, steps, min_done):
    # default-length episodes (natural resets), episodes of a few ticks, and episodes that end on their first tick: every env consumes a landscape per step
    N, A = 10, 2
    og, hg = make_pair(N, A, 32, 32, seed=6, params={"episodeLengthSec": base}, scenario="Collect")
    ndone = 0
    for st in range(steps):
        set_same_actions(og, hg, N, A, 77, st)
        og.step_norender(); hg.step_no_render()
        do = np.array([og.is_done(e) for e in range(N)])
        assert np.array_equal(do, hg.get_dones().astype(bool)), st
        ndone += int(do.sum())
        assert np.array_equal(og.get_last_rewards().view(np.uint32), hg.get_rewards_array().view(np.uint32)), st
        if st % 20 == 0 or st == steps - 1 or do.any() and st % 7 == 0:
            for e in range(N):
                d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
                assert not d, (st, e, d[:5])
    og.render(); hg.render()
    for e in range(0, N, 3):
        assert np.array_equal(og.get_observation(e, 0), hg.get_observation(e, 0)), e
    assert ndone >= min_done, ndone
    og.close(); hg.close()


def test_gym_forced_resets_and_reseed(hip, device_gen):
    N, A = 8, 2
    og, hg = make_pair(N, A, 32, 32, seed=9, scenario="Collect")
    for rd in range(5):
        for st in range(3):
            set_same_actions(og, hg, N, A, 3, 10 * rd + st)
            og.step_norender(); hg.step_no_render()
        og.reset(); hg.reset()
        if rd == 2:
            og.seed(1234); hg.seed(1234)
            og.reset(); hg.reset()
        for e in range(N):
            assert not diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A), (rd, e)
    og.close(); hg.close()


def test_batched_calls_equal_the_host_fed_gym(hip, monkeypatch, recwarn):
    """The bench's launch shape (calls of 16 ticks, open loop, the host far ahead) with episodes of 100 ticks and more: the device-fed gym never starves and ends
    in the very state the host-fed gym ends in."""
    import torch
    from megaverse_amd.extension import MegaverseGym
    N, A, W, H, k, ticks = 64, 1, 32, 32, 16, 1600

    def run(device):
        monkeypatch.setenv("MV_COLLECT_DEVICE_GEN", "1" if device else "0")
        g = MegaverseGym("Collect", W, H, N, A, 8, False, {"episodeLengthSec": 70.0 / 15.0})
        g.set_pixel_mode("fast")
        ring = torch.zeros((2 * k, N * A, H, W, 4), dtype=torch.uint8, device="cuda:0")
        g.set_output_ring(2 * k, ring.data_ptr())
        g.seed(9); g.reset()
        for st in range(0, ticks, k):
            g.step_n(k, "multidiscrete", 31, st)
        g.synchronize(); torch.cuda.synchronize()
        snaps = [hip_snapshot(g, e).copy() for e in range(N)]
        last = ring[(ticks - 1) % (2 * k)].cpu().numpy().copy()
        g.close()
        return snaps, last

    a = run(True)
    assert not [w for w in recwarn.list if "capacity" in str(w.message) or "resident" in str(w.message)], [str(w.message) for w in recwarn.list]
    b = run(False)
    for e in range(N):
        d = diff_snapshots(a[0][e], b[0][e], A)
        assert not d, (e, d[:4])
    assert np.array_equal(a[1], b[1])


def test_a_group_with_a_device_fed_collect_gym_equals_the_host_fed_group(hip, monkeypatch):
    """configs[4]'s shape: Collect beside three other scenarios in one mv_group (union launches, one simulation stream for all members) with episodes short
    enough that every Collect env resets several times -- device-drawn against host-generated episodes: every env's state, rewards, dones and pixels."""
    import torch
    from megaverse_amd.multitask import MultiTaskGym
    names, N, A, W, H = ["TowerBuilding", "Collect", "ObstaclesEasy", "Collect"], 32, 1, 32, 32

    def run(device):
        monkeypatch.setenv("MV_COLLECT_DEVICE_GEN", "1" if device else "0")
        mt = MultiTaskGym(names, W, H, N, A, 2, {"episodeLengthSec": 3.0})
        mt.set_pixel_mode("fast")
        obs = mt.attach("cuda:0")
        mt.seed(11); mt.reset()
        assert [g.host_generator_threads() == 0 for g in mt.gyms] == [True, device, False, device]
        st = 0
        for _ in range(40):
            mt.sample_random_actions(9, st); mt.step(); st += 1
        for k in (8, 8, 5, 8) * 12:
            mt.step_n(k, "multidiscrete", 9, st); st += k
        mt.synchronize(); torch.cuda.synchronize()
        out = ([g.debug_snapshot_bytes(j).tobytes() for g in mt.gyms for j in range(N // len(names))], [g.get_rewards_array().tobytes() for g in mt.gyms],
               [g.get_dones().tobytes() for g in mt.gyms], obs.cpu().numpy().copy())
        mt.close()
        return out

    a, b = run(True), run(False)
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]
    assert np.array_equal(a[3], b[3]) and a[3][..., :3].max() > 0
