"""One-step-ahead pipelining: the step kernels run on an internal stream and step t + 1 may overlap the observation pass
of step t.  These tests drive the gym OPEN LOOP -- device-sampled actions, no host synchronisation between steps -- so that the overlap
really happens, and check that nothing observable changed: the simulation is bit-identical to the oracle's, and what a consumer
enqueued on the caller's stream between two steps reads (rewards, dones, true objectives, observations) is exactly step t's output."""
import numpy as np
import pytest

import oracle_lib
from hip_util import diff_snapshots, hip_snapshot
from megaverse_amd.extension import MegaverseGym
from megaverse_amd.rollout import action_masks, sample_actions

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


class DevArr:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def oracle_step(og, N, A, seed, st, render=False):
    masks = action_masks(sample_actions(seed, st, N * A))
    for e in range(N):
        for a in range(A):
            og.set_action_mask(e, a, int(masks[e * A + a]))
    og.step() if render else og.step_norender()


@pytest.mark.parametrize("scenario,A,params", [("TowerBuilding", 2, {"episodeLengthSec": -220.0}), ("Rearrange", 1, {"episodeLengthSec": 2.0}),
                                              ("Collect", 3, {"episodeLengthSec": 5.0}), ("HexMemory", 2, {"episodeLengthSec": 3.0})])
def test_open_loop_rollout_equals_the_oracle(hip, scenario, A, params):
    """400 steps without a host sync, short episodes (auto-resets, refills from the feeder) -- then state, last outputs and pixels"""
    import torch
    N, W, H, STEPS = 12, 48, 32, 400
    og = oracle_lib.OracleGym(scenario, W, H, N, A, 1, False, params)
    hg = MegaverseGym(scenario, W, H, N, A, 2, False, params)
    hg.set_pixel_mode("fast")
    og.seed(5); hg.seed(5); og.reset(); hg.reset()
    for st in range(STEPS):
        hg.sample_random_actions(77, st)
        hg.step()
    for st in range(STEPS):
        oracle_step(og, N, A, 77, st, render=(st == STEPS - 1))
    hg.synchronize()
    for e in range(N):
        d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
        assert not d, (e, d[:5])
    assert og.get_last_rewards().tobytes() == hg.get_rewards_array().tobytes()
    assert np.array_equal(np.array([og.is_done(e) for e in range(N)]), hg.get_dones().astype(bool))
    fo = np.stack([og.get_observation(e, a) for e in range(N) for a in range(A)])
    fh = np.stack([hg.get_observation(e, a) for e in range(N) for a in range(A)])
    diff = np.abs(fo.astype(np.int16) - fh.astype(np.int16)).max(axis=-1)
    assert (diff > 1).sum() <= max(2, 1e-4 * diff.size) and (diff > 0).sum() <= max(4, 5e-4 * diff.size)
    og.close(); hg.close()


def test_consumers_on_the_callers_stream_see_step_t_while_step_t_plus_1_runs(hip):
    """every step: enqueue copies of the device-side rewards / dones / true objectives / observations on the caller's stream, step again
    at once; afterwards every copy must be that step's value (rewards / dones bit-exact against the oracle, observations against a
    synchronous replay of the same gym)"""
    import torch
    scenario, N, A, W, H, STEPS = "TowerBuilding", 64, 2, 64, 64, 150
    params = {"episodeLengthSec": -220.0}   # TowerBuilding adds time per object: a few seconds in all

    def replay(sync_every_step):
        hg = MegaverseGym(scenario, W, H, N, A, 2, False, params)
        hg.set_pixel_mode("fast")
        stream = torch.cuda.Stream()
        hg.set_stream(stream.cuda_stream)
        hg.seed(9); hg.reset()
        rew = torch.as_tensor(DevArr(hg.rewards_device_ptr(), (N * A,), "<f4"), device="cuda:0")
        done = torch.as_tensor(DevArr(hg.dones_device_ptr(), (N,), "|u1"), device="cuda:0")
        tru = torch.as_tensor(DevArr(hg.true_objectives_device_ptr(), (N * A,), "<f4"), device="cuda:0")
        obs = torch.as_tensor(DevArr(hg.obs_device_ptr(), (N * A, H, W, 4), "|u1"), device="cuda:0")
        out = []
        with torch.cuda.stream(stream):
            for st in range(STEPS):
                hg.sample_random_actions(31, st)
                hg.step()
                out.append((rew.clone(), done.clone(), tru.clone(), obs.sum(dtype=torch.int64), obs[st % (N * A)].clone()))
                if sync_every_step:
                    hg.synchronize()
        hg.synchronize(); torch.cuda.synchronize()
        res = [tuple(t.cpu().numpy() for t in o) for o in out]
        hg.close()
        return res

    piped, serial = replay(False), replay(True)
    og = oracle_lib.OracleGym(scenario, W, H, N, A, 1, False, params)
    og.seed(9); og.reset()
    dones = 0
    for st in range(STEPS):
        oracle_step(og, N, A, 31, st)
        r, d, t, osum, oframe = piped[st]
        assert og.get_last_rewards().tobytes() == r.tobytes(), st
        do = np.array([og.is_done(e) for e in range(N)])
        assert np.array_equal(do, d.astype(bool)), st
        dones += int(do.sum())
        want_t = np.array([og.true_objective(e, a) for e in range(N) for a in range(A)], np.float32)
        assert want_t.tobytes() == t.tobytes(), st
        assert int(osum) == int(serial[st][3]) and np.array_equal(oframe, serial[st][4]), st
    assert dones > N          # auto-resets happened while pipelined
    og.close()


def test_switching_pixel_mode_and_pipelining_mid_run_changes_nothing(hip):
    """exact steps run on the caller's stream, fast ones pipelined on the simulation stream: switching back and forth (and turning the
    pipelining off and on) without a host sync in between must leave the simulation on the oracle's trajectory"""
    N, A, W, H = 16, 2, 48, 32
    og = oracle_lib.OracleGym("TowerBuilding", W, H, N, A, 1, False, {"episodeLengthSec": -220.0})
    hg = MegaverseGym("TowerBuilding", W, H, N, A, 2, False, {"episodeLengthSec": -220.0})
    og.seed(3); hg.seed(3); og.reset(); hg.reset()
    st = 0
    for phase in range(12):
        mode = ("fast", "exact", "fast", "fast")[phase % 4]
        hg.set_pixel_mode(mode)
        hg.set_pipelining(phase % 3 != 2)
        for _ in range(17):
            hg.sample_random_actions(5, st)
            hg.step() if (st % 3) else hg.step_no_render()
            oracle_step(og, N, A, 5, st)
            st += 1
    hg.synchronize()
    for e in range(N):
        d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
        assert not d, (e, d[:5])
    assert og.get_last_rewards().tobytes() == hg.get_rewards_array().tobytes()
    og.close(); hg.close()


def test_a_policy_in_the_loop_is_a_true_dependency(hip):
    """actions computed ON THE DEVICE from the observations of the previous tick (torch ops on the caller's stream) and handed over with
    set_actions_device: the next step must wait for them -- without any host synchronisation the rollout has to equal one that
    synchronises after every call"""
    import torch
    scenario, N, A, W, H, STEPS = "ObstaclesEasy", 32, 2, 32, 32, 120
    spaces = torch.tensor([3, 3, 3, 2, 2, 3], dtype=torch.int32, device="cuda:0")

    def rollout(sync):
        hg = MegaverseGym(scenario, W, H, N, A, 2, False, {})
        hg.set_pixel_mode("fast")
        stream = torch.cuda.Stream()
        hg.set_stream(stream.cuda_stream)
        hg.seed(17); hg.reset()
        obs = torch.as_tensor(DevArr(hg.obs_device_ptr(), (N * A, H, W, 4), "|u1"), device="cuda:0")
        keep = []
        with torch.cuda.stream(stream):
            for st in range(STEPS):
                feat = obs.view(N * A, -1)[:, 37:37 + 6 * 97:97].to(torch.int32) + st   # six bytes of every frame
                acts = (feat % spaces).contiguous()
                keep.append(acts)                                                         # (alive until the kernels that read it ran)
                hg.set_actions_device(acts.data_ptr())
                hg.step()
                if sync:
                    hg.synchronize(); torch.cuda.synchronize()
        hg.synchronize(); torch.cuda.synchronize()
        snaps = [hip_snapshot(hg, e).copy() for e in range(N)]
        frames = obs.cpu().numpy().copy()
        acts_sum = int(torch.stack(keep).sum().item())
        hg.close()
        return snaps, frames, acts_sum

    a, b = rollout(False), rollout(True)
    assert a[2] == b[2]
    assert all(x.tobytes() == y.tobytes() for x, y in zip(a[0], b[0]))
    assert np.array_equal(a[1], b[1])
    moved = sum(float(np.abs(s["agents"]["hv"][:A]).sum()) > 0 for s in a[0])
    assert moved > 0          # the policy's actions did something


def _boxoban(monkeypatch):
    import os
    monkeypatch.setenv("BOXOBAN_LEVELS", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "boxoban"))


@pytest.mark.parametrize("scenario,A,params", [("TowerBuilding", 1, {"episodeLengthSec": -200.0}), ("TowerBuilding", 3, {}), ("ObstaclesHard", 2, {}),
                                              ("Collect", 1, {"episodeLengthSec": 6.0}), ("Sokoban", 2, {}), ("HexExplore", 1, {}),
                                              ("Rearrange", 2, {"episodeLengthSec": 0.3})])
def test_step_n_equals_k_single_steps(hip, monkeypatch, scenario, A, params):
    """mv_step_n(k) = k x (mv_sample_random_actions + mv_step): same ticks, same order, one queue hand-over per call instead of per tick.
    Chunks of every size up to the batch (and beyond: the call splits them), natural auto-resets, no host sync in between; then state,
    outputs and the whole observation slab must be equal byte for byte.  (Rearrange with 0.3 s episodes: the refill protocol reads the
    consumed counts after every tick, and mv_step_n falls back to one tick per hand-over.)"""
    _boxoban(monkeypatch)
    N, W, H = 24, 64, 48
    chunks = [1, 8, 3, 5, 8, 8, 2, 16, 7, 8, 19, 8, 1, 6]
    total = sum(chunks)

    def make():
        g = MegaverseGym(scenario, W, H, N, A, 2, False, params)
        g.set_pixel_mode("fast"); g.seed(21); g.reset()
        return g

    a, b = make(), make()
    for st in range(total):
        a.sample_random_actions(99, st); a.step()
    st = 0
    for k in chunks:
        b.step_n(k, "multidiscrete", 99, st)
        st += k
    a.synchronize(); b.synchronize()
    for e in range(N):
        assert a.debug_snapshot_bytes(e).tobytes() == b.debug_snapshot_bytes(e).tobytes(), e
    assert a.get_rewards_array().tobytes() == b.get_rewards_array().tobytes()
    assert np.array_equal(a.get_dones(), b.get_dones())
    assert a.get_true_objectives().tobytes() == b.get_true_objectives().tobytes()
    fa = np.stack([a.get_observation(e, k) for e in range(N) for k in range(A)])
    fb = np.stack([b.get_observation(e, k) for e in range(N) for k in range(A)])
    assert fa[..., :3].max() > 0 and np.array_equal(fa, fb)
    a.close(); b.close()


@pytest.mark.parametrize("pipe", ["0", "1"])
@pytest.mark.parametrize("scenario,params", [("TowerBuilding", {"episodeLengthSec": -200.0}), ("ObstaclesHard", {}), ("ObstaclesEasy", {"episodeLengthSec": -250.0}),
                                             ("Empty", {})])
def test_both_resident_step_kernels_equal_single_steps(hip, monkeypatch, scenario, params, pipe):
    """The resident multi-tick step launch has two flavours -- one wave per env, or two (wave 0 ticks while wave 1 sets the previous tick's frame up:
    step_ticks_pipe_kernel / step_obstacles_ticks_pipe_kernel) -- chosen by env count (mv_step.hip: step_pipe_enabled); MV_STEP_PIPE forces one.  Either must
    leave what single ticks leave: state, outputs, every frame of every ring entry of the last call."""
    import torch
    monkeypatch.setenv("MV_STEP_PIPE", pipe)
    N, W, H, K = 40, 64, 64, 8
    obs = torch.zeros((K, N, H, W, 4), dtype=torch.uint8, device="cuda:0")

    def make():
        g = MegaverseGym(scenario, W, H, N, 1, 2, False, params)
        g.set_pixel_mode("fast"); g.seed(31); g.reset()
        return g

    a, b = make(), make()
    b.set_output_ring(K, obs.data_ptr())
    calls = 9
    frames = []
    for st in range(calls * K):
        a.sample_random_actions(77, st); a.step()
        if st >= (calls - 1) * K:
            frames.append(np.stack([a.get_observation(e, 0) for e in range(N)]))
    for c in range(calls):
        b.step_n(K, "multidiscrete", 77, c * K)
    a.synchronize(); b.synchronize(); torch.cuda.synchronize()
    for e in range(N):
        assert a.debug_snapshot_bytes(e).tobytes() == b.debug_snapshot_bytes(e).tobytes(), e
    assert a.get_rewards_array().tobytes() == b.get_rewards_array().tobytes()
    assert np.array_equal(a.get_dones(), b.get_dones())
    ring = obs.cpu().numpy()
    for j in range(K):
        assert np.array_equal(ring[j], frames[j]), j
    assert ring[..., :3].max() > 0
    a.close(); b.close()


def test_step_n_against_the_oracle(hip):
    """the batched call against the CPU oracle directly (not only against the single-step path): state, rewards, dones after 200 ticks with resets"""
    scenario, N, A, W, H = "TowerBuilding", 16, 2, 48, 32
    params = {"episodeLengthSec": -220.0}
    og = oracle_lib.OracleGym(scenario, W, H, N, A, 1, False, params)
    hg = MegaverseGym(scenario, W, H, N, A, 2, False, params)
    og.seed(4); hg.seed(4); og.reset(); hg.reset()
    for first in range(0, 200, 8):
        hg.step_n(8, "multidiscrete", 55, first)
    for st in range(200):
        oracle_step(og, N, A, 55, st, render=(st == 199))
    hg.synchronize()
    for e in range(N):
        d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
        assert not d, (e, d[:5])
    assert og.get_last_rewards().tobytes() == hg.get_rewards_array().tobytes()
    assert np.array_equal(np.array([og.is_done(e) for e in range(N)]), hg.get_dones().astype(bool))
    for e in range(N):   # (new gyms of this test session render with the exact pixel arithmetic: bit for bit)
        for a in range(A):
            assert np.array_equal(og.get_observation(e, a), hg.get_observation(e, a))
    og.close(); hg.close()


def test_output_ring_keeps_every_tick_of_a_batch(hip):
    """mv_set_output_ring: tick t leaves observations / rewards / dones in entry t % count -- a k-step rollout buffer filled by one call.
    Every entry must equal what a gym stepped tick by tick reported at that tick."""
    import torch
    scenario, N, A, W, H, K = "ObstaclesEasy", 16, 2, 40, 24, 8
    obs = torch.zeros((K, N * A, H, W, 4), dtype=torch.uint8, device="cuda:0")
    rew = torch.full((K, N * A), -7.0, dtype=torch.float32, device="cuda:0")
    don = torch.full((K, N), 9, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()

    def make():
        g = MegaverseGym(scenario, W, H, N, A, 2, False, {})
        g.set_pixel_mode("fast"); g.seed(8); g.reset()
        return g

    ref, ring = make(), make()
    ring.set_output_ring(K, obs.data_ptr(), rew.data_ptr(), don.data_ptr())
    want = []
    for st in range(3 * K):
        ref.sample_random_actions(12, st); ref.step()
        want.append((np.stack([ref.get_observation(e, a) for e in range(N) for a in range(A)]), ref.get_rewards_array().copy(), ref.get_dones().copy()))
    for first in range(0, 3 * K, K):
        ring.step_n(K, "multidiscrete", 12, first)
        ring.synchronize(); torch.cuda.synchronize()
        o, r, d = obs.cpu().numpy(), rew.cpu().numpy(), don.cpu().numpy()
        for j in range(K):
            wo, wr, wd = want[first + j]
            assert np.array_equal(o[j], wo), (first, j)
            assert wr.tobytes() == r[j].tobytes() and np.array_equal(wd, d[j]), (first, j)
    # the host getters read the entry of the last tick
    assert np.array_equal(ring.get_observation(3, 1), want[-1][0][3 * A + 1])
    assert ring.get_rewards_array().tobytes() == want[-1][1].tobytes()
    # back to the single slab
    ring.set_output_ring(0)
    ring.sample_random_actions(12, 3 * K); ring.step()
    ref.sample_random_actions(12, 3 * K); ref.step()
    assert np.array_equal(ring.get_observation(0, 0), ref.get_observation(0, 0))
    ref.close(); ring.close()


@pytest.mark.parametrize("batched", [False, True])
def test_single_bit_policy_equals_host_stream(hip, batched):
    """the reference's own benchmark policy, Action(1 << randRange(0, NumActions)) (megaverse_test_app.cpp:140-147), drawn inside the step
    kernel == the host twin (rollout.sample_single_bit_masks) fed to the oracle mask by mask"""
    from megaverse_amd.rollout import sample_single_bit_masks
    scenario, N, A, W, H, STEPS = "TowerBuilding", 12, 2, 32, 32, 160
    og = oracle_lib.OracleGym(scenario, W, H, N, A, 1, False, {})
    hg = MegaverseGym(scenario, W, H, N, A, 2, False, {})
    og.seed(2); hg.seed(2); og.reset(); hg.reset()
    hg.set_sample_policy("single-bit")
    seen = set()
    for st in range(STEPS):
        masks = sample_single_bit_masks(41, st, N * A)
        seen.update(int(m) for m in masks)
        for e in range(N):
            for a in range(A):
                og.set_action_mask(e, a, int(masks[e * A + a]))
        og.step_norender()
    if batched:
        for first in range(0, STEPS, 8):
            hg.step_n(8, "single-bit", 41, first)
    else:
        for st in range(STEPS):
            hg.sample_random_actions(41, st); hg.step_no_render()
    assert seen == {1 << b for b in range(11)}
    for e in range(N):
        d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
        assert not d, (e, d[:5])
    og.close(); hg.close()


def test_a_pending_device_action_buffer_is_read_by_what_comes_next(hip):
    """mv_set_actions_device keeps the caller's buffer until the next step kernel reads it.  A host-side setter or a reset in between
    must read it AT ONCE -- while it is certainly alive -- and the last writer wins: device actions, then host actions for every agent => the
    host's; device actions, then the buffer is overwritten after a reset => what it held when the reset came."""
    import torch
    from megaverse_amd.rollout import sample_actions
    N, A = 16, 2
    def gym():
        g = MegaverseGym("TowerBuilding", 32, 32, N, A, 1, False, {})
        g.seed(4); g.reset()
        return g
    a1 = sample_actions(1, 0, N * A); a2 = sample_actions(2, 0, N * A)
    # (1) device buffer, then host actions for everyone: the step uses the host's
    g, ref = gym(), gym()
    buf = torch.as_tensor(a1, device="cuda:0").contiguous()
    g.set_actions_device(buf.data_ptr())
    g.set_actions_batched(a2)
    buf.zero_()
    ref.set_actions_batched(a2)
    g.step(); ref.step()
    for e in range(N):
        assert hip_snapshot(g, e).tobytes() == hip_snapshot(ref, e).tobytes(), e
    g.close(); ref.close()
    # (2) device buffer, then ONE host-side per-agent setter: the buffer is converted at that moment (it may be gone afterwards), the host's upload
    # of the whole action array at the next step then wins as the last writer -- exactly what a host-only caller of set_actions gets
    g, ref = gym(), gym()
    buf = torch.as_tensor(a1, device="cuda:0").contiguous()
    g.set_actions_device(buf.data_ptr())
    g.set_actions(3, 1, [1, 0, 2, 0, 1, 0])
    buf.fill_(7)   # (garbage: must no longer be looked at)
    torch.cuda.synchronize()
    ref.set_actions(3, 1, [1, 0, 2, 0, 1, 0])
    g.step(); ref.step()
    for e in range(N):
        assert hip_snapshot(g, e).tobytes() == hip_snapshot(ref, e).tobytes(), e
    g.close(); ref.close()


@pytest.mark.parametrize("scenario,N,A,W,H,params", [("TowerBuilding", 40, 1, 48, 20, {"episodeLengthSec": -200.0}), ("TowerBuilding", 33, 3, 128, 72, {}),
                                                     ("TowerBuilding", 100, 1, 64, 64, {}), ("ObstaclesEasy", 800, 1, 33, 17, {}), ("Empty", 770, 1, 128, 128, {}),
                                                     ("ObstaclesHard", 96, 1, 128, 128, {}), ("Rearrange", 70, 1, 128, 128, {}), ("Rearrange", 130, 1, 50, 30, {}),
                                                     ("Sokoban", 90, 1, 128, 128, {}), ("Collect", 60, 1, 128, 128, {}), ("Collect", 140, 1, 64, 64, {}),
                                                     ("HexMemory", 48, 1, 128, 128, {}), ("HexExplore", 72, 1, 64, 64, {})])
def test_one_launch_calls_fill_the_ring_like_single_ticks(hip, monkeypatch, scenario, N, A, W, H, params):
    """a batched call with an output ring -- ONE step launch for its k ticks (step_ticks_kernel / step_<scenario>_ticks_kernel) and ONE raster launch for its k
    observation passes (raster_fast_batch_kernel, raster_glist_batch_kernel for the long lists) -- against single ticks: every slab of the ring, rewards /
    dones rings, state; ragged sizes, one and several agents, every scenario, call sizes 2 .. 8 and chunks the ring wraps around in"""
    import os
    import torch
    monkeypatch.setenv("BOXOBAN_LEVELS", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "boxoban"))
    R = 8
    def make(with_ring):
        g = MegaverseGym(scenario, W, H, N, A, 2, False, params)
        g.set_pixel_mode("fast"); g.seed(33); g.reset()
        obs = torch.zeros((R, N * A, H, W, 4), dtype=torch.uint8, device="cuda:0")
        rew = torch.zeros((R, N * A), dtype=torch.float32, device="cuda:0")
        don = torch.zeros((R, N), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        g.set_output_ring(R, obs.data_ptr(), rew.data_ptr(), don.data_ptr())
        return g, obs, rew, don
    a, oa, ra, da = make(True)
    b, ob, rb, db = make(True)
    chunks = [8, 2, 5, 8, 3, 8, 7, 8]
    st = 0
    for k in chunks:
        a.step_n(k, "multidiscrete", 5, st)
        for j in range(k):
            b.sample_random_actions(5, st + j); b.step()
        st += k
    a.synchronize(); b.synchronize(); torch.cuda.synchronize()
    assert oa.cpu().numpy()[..., :3].max() > 0
    assert torch.equal(oa, ob), "observation ring differs"
    assert torch.equal(ra.view(torch.int32), rb.view(torch.int32)) and torch.equal(da, db)
    for e in range(N):
        assert a.debug_snapshot_bytes(e).tobytes() == b.debug_snapshot_bytes(e).tobytes(), e
    assert a.get_true_objectives().tobytes() == b.get_true_objectives().tobytes()
    a.close(); b.close()


@pytest.mark.parametrize("scenario,N,A,W,H,params", [("TowerBuilding", 300, 1, 128, 128, {}), ("TowerBuilding", 64, 4, 64, 64, {}), ("ObstaclesHard", 200, 1, 128, 128, {}),
                                                     ("Rearrange", 100, 1, 64, 64, {}), ("Collect", 80, 1, 64, 64, {}), ("HexMemory", 60, 1, 64, 64, {}), ("HexExplore", 40, 1, 128, 72, {}),
                                                     ("Sokoban", 90, 1, 64, 64, {}), ("TowerBuilding", 50, 1, 64, 64, {"episodeLengthSec": -200.0})])
def test_overlapped_passes_fill_the_ring_like_single_ticks(hip, monkeypatch, scenario, N, A, W, H, params):
    """mv_set_pass_overlap: with a ring two calls deep the one-launch observation passes of consecutive batched calls run on two internal streams
    (the passes of call c + 1 begin while those of call c drain) -- against single ticks: every slab of the ring, the rewards / dones rings, the
    state, the true objectives; full calls and ragged ones, and episodes short enough that the library must decline to overlap (last case)."""
    import os
    import torch
    monkeypatch.setenv("BOXOBAN_LEVELS", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "boxoban"))
    R = 16
    def make(overlap):
        g = MegaverseGym(scenario, W, H, N, A, 2, False, params)
        g.set_pixel_mode("fast"); g.seed(35); g.reset()
        obs = torch.zeros((R, N * A, H, W, 4), dtype=torch.uint8, device="cuda:0")
        rew = torch.zeros((R, N * A), dtype=torch.float32, device="cuda:0")
        don = torch.zeros((R, N), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        g.set_output_ring(R, obs.data_ptr(), rew.data_ptr(), don.data_ptr())
        if overlap:
            g.set_pass_overlap(True)
        return g, obs, rew, don
    a, oa, ra, da = make(True)
    b, ob, rb, db = make(False)
    chunks = [8, 8, 8, 8, 8, 3, 8, 8, 5, 8, 8, 8, 2, 8, 8, 8]
    st = 0
    seen = []
    for k in chunks:
        a.step_n(k, "multidiscrete", 5, st)
        for j in range(k):
            b.sample_random_actions(5, st + j); b.step()
        st += k
        # a consumer on the caller's stream right after the call, as the contract asks: the entries this call filled
        seen.append(int(oa[(st - 1) % R, ::7, ::5, ::3].to(torch.int64).sum().item()) - int(ob[(st - 1) % R, ::7, ::5, ::3].to(torch.int64).sum().item()))
    a.synchronize(); b.synchronize(); torch.cuda.synchronize()
    assert oa.cpu().numpy()[..., :3].max() > 0
    assert not any(seen), "an entry read right after its call differs"
    assert torch.equal(oa, ob), "observation ring differs"
    assert torch.equal(ra.view(torch.int32), rb.view(torch.int32)) and torch.equal(da, db)
    for e in range(N):
        assert a.debug_snapshot_bytes(e).tobytes() == b.debug_snapshot_bytes(e).tobytes(), e
    assert a.get_true_objectives().tobytes() == b.get_true_objectives().tobytes()
    a.close(); b.close()


@pytest.mark.parametrize("k,R,rings", [(16, 16, True), (16, 32, True), (8, 16, True), (8, 16, False), (12, 32, True), (24, 32, True), (24, 48, True)])
def test_overlapped_passes_with_a_late_consumer(hip, k, R, rings):
    """The ring contract of mv_set_pass_overlap under load: the consumer of a call's entries is ENQUEUED right after the call, as the
    header asks, but still pending -- behind a slow kernel on the caller's stream -- when the next calls are issued; no host synchronisation between
    calls.  k = 24 runs as two chunks (the internal batch is 16): with a ring of 32 the library must decline to overlap (the ring is not two CALLS deep),
    with 48 it may; k = 16 with a ring of 16 must decline, with 32 it may; without rewards / dones rings it must decline, too (two passes in flight would both publish the single arrays).  What every
    consumer copied must be what single ticks produce."""
    import torch
    N, A, W, H = 96, 1, 64, 64
    def make(overlap):
        g = MegaverseGym("ObstaclesHard", W, H, N, A, 2, False, {})
        g.set_pixel_mode("fast"); g.seed(77); g.reset()
        obs = torch.zeros((R, N * A, H, W, 4), dtype=torch.uint8, device="cuda:0")
        rew = torch.zeros((R, N * A), dtype=torch.float32, device="cuda:0")
        don = torch.zeros((R, N), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        g.set_output_ring(R, obs.data_ptr(), rew.data_ptr() if rings else 0, don.data_ptr() if rings else 0)
        if overlap:
            g.set_pass_overlap(True)
        return g, obs, rew, don
    a, oa, ra, da = make(True)
    b, ob, rb, db = make(False)
    calls = 6
    big = torch.randn((3072, 3072), device="cuda:0")
    log_o = torch.zeros((calls, k, N * A, H, W, 4), dtype=torch.uint8, device="cuda:0")
    log_r = torch.zeros((calls, k, N * A), dtype=torch.float32, device="cuda:0")
    log_d = torch.zeros((calls, k, N), dtype=torch.uint8, device="cuda:0")
    want_o, want_r, want_d = torch.zeros_like(log_o), torch.zeros_like(log_r), torch.zeros_like(log_d)
    torch.cuda.synchronize()
    st = 0
    for c in range(calls):
        a.step_n(k, "multidiscrete", 9, st)
        idx = torch.arange(st, st + k, device="cuda:0") % R
        # the consumer, enqueued now (before the next call is issued) and slow to start: a few milliseconds of matrix products in front of it
        for _ in range(3):
            big = (big @ big).clamp_(-1.0, 1.0)
        log_o[c].copy_(oa[idx])
        if rings:
            log_r[c].copy_(ra[idx]); log_d[c].copy_(da[idx])
        st += k
    torch.cuda.synchronize()
    st = 0
    for c in range(calls):
        for j in range(k):
            b.sample_random_actions(9, st + j); b.step()
            b.synchronize()
            want_o[c, j].copy_(ob[(st + j) % R])
            if rings:
                want_r[c, j].copy_(rb[(st + j) % R]); want_d[c, j].copy_(db[(st + j) % R])
        st += k
    torch.cuda.synchronize()
    assert want_o[..., :3].max().item() > 0
    assert torch.equal(log_o, want_o), "a consumer enqueued right after its call read something else than single ticks produce"
    assert torch.equal(log_r.view(torch.int32), want_r.view(torch.int32)) and torch.equal(log_d, want_d)
    if not rings:   # the single arrays hold the LAST tick's values, whatever ran in between
        assert a.get_rewards_array().tobytes() == b.get_rewards_array().tobytes() and a.get_dones().tobytes() == b.get_dones().tobytes()
    for e in range(N):
        assert a.debug_snapshot_bytes(e).tobytes() == b.debug_snapshot_bytes(e).tobytes(), e
    a.close(); b.close()
