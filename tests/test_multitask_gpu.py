"""MultiTaskGym (BASELINE.json configs[4] layout: scenarios dealt round-robin by env index, one HIP gym and one stream per
scenario, one shared observation slab): every sub-gym must simulate exactly the envs of the job-wide seed/action streams."""
import numpy as np
import pytest

import oracle_lib
from hip_util import diff_snapshots, hip_snapshot
from megaverse_amd.multitask import MEGAVERSE_IN_SCOPE, MultiTaskGym
from megaverse_amd.rollout import action_masks, sample_actions

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def test_multitask_equals_per_scenario_oracles(hip, monkeypatch):
    import os
    import torch
    monkeypatch.setenv("BOXOBAN_LEVELS", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "boxoban"))   # Sokoban
    N, A, W, H, S = 16, 2, 64, 36, len(MEGAVERSE_IN_SCOPE)
    assert S == 8
    mt = MultiTaskGym(MEGAVERSE_IN_SCOPE, W, H, N, A, 2)
    obs = mt.attach("cuda:0")
    mt.seed(77); mt.reset()
    ogs = []
    for name in MEGAVERSE_IN_SCOPE:   # the oracle has no stride: simulate all N global envs as this scenario, compare the owned ones
        og = oracle_lib.OracleGym(name, W, H, N, A, 2)
        og.seed(77); og.reset()
        ogs.append(og)

    def compare(tag):
        mt.synchronize()
        torch.cuda.synchronize()
        slab = obs.cpu().numpy()
        for i in range(N):
            k, j = i % S, i // S
            d = diff_snapshots(ogs[k].snapshot(i), hip_snapshot(mt.gyms[k], j), A)
            assert not d, (tag, i, MEGAVERSE_IN_SCOPE[k], d[:4])
            for a in range(A):
                assert np.array_equal(ogs[k].get_observation(i, a), slab[mt.frame_row(i, a)]), (tag, i, a)

    compare("reset")
    for st in range(50):
        masks = action_masks(sample_actions(1234, st, N * A))
        for og in ogs:
            for e in range(N):
                for a in range(A):
                    og.set_action_mask(e, a, int(masks[e * A + a]))
            og.step()
        mt.sample_random_actions(1234, st)
        mt.step()
    compare("after 50 steps")
    r = mt.get_last_rewards().reshape(N, A)
    for i in range(N):
        assert np.array_equal(r[i], ogs[i % S].get_last_rewards().reshape(N, A)[i])
    for og in ogs:
        og.close()
    mt.close()


@pytest.mark.parametrize("A,W,H", [(1, 64, 64), (2, 48, 32)])
def test_union_launches_equal_one_launch_per_gym(hip, monkeypatch, A, W, H):
    """mv_group (one step launch for all eight scenarios, two observation launches: the short-list and the long-list raster variant) against
    the same eight gyms stepped one by one on their own streams (MV_MULTITASK_UNION=0): state of every env, rewards, dones and every byte of
    the shared observation slab must be equal -- in the product's default mode (fast pixels, pipelined), single ticks and batched ones."""
    import os
    import torch
    monkeypatch.setenv("BOXOBAN_LEVELS", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "boxoban"))
    N, S = 32, len(MEGAVERSE_IN_SCOPE)

    def make(union):
        monkeypatch.setenv("MV_MULTITASK_UNION", "1" if union else "0")
        mt = MultiTaskGym(MEGAVERSE_IN_SCOPE, W, H, N, A, 2)
        mt.set_pixel_mode("fast")
        obs = mt.attach("cuda:0")
        mt.seed(5); mt.reset()
        return mt, obs

    a, oa = make(True)
    b, ob = make(False)
    assert a.union and not b.union

    def same(tag):
        a.synchronize(); b.synchronize(); torch.cuda.synchronize()
        for k in range(S):
            for j in range(N // S):
                assert a.gyms[k].debug_snapshot_bytes(j).tobytes() == b.gyms[k].debug_snapshot_bytes(j).tobytes(), (tag, MEGAVERSE_IN_SCOPE[k], j)
            assert a.gyms[k].get_rewards_array().tobytes() == b.gyms[k].get_rewards_array().tobytes(), (tag, k)
            assert np.array_equal(a.gyms[k].get_dones(), b.gyms[k].get_dones()), (tag, k)
        sa, sb = oa.cpu().numpy(), ob.cpu().numpy()
        assert sa[..., :3].max() > 0
        assert np.array_equal(sa, sb), (tag, int((sa != sb).any(axis=-1).sum()), "pixels differ")

    same("reset")
    st = 0
    ring_tick = 0
    for _ in range(30):
        a.sample_random_actions(9, st); a.step()
        b.sample_random_actions(9, st); b.step()
        st += 1
    same("30 single ticks")
    for k in (8, 3, 8, 5):
        a.step_n(k, "multidiscrete", 9, st)
        for j in range(k):
            b.sample_random_actions(9, st + j); b.step()
        st += k
    same("batched ticks")
    # rollout rings on every scenario: a batched group call is TWO launches (step_union_ticks_kernel, raster_union_batch_kernel; one agent per env) --
    # every ring entry against the single ticks of the gyms stepped one by one
    R = 8
    ro, rr, rd = a.set_output_ring(R)
    for k in (8, 5, 8, 2, 8):
        a.step_n(k, "multidiscrete", 9, st)
        want = []
        for j in range(k):
            b.sample_random_actions(9, st + j); b.step()
            b.synchronize(); torch.cuda.synchronize()
            want.append((ob.clone(), [g.get_rewards_array().copy() for g in b.gyms], [g.get_dones().copy() for g in b.gyms]))
        a.synchronize(); torch.cuda.synchronize()
        n = N // S
        for j in range(k):   # (ring entries of this call: ticks t0 .. t0 + k - 1 since the ring was set)
            e = (ring_tick + j) % R
            for q in range(S):
                assert torch.equal(ro[q][e], want[j][0][q * n * A:(q + 1) * n * A]), ("ring observations", MEGAVERSE_IN_SCOPE[q], k, j)
                assert rr[q][e].cpu().numpy().tobytes() == want[j][1][q].tobytes(), ("ring rewards", MEGAVERSE_IN_SCOPE[q], k, j)
                assert np.array_equal(rd[q][e].cpu().numpy(), want[j][2][q]), ("ring dones", MEGAVERSE_IN_SCOPE[q], k, j)
        ring_tick += k
        st += k
    for q in range(S):
        for j in range(N // S):
            assert a.gyms[q].debug_snapshot_bytes(j).tobytes() == b.gyms[q].debug_snapshot_bytes(j).tobytes(), ("rings", MEGAVERSE_IN_SCOPE[q], j)
    a.close(); b.close()


def test_a_grouped_gym_is_stepped_through_its_group_only(hip, monkeypatch):
    import os
    monkeypatch.setenv("BOXOBAN_LEVELS", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "boxoban"))
    from megaverse_amd.extension import GymGroup, MegaverseGym
    g1 = MegaverseGym("TowerBuilding", 32, 32, 4, 1, 1, False, {})
    g2 = MegaverseGym("Collect", 32, 32, 4, 1, 1, False, {})
    g3 = MegaverseGym("Collect", 48, 32, 4, 1, 1, False, {})
    for g in (g1, g2, g3):
        g.seed(1); g.reset()
    with pytest.raises(RuntimeError, match="share device, observation size"):
        GymGroup([g1, g3])
    grp = GymGroup([g1, g2])
    with pytest.raises(RuntimeError, match="belongs to an mv_group"):
        g1.step()
    grp.step(4, True, "multidiscrete", 3, 0)
    g2.close()                      # a member leaves: the group is dissolved, the other gym works on its own again
    with pytest.raises(RuntimeError, match="group is gone"):
        grp.step()
    g1.sample_random_actions(3, 4); g1.step()
    assert g1.get_dones().shape == (4,)
    grp.close(); g1.close(); g3.close()
