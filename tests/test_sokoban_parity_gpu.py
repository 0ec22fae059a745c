"""Sokoban on the HIP path (second half) against the oracle, bit for bit: reset (level sequence, slabs, cells,
boxes, spawn poses), pixels (scaled slabs, wall caps, goal pads, boxes), long random rollouts with auto-resets, a scripted push onto
a goal with its reward, short episodes (a new level every few ticks: the per-env shuffled level list + refill ring), mid-run
re-seeding (the level list survives Env::seed, what was generated ahead is put back).  Levels: synthetic files in the public Boxoban
text format under tests/golden/boxoban (BOXOBAN_LEVELS)."""
import os

import numpy as np
import pytest

from hip_util import diff_snapshots, hip_snapshot, make_pair, set_same_actions

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
LEVEL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "boxoban")


@pytest.fixture(autouse=True)
def boxoban_env(monkeypatch):
    monkeypatch.setenv("BOXOBAN_LEVELS", LEVEL_DIR)


def frames(g, N, A):
    return np.stack([g.get_observation(e, a) for e in range(N) for a in range(A)])


@pytest.mark.parametrize("A,seed", [(1, 42), (2, 7), (4, 2024)])
def test_reset_parity_over_several_episodes(hip, A, seed):
    N = 12
    og, hg = make_pair(N, A, 32, 32, seed=seed, scenario="Sokoban")
    for episode in range(9):   # 9 > 6 usable levels per file: crosses a level-file reload
        for e in range(N):
            d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
            assert not d, (episode, e, d[:5])
        og.reset(); hg.reset()
    og.close(); hg.close()


@pytest.mark.parametrize("W,H", [(128, 128), (128, 72), (48, 20)])
def test_pixel_parity(hip, W, H):
    N, A = 6, 2
    og, hg = make_pair(N, A, W, H, seed=11, scenario="Sokoban")
    assert np.array_equal(frames(og, N, A), frames(hg, N, A))
    for st in range(120):
        set_same_actions(og, hg, N, A, 5, st)
        og.step_norender(); hg.step_no_render()
    og.render(); hg.render()
    fo, fh = frames(og, N, A), frames(hg, N, A)
    assert np.array_equal(fo, fh), f"{int((fo != fh).sum())} differing bytes"
    assert fo[..., :3].max() > 0 and fo[..., 3].min() == 255
    og.close(); hg.close()


@pytest.mark.parametrize("A,N,steps", [(1, 16, 2600), (3, 8, 1300)])
def test_rollout_parity_with_auto_resets(hip, A, N, steps):
    og, hg = make_pair(N, A, 32, 32, seed=4, scenario="Sokoban")
    start = [og.snapshot(e)["objects"].copy() for e in range(N)]
    moved, ndone = 0, 0
    for st in range(steps):
        set_same_actions(og, hg, N, A, 6, st)
        og.step_norender(); hg.step_no_render()
        assert np.array_equal(og.get_last_rewards().view(np.uint32), hg.get_rewards_array().view(np.uint32)), st
        if st % 100 == 99 or st == steps - 1:
            do = [og.is_done(e) for e in range(N)]
            assert do == hg.get_dones().astype(bool).tolist()
            for e in range(N):
                so = og.snapshot(e)
                d = diff_snapshots(so, hip_snapshot(hg, e), A)
                assert not d, (st, e, d[:5])
                moved += int((so["objects"] != start[e]).any())
        ndone += sum(og.is_done(e) for e in range(N))
    assert moved > 0, "no box was ever pushed: the push path was not exercised"
    assert ndone >= N, "80 s episodes = 1200 ticks: every env must have reset at least once"
    og.close(); hg.close()


def test_scripted_push_onto_a_goal(hip):
    """Find, in the oracle's state, a box with a free goal cell behind it, teleport the agent into the adjacent cell facing the box
    (debug hook on both sides) and press Interact: the box must move one cell and sokobanBoxOnTarget (+1) must be paid, identically."""
    N, A = 16, 1
    og, hg = make_pair(N, A, 32, 32, seed=3, scenario="Sokoban")
    pushed = 0
    for e in range(N):
        s = og.snapshot(e)
        grid = s["soko"].reshape(32, 32)
        no = int(s["num_objects"])
        boxes = {(int(o[0]), int(o[2])) for o in s["objects"][:no]}
        done = False
        for (bx, bz) in sorted(boxes):
            for dx, dz in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                tx, tz, ax, az = bx + dx, bz + dz, bx - dx, bz - dz
                if not (0 <= tx < 32 and 0 <= tz < 32 and 0 <= ax < 32 and 0 <= az < 32):
                    continue
                if grid[tx, tz] == 2 and grid[bx, bz] != 2 and (tx, tz) not in boxes and grid[ax, az] != 1 and (ax, az) not in boxes:
                    # into cell (ax, 1, az), standing on the floor, a little towards the box: the interact spot is 1 unit ahead of the
                    # camera and a cell is 2 units wide.  The yaw is whatever the spawn drew: the loop below turns on the spot.
                    for g in (og, hg):
                        g.debug_set_agent_pos(e, 0, (ax + 0.5) * 2.0 + dx * 0.55, 2.9, (az + 0.5) * 2.0 + dz * 0.55)
                    done = True
                    break
            if done:
                break
    # every env: turn on the spot (LookLeft) and press Interact every tick; whenever the agent faces its box, it is pushed
    look_left_interact = [0, 0, 1, 0, 1, 0]
    for st in range(60):
        for e in range(N):
            og.set_actions(e, 0, look_left_interact)
            hg.set_actions(e, 0, look_left_interact)
        og.step_norender(); hg.step_no_render()
        ro, rh = og.get_last_rewards(), hg.get_rewards_array()
        assert np.array_equal(ro.view(np.uint32), rh.view(np.uint32)), st
        pushed += int((ro >= 1.0).sum())
        for e in range(N):
            assert not diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A), (st, e)
    assert pushed >= 5, "the scripted pushes did not reach their goals"
    og.close(); hg.close()


def test_short_episodes_and_midrun_reseed(hip):
    N, A = 8, 2
    og, hg = make_pair(N, A, 32, 32, seed=5, scenario="Sokoban", params={"episodeLengthSec": 0.3})   # 5 ticks per episode
    ndone = 0
    for st in range(120):
        set_same_actions(og, hg, N, A, 77, st)
        og.step(); hg.step()
        do = np.array([og.is_done(e) for e in range(N)])
        assert np.array_equal(do, hg.get_dones().astype(bool)), st
        ndone += int(do.sum())
        if st == 50:   # Env::seed mid-run: levels generated ahead from the old stream are dropped, the level list is restored
            og.seed(99); hg.seed(99)
        if st % 10 == 0 or st > 110:
            for e in range(N):
                d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
                assert not d, (st, e, d[:5])
            assert np.array_equal(og.get_observation(0, 0), hg.get_observation(0, 0))
    assert ndone > 150
    og.close(); hg.close()
