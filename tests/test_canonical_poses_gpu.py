"""The canonical-pose scripts of tests/canonical.py on the HIP path: the same hooks, the same actions, and after EVERY tick the env's whole
state must equal the oracle's bit for bit.  (What the outcomes must be is asserted on the oracle side, tests/test_canonical_poses.py.)"""
import numpy as np
import pytest

from canonical import REST_ON, box_block_then_jump, corner_push, drop_onto_box, find_isolated_box, find_wall_strip, free_jump, head_on, pose, stairs, wall_slide
from hip_util import diff_snapshots, hip_snapshot, make_pair

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def lockstep(og, hg, e, A, script_o, script_h):
    n = 0
    for _ in zip(script_o, script_h):
        d = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
        assert not d, (n, d[:4])
        n += 1
    assert n > 0
    return n


@pytest.mark.parametrize("deg", [30, 45, 60])
def test_wall_slide_bit_exact(hip, deg):
    og, hg = make_pair(64, 1, 16, 16, seed=3)
    e = next(e for e in range(64) if find_wall_strip(og.snapshot(e)))
    W = int(og.snapshot(e)["W"])
    lockstep(og, hg, e, 1, wall_slide(og, e, W, deg), wall_slide(hg, e, W, deg))
    og.close(); hg.close()


def test_corner_push_bit_exact(hip):
    """the ticks whose forward-and-strafe loop runs longest; the HIP path leaves that loop once the target repeats bit for bit (mv_physics.h),
    the oracle runs all its iterations"""
    og, hg = make_pair(64, 1, 16, 16, seed=3)
    e = next(e for e in range(64) if find_wall_strip(og.snapshot(e)))
    W = int(og.snapshot(e)["W"])
    lockstep(og, hg, e, 1, corner_push(og, e, W), corner_push(hg, e, W))
    og.close(); hg.close()


def test_box_block_and_jump_bit_exact(hip):
    og, hg = make_pair(64, 1, 16, 16, seed=3)
    e, (ox, oz) = next((e, b) for e in range(64) for b in [find_isolated_box(og.snapshot(e))] if b)
    lockstep(og, hg, e, 1, box_block_then_jump(og, e, ox, oz), box_block_then_jump(hg, e, ox, oz))
    og.close(); hg.close()


def test_stairs_bit_exact(hip):
    og, hg = make_pair(1, 1, 16, 16, seed=3, scenario="Rearrange")
    lockstep(og, hg, 0, 1, stairs(og, 0), stairs(hg, 0))
    og.close(); hg.close()


def test_head_on_bit_exact(hip):
    og, hg = make_pair(1, 2, 16, 16, seed=3, scenario="Empty")
    for _ in range(5):
        og.step_norender(); hg.step_no_render()
    y = float(og.snapshot(0)["agents"][0]["pos"][1])
    lockstep(og, hg, 0, 2, head_on(og, 0, y), head_on(hg, 0, y))
    og.close(); hg.close()


def test_free_jump_bit_exact(hip):
    og, hg = make_pair(64, 1, 16, 16, seed=3)
    e = next(e for e in range(64) if find_wall_strip(og.snapshot(e)))
    W = int(og.snapshot(e)["W"])
    for g in (og, hg):
        pose(g, e, 0, 3.0, REST_ON(1.0), W - 2.5, 0.0)
    lockstep(og, hg, e, 1, free_jump(og, e), free_jump(hg, e))
    og.close(); hg.close()


def test_drop_onto_a_box_bit_exact(hip):
    og, hg = make_pair(64, 1, 16, 16, seed=3)
    e, (ox, oz) = next((e, b) for e in range(64) for b in [find_isolated_box(og.snapshot(e))] if b)
    lockstep(og, hg, e, 1, drop_onto_box(og, e, ox, oz), drop_onto_box(hg, e, ox, oz))
    og.close(); hg.close()
