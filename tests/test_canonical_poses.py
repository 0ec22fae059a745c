"""Canonical-pose known answers for the restated controller: scripted cases whose outcome is
derived BY HAND from the cited reference lines -- not read off the oracle -- asserted on the oracle with a stated tolerance.  With Bullet
absent this is the tightest pin the physics can get: the leaf constants and the control flow of KinematicCharacterController decide these
numbers, the restated convex cast only has to deliver "hit fraction 0 when resting 0.04 deep" and "no hit when moving along a face".
The GPU twin (test_canonical_poses_gpu.py) runs the same scripts and asserts HIP == oracle bit for bit after every tick.
Not constructible in the in-scope scenarios: a 0.25 ledge (the scenes' ledges are 0.16 / 0.18 -- Rearrange's pedestal -- or 0.8985 and
up) and a ceiling within jump reach (no scenario has an overhang)."""
import numpy as np
import pytest

import oracle_lib
from canonical import (CCD, HH, R, REST_ON, box_block_then_jump, corner_push, drop_onto_box, find_isolated_box, find_wall_strip, free_jump, head_on, jump_heights,
                       pose, slide_fixed_point, stairs, wall_slide)


def agent(g, e, a=0):
    return g.snapshot(e)["agents"][a]


@pytest.fixture(scope="module")
def tower():
    g = oracle_lib.OracleGym("TowerBuilding", 16, 16, 64, 1, 1, False, {})
    g.seed(3)
    g.reset()
    yield g
    g.close()


@pytest.mark.parametrize("deg", [30, 45, 60])
def test_wall_slide_keeps_the_tangential_component(tower, deg):
    e = next(e for e in range(64) if find_wall_strip(tower.snapshot(e)))
    W = int(tower.snapshot(e)["W"])
    xs, speeds = [], []
    for _ in wall_slide(tower, e, W, deg):
        a = agent(tower, e)
        xs.append(float(a["pos"][0]))
        speeds.append((float(a["hv"][0]), float(a["hv"][1])))
    # blocked 0.04 inside the nominal contact (wall face x = 1): never deeper than the 0.041 the depenetration tolerates, and it stays there
    assert abs(xs[-1] - (1.0 + R - CCD)) < 1.5e-3 and max(abs(x - xs[-1]) for x in xs[-10:]) < 1e-5
    # the wall-normal velocity is gone, the tangential one settles at the hand-derived fixed point
    assert abs(speeds[-1][0]) < 1e-4
    assert abs(-speeds[-1][1] - slide_fixed_point(deg)) < 2e-3 * slide_fixed_point(deg), (speeds[-1], slide_fixed_point(deg))
    assert abs(float(agent(tower, e)["pos"][1]) - REST_ON(1.0)) < 1e-4   # still on the floor


def test_corner_push_comes_to_rest_in_the_corner(tower):
    e = next(e for e in range(64) if find_wall_strip(tower.snapshot(e)))
    W = int(tower.snapshot(e)["W"])
    trace = []
    for _ in corner_push(tower, e, W):
        a = agent(tower, e)
        trace.append((float(a["pos"][0]), float(a["pos"][2]), float(a["hv"][0]), float(a["hv"][1]), float(a["pos"][1])))
    # blocked 0.04 inside both nominal contacts (faces x = 1 and z = W - 1), at rest, still on the floor
    x, z, hvx, hvz, y = trace[-1]
    assert abs(x - (1.0 + R - CCD)) < 2.5e-3 and abs(z - (W - 1.0 - R + CCD)) < 2.5e-3, (x, z, W)
    assert abs(hvx) < 1e-3 and abs(hvz) < 1e-3 and abs(y - REST_ON(1.0)) < 1e-4
    assert max(abs(t[0] - x) + abs(t[1] - z) for t in trace[-8:]) < 1e-4


def test_a_box_is_not_a_step_but_a_jump_clears_it(tower):
    e, (ox, oz) = next((e, b) for e in range(64) for b in [find_isolated_box(tower.snapshot(e))] if b)
    trace = []
    for _ in box_block_then_jump(tower, e, ox, oz):
        a = agent(tower, e)
        trace.append((float(a["pos"][0]), float(a["pos"][1]), float(a["hv"][0]), float(a["vvel"])))
    face = ox + 0.5 + 0.39 * 1.15          # +x face of the box's collision shape
    top = 1.5 - 0.05 + 0.39 * 1.15         # its top: 0.8985 above the floor, more than the 0.2 step height
    x, y, hvx, _ = trace[19]               # after 20 ticks of walking: stopped at the face, still on the floor
    assert abs(x - (face + R - CCD)) < 1.5e-3 and abs(y - REST_ON(1.0)) < 1e-4 and abs(hvx) < 1e-4
    # the jump: vvel = 6.2 - 13.72 dt after the first tick; while the capsule is below the top the box still blocks it; in the air the
    # horizontal speed grows by 3 dt per tick up to 1.0; it comes down ON the box: centre = top + 0.525 + 0.33 - 0.04
    assert abs(trace[20][3] - (6.2 - 13.72 / 15.0)) < 1e-4
    assert abs(trace[20][2]) < 1e-4 and abs(trace[21][2]) < 1e-4
    assert abs(-trace[22][2] - 0.2) < 1e-3 and max(-t[2] for t in trace[20:29]) < 1.0 + 1e-3
    assert 1.0 < max(t[1] for t in trace[20:]) - REST_ON(1.0) < 1.4 + 1e-3          # apex below 6.2^2 / (2 * 13.72) = 1.40
    landed = [t for t in trace[20:] if t[3] == 0.0]
    assert landed and abs(landed[0][1] - REST_ON(top)) < 1.5e-3 and landed[0][0] < face   # standing on top of the box


def test_ledges_up_to_the_step_height_are_walked_up():
    g = oracle_lib.OracleGym("Rearrange", 16, 16, 1, 1, 1, False, {})
    g.seed(3)
    g.reset()
    ys, zs = [], []
    for _ in stairs(g, 0):
        a = agent(g, 0)
        ys.append(float(a["pos"][1])); zs.append(float(a["pos"][2]))
        assert float(a["vvel"]) == 0.0          # never airborne: no jump involved
    # 0.3 per tick (4.5 dt) after the first two ticks; heights while standing on the raised floor and on the three steps
    assert abs((zs[3] - zs[4]) - 0.3) < 1e-4
    for z, y in zip(zs, ys):
        top = 1.5 if z > 9.5 + R else 1.68 if 9.0 + R < z < 9.5 - R else 1.84 if 8.5 + R < z < 9.0 - R else 2.0 if 7.0 + R < z < 8.5 - R else None
        if top is not None:   # (between two of them the capsule's round bottom rides the edge)
            assert abs(y - REST_ON(top)) < 1.5e-3, (z, y, top)
    assert abs(ys[-1] - REST_ON(2.0)) < 1.5e-3 and ys[0] < REST_ON(1.5) + 1e-3
    g.close()


def test_two_agents_head_on_stop_each_other_in_agent_order():
    g = oracle_lib.OracleGym("Empty", 16, 16, 1, 2, 1, False, {})
    g.seed(3)
    g.reset()
    for _ in range(5):
        g.step_norender()
    y = float(agent(g, 0)["pos"][1])
    for _ in head_on(g, 0, y):
        pass
    a0, a1 = agent(g, 0, 0), agent(g, 0, 1)
    gap = float(a1["pos"][0] - a0["pos"][0])
    assert abs(gap - (2 * R - CCD)) < 1.5e-3                                   # capsule against capsule: summed radii minus the CCD allowance
    assert np.all(a0["hv"] == 0) and np.all(a1["hv"] == 0)                     # both cancelled ("stop dead", :371-385)
    assert float(a0["pos"][2]) == 5.0 and float(a1["pos"][2]) == 5.0           # exactly head-on: nothing to slide along
    mid = 0.5 * float(a0["pos"][0] + a1["pos"][0])
    assert 6.0 < mid < 6.0 + 0.3                                               # agent 0 moves first: it gains up to one tick's travel (4.5 dt)
    g.close()


# ---- cases that depend on the cast's FRACTION, tick by tick: a reformulated convex_cast / player_step has to keep reproducing numbers
# derived from kinematic_character_controller.cpp, not a regenerated golden file

@pytest.mark.parametrize("deg", [30, 60])
def test_wall_slide_holds_the_contact_depth_on_every_tick(tower, deg):
    """from the tick the capsule reaches the wall on, EVERY tick's forward sweep must stop it 0.04 inside the nominal contact again (hit fraction 0 of
    the normal component, the tangential remainder kept: updateTargetPositionBasedOnCollision, :313-329) -- not only the last tick"""
    e = next(e for e in range(64) if find_wall_strip(tower.snapshot(e)))
    W = int(tower.snapshot(e)["W"])
    xs, zs, us = [], [], []
    for _ in wall_slide(tower, e, W, deg, ticks=30):
        a = agent(tower, e)
        xs.append(float(a["pos"][0])); zs.append(float(a["pos"][2])); us.append(-float(a["hv"][1]))
    first = next(i for i, x in enumerate(xs) if x < 1.0 + R)          # the tick it touches
    assert first < 12
    for i in range(first + 1, len(xs)):
        assert abs(xs[i] - (1.0 + R - CCD)) < 1.5e-3, (i, xs[i])
        # the tangential travel of the tick is the speed the tick started with after acceleration and clamp, i.e. (speed after friction) + 15 dt
        assert abs((zs[i - 1] - zs[i]) * 15.0 - (us[i] + 1.0)) < 5e-3 or us[i] == 0.0, (i, zs[i - 1] - zs[i], us[i])
    assert abs(us[-1] - slide_fixed_point(deg)) < 2e-3 * slide_fixed_point(deg)


def test_free_jump_follows_the_derived_heights_and_lands_on_tick_13(tower):
    e = next(e for e in range(64) if find_wall_strip(tower.snapshot(e)))
    W = int(tower.snapshot(e)["W"])
    pose(tower, e, 0, 3.0, REST_ON(1.0), W - 2.5, 0.0)
    for _ in range(3):   # settle
        tower.set_action_mask(e, 0, 0); tower.step_norender()
    y0 = float(agent(tower, e)["pos"][1])
    assert abs(y0 - REST_ON(1.0)) < 1e-4
    ys, vs = [], []
    for _ in free_jump(tower, e, ticks=16):
        a = agent(tower, e)
        ys.append(float(a["pos"][1]) - y0); vs.append(float(a["vvel"]))
    want = jump_heights(12)
    for n in range(12):                                   # ticks 1 .. 12: in the air, on the derived parabola of the semi-implicit integrator
        assert abs(ys[n] - want[n]) < 2e-4, (n + 1, ys[n], want[n])
        assert abs(vs[n] - (6.2 - 13.72 * (n + 1) / 15.0)) < 1e-4
    # apex 1.1995 after tick 6 (the continuous 6.2^2 / (2 * 13.72) = 1.40 is never reached)
    assert abs(max(ys) - want[5]) < 2e-4 and 1.19 < want[5] < 1.21 and want[6] < want[5]
    assert want[11] > 0.2 and ys[12] != want[11]          # tick 13 would go 0.38 below the floor: stepDown's sweep hits it
    assert abs(ys[12]) < 1.5e-3 and vs[12] == 0.0         # landed: back at the resting height (0.04 inside the nominal contact), vertical velocity cleared
    assert all(abs(y) < 1.5e-3 and v == 0.0 for y, v in zip(ys[13:], vs[13:]))


def test_a_drop_lands_on_a_movable_box_on_tick_6(tower):
    e, (ox, oz) = next((e, b) for e in range(64) for b in [find_isolated_box(tower.snapshot(e))] if b)
    top = 1.5 - 0.05 + 0.39 * 1.15
    ys, vs = [], []
    for _ in drop_onto_box(tower, e, ox, oz, height=1.0, ticks=10):
        a = agent(tower, e)
        ys.append(float(a["pos"][1])); vs.append(float(a["vvel"]))
    g_dt2 = 13.72 / 225.0
    for n in range(1, 6):                                 # free fall: y = start - g dt^2 n (n + 1) / 2 (vvel -= g dt before every move)
        assert abs(ys[n - 1] - (REST_ON(top) + 1.0 - g_dt2 * n * (n + 1) / 2.0)) < 2e-4, (n, ys[n - 1])
        assert abs(vs[n - 1] + 13.72 * n / 15.0) < 1e-4
    assert g_dt2 * 15 < 1.0 < g_dt2 * 21                  # 5 ticks fall 0.915, 6 ticks would fall 1.28: the sixth tick's sweep meets the box
    assert abs(ys[5] - REST_ON(top)) < 1.5e-3 and vs[5] == 0.0, (ys[5], REST_ON(top))
    assert all(abs(y - REST_ON(top)) < 1.5e-3 for y in ys[5:])
