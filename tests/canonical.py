"""Scripted single-step physics cases from canonical poses, shared by the CPU test of the oracle
(test_canonical_poses.py: hand-derived expectations) and the GPU test (test_canonical_poses_gpu.py: HIP == oracle, bit for bit, tick by
tick).  A case is a generator: it drives ONE gym-like object -- OracleGym or MegaverseGym, they share the debug hooks -- and yields after
every tick, so that two gyms can be advanced in lock step.

Geometry the cases rely on (all from the cited reference lines):
  capsule radius 0.33, half height 0.525 (agent.cpp:52-54); allowed CCD penetration 0.04 (Bullet dispatch info default): a resting or blocked
  capsule sits 0.04 inside the nominal surface; step height 0.2 (agent.cpp:59); jump speed 6.2 (agent.cpp:157-161), gravity 13.72;
  ground acceleration 50, friction 15, over-speed deceleration 100, max walk speed 4.5, air acceleration 3, max air speed 1
  (kinematic_character_controller.hpp:169-177); dt = 1/15 (env.hpp:160-161);
  movable box: collision half extent 0.39 * 1.15 = 0.4485 around voxel centre + (0, -0.05, 0) (component_object_stacking.hpp:172-185);
  forward = (-sin yaw, 0, -cos yaw) for the yaw basis [[c, s], [-s, c]] (agent.cpp:100-108)."""
import numpy as np

FWD, JUMP = 1 << 3, 1 << 7
R, HH, CCD = 0.33, 0.525, 0.04
REST_ON = lambda top: top + HH + R - CCD   # capsule centre when standing on a surface at height `top`


def yaw_cs(psi):
    return float(np.float32(np.cos(psi))), float(np.float32(np.sin(psi)))


def act(g, e, a, mask):
    """one agent's action for the next tick, by mask, on either kind of gym"""
    if hasattr(g, "set_action_mask"):
        g.set_action_mask(e, a, int(mask))
    else:
        g.set_actions(e, a, [0, 1 if mask & FWD else 0, 0, 1 if mask & JUMP else 0, 0, 0])


def tick(g):
    g.step_norender() if hasattr(g, "step_norender") else g.step_no_render()


def pose(g, e, a, x, y, z, psi):
    c, s = yaw_cs(psi)
    g.debug_set_agent_pos(e, a, float(x), float(y), float(z))
    g.debug_set_agent_yaw(e, a, c, s)
    g.debug_set_agent_velocity(e, a, 0.0, 0.0, 0.0)


def find_wall_strip(snap):
    """a TowerBuilding env whose cells x <= 4 hold no movable box and whose room is at least 16 deep: a free run along the wall x in [0, 1)"""
    objs = snap["objects"][: int(snap["num_objects"])]
    return int(snap["W"]) >= 16 and not any(int(o[0]) <= 4 for o in objs)


def find_isolated_box(snap):
    """a movable box on the floor (y = 1) with nothing on it, nothing within 3 cells in x / 2 in z, and three free cells on its +x side"""
    objs = [tuple(int(v) for v in o) for o in snap["objects"][: int(snap["num_objects"])]]
    cells = {(o[0], o[2]) for o in objs}
    L, W = int(snap["L"]), int(snap["W"])
    for o in objs:
        if o[1] != 1 or o[3] != 0:
            continue
        ox, oz = o[0], o[2]
        near = [c for c in cells if c != (ox, oz) and abs(c[0] - ox) <= 3 and abs(c[1] - oz) <= 2]
        stacked = [p for p in objs if p[0] == ox and p[2] == oz and p[1] > 1]
        if not near and not stacked and ox + 4 < L - 1 and 2 <= oz < W - 2 and ox >= 2:
            return ox, oz
    return None


def wall_slide(g, e, W, deg, ticks=40):
    """walk into the wall x in [0, 1) at `deg` degrees from its normal, sliding towards -z (updateTargetPositionBasedOnCollision,
    kinematic_character_controller.cpp:313-329)"""
    th = np.deg2rad(deg)
    pose(g, e, 0, 3.0, REST_ON(1.0), W - 2.5, np.pi / 2 - th)   # forward = (-cos th, 0, -sin th)
    for _ in range(ticks):
        act(g, e, 0, FWD)
        tick(g)
        yield


def slide_fixed_point(deg):
    """tangential speed the controller settles at, derived by hand from the cited lines: per tick the velocity gains 50 dt along the
    heading (setAcceleration, :753-792), is scaled back to 4.5 when it exceeds it (the over-speed deceleration 100 dt is more than the
    excess), loses its wall-normal component in the slide (hit fraction 0 once the capsule rests 0.04 inside the wall), becomes
    displacement / dt (:587-588), and loses 15 dt to ground friction (:590-602)"""
    th, a, u = np.deg2rad(deg), 50.0 / 15.0, 0.0
    for _ in range(200):
        vt, vn = u + a * np.sin(th), a * np.cos(th)
        s = np.hypot(vt, vn)
        if s > 4.5:
            vt *= (s - 100.0 / 15.0) / s if s - 100.0 / 15.0 > 4.5 else 4.5 / s
        u = max(vt - 15.0 / 15.0, 0.0)
    return u


def box_block_then_jump(g, e, ox, oz, walk=20, jump=12):
    """walk in -x into a movable box standing on the floor: 0.8985 high, no step-up (step height 0.2); then jump with Forward held"""
    pose(g, e, 0, ox + 3.0, REST_ON(1.0), oz + 0.5, np.pi / 2)
    for _ in range(walk):
        act(g, e, 0, FWD)
        tick(g)
        yield
    for t in range(jump):
        act(g, e, 0, FWD | (JUMP if t == 0 else 0))
        tick(g)
        yield


def stairs(g, e, ticks=12):
    """Rearrange's left pedestal from the +z side at x = 7.75: raised floor (top 1.5) -> steps with tops 1.68, 1.84, 2.0
    (scenario_rearrange.cpp:285-298): ledges of 0.18 / 0.16 / 0.16 <= step height 0.2, walked up without jumping"""
    pose(g, e, 0, 7.75, REST_ON(1.5), 11.5, 0.0)   # forward = (0, 0, -1)
    for _ in range(ticks):
        act(g, e, 0, FWD)
        tick(g)
        yield


def head_on(g, e, y, ticks=20):
    """two agents on the bare Empty floor walk at each other along x; the controllers run in agent order (env.cpp:126, agent.cpp:64)"""
    pose(g, e, 0, 3.0, y, 5.0, -np.pi / 2)   # forward = (+1, 0, 0)
    pose(g, e, 1, 9.0, y, 5.0, np.pi / 2)    # forward = (-1, 0, 0)
    for _ in range(ticks):
        act(g, e, 0, FWD)
        act(g, e, 1, FWD)
        tick(g)
        yield


def corner_push(g, e, W, ticks=30):
    """walk diagonally into the corner of the walls x in [0, 1) and z in [W - 1, W): the forward-and-strafe loop slides along one wall into
    the other and back until its target stops changing or its ten iterations are used up (kinematic_character_controller.cpp:331-420) --
    the ticks with the most sweeps there are"""
    pose(g, e, 0, 2.5, REST_ON(1.0), W - 3.0, 3 * np.pi / 4 + 0.1)   # forward = (-sin, 0, -cos): towards -x and +z, 5.7 degrees off the diagonal
    for _ in range(ticks):
        act(g, e, 0, FWD)
        tick(g)
        yield


def free_jump(g, e, ticks=16):
    """jump on the spot from the floor, nothing overhead, no key held afterwards (agent.cpp:157-161: vvel = 6.2 on the jump tick)"""
    for t in range(ticks):
        act(g, e, 0, JUMP if t == 0 else 0)
        tick(g)
        yield


def jump_heights(n_max=12):
    """height above the take-off point after tick n = 1 .. n_max of a free jump, derived from playerStep / stepUp / stepDown
    (kinematic_character_controller.cpp:553-561, :223-238, :398-407): every tick vvel -= 13.72 dt FIRST, then the capsule moves by vvel dt
    -- up through stepUp's jump offset while vvel > 0; once vvel < 0 stepUp raises it by the step height 0.2 and stepDown drops it by
    0.2 + |vvel| dt, the same net vvel dt: y_n = dt (6.2 n - (13.72 dt) n (n + 1) / 2)"""
    dt = 1.0 / 15.0
    return [dt * (6.2 * n - 13.72 * dt * n * (n + 1) / 2.0) for n in range(1, n_max + 1)]


def drop_onto_box(g, e, ox, oz, height=1.0, ticks=10):
    """let go one unit above the resting height on top of a movable box (its collision shape's top: 1.5 - 0.05 + 0.4485), centred on it"""
    top = 1.5 - 0.05 + 0.39 * 1.15
    pose(g, e, 0, ox + 0.5, REST_ON(top) + height, oz + 0.5, 0.0)
    for _ in range(ticks):
        act(g, e, 0, 0)
        tick(g)
        yield
