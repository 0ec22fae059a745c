"""Renderer known answers: pixels derived BY HAND from magnum_env_renderer.cpp:200-203 (Phong uniforms), env_renderer.hpp:34-38 /
agent.cpp:33-37 (projection, eye height) and the scenario geometry -- tests/canonical_frames.py holds the closed forms -- asserted on the ORACLE with a
stated tolerance: +-1 of 255 per channel for shaded values, exact for which surface a pixel shows (silhouette columns, horizon rows, background).
With Magnum / GL absent this is the pin the pixels can get: the camera model, the row order, the light's frame, the colour formula and the depth order are
decided by these numbers; the oracle's general ray / box code only has to reproduce them.  The GPU twin (test_canonical_frames_gpu.py) renders the same
poses: exact mode bit-equal to the oracle, fast mode within DESIGN.md's tolerance."""
import numpy as np
import pytest

import oracle_lib
from canonical import find_isolated_box
from canonical_frames import BOX_COLOR, BOX_HALF, EYE_Y, REST_Y, col_of, face_wall, find_wall_env, plane_pixel, ray, row_of

W = H = 128
N_ENVS = 64


@pytest.fixture(scope="module")
def tower():
    g = oracle_lib.OracleGym("TowerBuilding", W, H, N_ENVS, 1, 1, False, {})
    g.seed(3)
    g.reset()
    yield g
    g.close()


def close(px, want, what):
    assert px[3] == 255 and all(abs(int(a) - int(b)) <= 1 for a, b in zip(px[:3], want[:3])), f"{what}: got {px.tolist()}, derived {want}"


def wall_square_on(g):
    """-> (frame, wall colour, floor colour): three units in front of the wall x in [0, 1), no movable box in sight"""
    e = next(e for e in range(N_ENVS) for s in [g.snapshot(e)] if int(s["draw_walls"]) and int(s["W"]) >= 12 and
             not any(int(o[0]) <= 5 for o in s["objects"][: int(s["num_objects"])]) and not (int(s["bz"][0]) <= 5))
    s = g.snapshot(e)
    face_wall(g, e, int(s["W"]) // 2 + 0.5, 3.0)
    g.render()
    return e, g.get_observation(e, 0).copy(), int(s["wall_color"]), int(s["layout_color"])


def test_wall_seen_square_on(tower):
    e, f, wallc, floorc = wall_square_on(tower)
    assert f.shape == (H, W, 4)
    d = 3.0
    # the wall is the camera-space plane z = -d with normal (0, 0, 1): centre pixels and three off-centre ones (all inside the wall's extent)
    for i, j in [(63, 63), (64, 64), (63, 64), (20, 100), (110, 40), (5, 70), (122, 120)]:
        close(f[j, i], plane_pixel(i, j, W, H, 2, -d, (0, 0, 1), wallc), f"wall pixel ({i}, {j})")
    # the floor (top at y = 1) is the plane y = -(EYE_Y - 1) with normal (0, 1, 0) -- pitch 0: camera up is world up; it meets the wall at row
    # row_of(-(EYE_Y - 1), d) = 23.4 counted from the BOTTOM: rows 0 .. 21 floor, rows 25 .. wall
    edge = row_of(-(EYE_Y - 1.0), d, H)
    assert 23.0 < edge < 24.0
    for i, j in [(64, 5), (30, 10), (100, 21), (64, 3)]:   # (rows 0 - 1 of the middle columns belong to the time bar, below)
        close(f[j, i], plane_pixel(i, j, W, H, 1, -(EYE_Y - 1.0), (0, 1, 0), floorc), f"floor pixel ({i}, {j})")
    for i in (10, 64, 120):
        close(f[25, i], plane_pixel(i, 25, W, H, 2, -d, (0, 0, 1), wallc), f"wall just above the floor line ({i}, 25)")
    # the wall is 5 high: its top is above the frame at this distance (row_of > H): no background anywhere
    assert row_of(5.0 - EYE_Y, d, H) > H and (f[..., :3].max(axis=-1) > 0).all()
    # the agent's own time bar (scenario_default.hpp:99-170): a 0.003 high, 0.002 deep box 0.2 in front of the eye and 0.131 below it, colour 0x2eb5d0,
    # in the CAMERA's frame: its front face (camera-space plane z = -0.199, normal (0, 0, 1)) covers the centre of row 1 --
    # row_of(-0.131, 0.2) = 1.47 -- between columns col_of(-+half width, 0.2)
    bw = float(tower.snapshot(e)["bar_half_width"])
    assert 1.0 < row_of(-0.131, 0.2, H) < 2.0
    lo, hi = col_of(-bw, 0.199, W), col_of(bw, 0.199, W)
    for i in range(int(np.ceil(lo)) + 1, int(np.floor(hi)) - 1, 7):
        close(f[1, i], plane_pixel(i, 1, W, H, 2, -0.199, (0, 0, 1), 0x2EB5D0), f"time bar ({i}, 1)")
    if lo > 3:
        close(f[1, 0], plane_pixel(0, 1, W, H, 1, -(EYE_Y - 1.0), (0, 1, 0), floorc), "floor left of the time bar (0, 1)")


def test_sky_above_a_distant_wall_and_the_horizon_rows(tower):
    hit = find_wall_env(tower, N_ENVS, 15, 1.5)
    assert hit, "no TowerBuilding env with drawn walls, 15 cells of length and a free lane among the first 64 of seed 3"
    e, z0 = hit
    s = tower.snapshot(e)
    d = 12.0
    face_wall(tower, e, z0, d)
    tower.render()
    f = tower.get_observation(e, 0)
    wallc, floorc = int(s["wall_color"]), int(s["layout_color"])
    Hw = float(int(s["H"]))
    top = row_of(Hw - EYE_Y, d, H)                 # the wall's top edge: above it nothing is drawn -- the clear colour (0, 0, 0), alpha 255
    foot = row_of(-(EYE_Y - 1.0), d, H)            # where the floor meets the wall
    assert top < H - 3 and foot > 3
    for i in (63, 64):
        for j in range(int(np.ceil(top)) + 1, H):
            assert f[j, i].tolist() == [0, 0, 0, 255], f"background expected at ({i}, {j}), wall top at row {top:.2f}"
        for j in range(int(np.ceil(foot)) + 1, int(np.floor(top)) - 1):
            close(f[j, i], plane_pixel(i, j, W, H, 2, -d, (0, 0, 1), wallc), f"distant wall ({i}, {j})")
        for j in range(3, int(np.floor(foot)) - 1):   # (rows 0 - 2: the time bar)
            close(f[j, i], plane_pixel(i, j, W, H, 1, -(EYE_Y - 1.0), (0, 1, 0), floorc), f"floor in front of the distant wall ({i}, {j})")


def test_a_movable_box_lands_on_the_derived_columns_and_rows(tower):
    e, (ox, oz) = next((e, b) for e in range(N_ENVS) for b in [find_isolated_box(tower.snapshot(e))] if b)
    s = tower.snapshot(e)
    d = 2.5
    c, sn = float(np.float32(np.cos(np.pi / 2))), float(np.float32(np.sin(np.pi / 2)))
    ax = ox + 0.5 + BOX_HALF + d
    tower.debug_set_agent_pos(e, 0, ax, REST_Y, oz + 0.5)      # d in front of the box's +x face, on its axis, looking at it (towards -x)
    tower.debug_set_agent_yaw(e, 0, c, sn)
    tower.render()
    f = tower.get_observation(e, 0)
    floorc = int(s["layout_color"])
    left, right = col_of(-BOX_HALF, d, W), col_of(BOX_HALF, d, W)                   # 55.62 and 72.38: pixel centres 56.5 .. 71.5 lie on the face
    bottom, topf = row_of(1.5 - BOX_HALF - EYE_Y, d, H), row_of(1.5 + BOX_HALF - EYE_Y, d, H)   # 19.5 and 49.3
    assert abs(left - 55.62) < 0.01 and abs(right - 72.38) < 0.01 and abs(bottom - 19.5) < 0.1 and abs(topf - 49.3) < 0.1
    for j in (21, 30, 40, 48):
        for i in (56, 60, 64, 71):
            close(f[j, i], plane_pixel(i, j, W, H, 2, -d, (0, 0, 1), BOX_COLOR), f"box face ({i}, {j})")
        for i in (55, 72):                                                            # one column outside the silhouette: what lies behind the box's plane
            t_floor = (EYE_Y - 1.0) / -ray(i, j, W, H)[1]                             # where this pixel's ray reaches the floor's height ...
            if ax - t_floor > 1.05:                                                   # ... still on the floor (it starts at x = 1): the floor
                close(f[j, i], plane_pixel(i, j, W, H, 1, -(EYE_Y - 1.0), (0, 1, 0), floorc), f"floor beside the box ({i}, {j})")
            elif ax - t_floor < 0.95 and not int(s["draw_walls"]):                    # ... beyond its end, and this room's walls are not drawn: nothing
                assert f[j, i].tolist() == [0, 0, 0, 255], f"background beside the box ({i}, {j})"
    # the eye is above the box: over the front face's top edge (row 49.3) its TOP face shows for a few rows (far edge at row_of(.., d + 0.78) = 52.8)
    far_edge = row_of(1.5 + BOX_HALF - EYE_Y, d + 2 * BOX_HALF, H)
    assert 52.5 < far_edge < 53.0
    for j in (50, 51):
        for i in (58, 64, 69):
            close(f[j, i], plane_pixel(i, j, W, H, 1, 1.5 + BOX_HALF - EYE_Y, (0, 1, 0), BOX_COLOR), f"box top ({i}, {j})")
    # below the face (row 19.5) the floor in front of the box
    for i in (60, 68):
        close(f[17, i], plane_pixel(i, 17, W, H, 1, -(EYE_Y - 1.0), (0, 1, 0), floorc), f"floor in front of the box ({i}, 17)")


def test_rows_are_bottom_up(tower):
    """glReadPixels order: row 0 is the bottom of the picture -- standing on the floor looking level, the floor is in the LOW rows"""
    e, f, wallc, floorc = wall_square_on(tower)
    close(f[4, 64], plane_pixel(64, 4, W, H, 1, -(EYE_Y - 1.0), (0, 1, 0), floorc), "bottom rows show the floor")
    close(f[H - 3, 64], plane_pixel(64, H - 3, W, H, 2, -3.0, (0, 0, 1), wallc), "top rows show the wall")
