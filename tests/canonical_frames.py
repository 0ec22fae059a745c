"""Scripted camera poses whose pixels can be derived BY HAND from the reference's renderer set-up -- not read off the oracle.

What the derivation uses, all from the cited reference lines:
  * projection (env_renderer.hpp:34-38, agent.cpp:33-37): Matrix4::perspectiveProjection(100 deg, 128/72, 0.01, 120) -- a HORIZONTAL field of view:
    x_ndc = x / (-z tan 50deg), y_ndc = y (128/72) / (-z tan 50deg), whatever the framebuffer's own aspect;
  * camera (agent.cpp:33,95): the eye sits 0.05 + 0.41 above the capsule centre, looks along the agent's forward (-sin yaw, 0, -cos yaw), no roll; pitch 0 here;
  * pixel (i, j) of a W x H frame covers x_ndc in [2 i / W - 1, 2 (i + 1) / W - 1], its colour is the surface under its CENTRE; row 0 is the BOTTOM row
    (glReadPixels order, megaverse_env.py:164-166 flips it for display);
  * shading (magnum_env_renderer.cpp:200-203 + Magnum Shaders::Phong with Flag::VertexColor [3P], as restated in DESIGN.md 5): per channel
    0x55/255 c + 0xbb/255 c 0xaa/255 max(N.L, 0) + pow(max(V.R, 0), 300) for N.L > 0.001, light at (0, 4, 2) IN CAMERA SPACE, L, V, R normalised,
    written as round(255 v) without gamma; background (0, 0, 0), alpha 255;
  * geometry: TowerBuilding's floor top is y = 1, its walls are 1 thick and H high (scenario_tower_building.cpp:19-154 via platforms.hpp), a movable box is a
    cube of half extent 0.39 around its voxel's centre, colour 0xadd8e6 (component_object_stacking.hpp:170-198); the capsule rests 0.04 inside the floor
    (Bullet's allowed CCD penetration): centre = 1 + 0.525 + 0.33 - 0.04.
The expected values are float64 evaluations of these closed forms for PLANAR faces seen square-on; the oracle computes in fp32 through a general
ray / box intersection: the stated tolerance is +-1 of 255 per channel, and exact agreement for which surface a pixel shows away from silhouettes."""
import numpy as np

TAN = np.tan(np.deg2rad(50.0))
TAN_Y = TAN / (128.0 / 72.0)
LIGHT = np.array([0.0, 4.0, 2.0])
AMB, DIF, LCOL = 0x55 / 255.0, 0xBB / 255.0, 0xAA / 255.0
EYE_ABOVE_CENTRE = 0.05 + 0.41
REST_Y = 1.0 + 0.525 + 0.33 - 0.04        # capsule centre of an agent standing on the floor (top at y = 1)
EYE_Y = REST_Y + EYE_ABOVE_CENTRE
BOX_COLOR, BOX_HALF = 0xADD8E6, 0.39


def ray(i, j, W, H):
    """camera-space direction through the centre of pixel (column i, row j counted from the bottom)"""
    return np.array([((i + 0.5) / W * 2 - 1) * TAN, ((j + 0.5) / H * 2 - 1) * TAN_Y, -1.0])


def phong(P, N, color):
    """RGBA8 of a surface point P with unit normal N (both camera space) and 24-bit vertex colour"""
    Ld = LIGHT - P
    Ld = Ld / np.linalg.norm(Ld)
    inten = max(float(N @ Ld), 0.0)
    spec = 0.0
    if inten > 0.001:
        R = 2.0 * float(N @ Ld) * N - Ld
        V = -P / np.linalg.norm(P)
        spec = max(float(V @ R), 0.0) ** 300
    out = []
    for sh in (16, 8, 0):
        c = ((color >> sh) & 255) / 255.0
        out.append(int(np.floor(min(max(AMB * c + DIF * c * LCOL * inten + spec, 0.0), 1.0) * 255.0 + 0.5)))
    return out + [255]


def plane_pixel(i, j, W, H, axis, offset, N, color):
    """pixel (i, j) when its ray hits the camera-space plane {P[axis] = offset}: P = dc * offset / dc[axis]"""
    dc = ray(i, j, W, H)
    return phong(dc * (offset / dc[axis]), np.asarray(N, float), color)


def row_of(y_rel, dist, H):
    """continuous row coordinate (pixel j spans [j, j + 1)) where a point y_rel above the eye at distance `dist` in front of it lands"""
    return ((y_rel / dist) / TAN_Y * 0.5 + 0.5) * H


def col_of(x_rel, dist, W):
    return ((x_rel / dist) / TAN * 0.5 + 0.5) * W


def face_wall(g, e, z0, d):
    """stand on the floor at distance d from the face x = 1 of the wall x in [0, 1), looking at it square-on (forward = (-1, 0, 0): yaw = pi / 2)"""
    c, s = float(np.float32(np.cos(np.pi / 2))), float(np.float32(np.sin(np.pi / 2)))
    g.debug_set_agent_pos(e, 0, 1.0 + d, REST_Y, float(z0))
    g.debug_set_agent_yaw(e, 0, c, s)
    g.debug_set_agent_velocity(e, 0, 0.0, 0.0, 0.0)


def find_wall_env(g, n, min_len, lane_clear):
    """-> (env, z0): drawn walls (draw_walls), the room at least min_len long in x, no movable box within lane_clear of the line z = z0 (and none at all
    within 1.5 of it at x <= 4), z0 at a cell centre at least 4 cells from either end of the wall x in [0, 1)"""
    for e in range(n):
        s = g.snapshot(e)
        if not int(s["draw_walls"]) or int(s["L"]) < min_len:
            continue
        objs = [tuple(int(v) for v in o) for o in s["objects"][: int(s["num_objects"])]]
        for zc in range(5, int(s["W"]) - 5):
            z0 = zc + 0.5
            if any(abs(o[2] + 0.5 - z0) < lane_clear for o in objs):
                continue
            bz = [int(v) for v in s["bz"]]
            if bz[2] - 1 <= zc <= bz[3] + 1:   # keep the building-zone slab out of the lane: the floor checks want the bare floor
                continue
            return e, z0
    return None
