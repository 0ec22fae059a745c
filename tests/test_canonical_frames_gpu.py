"""The scripted camera poses of tests/canonical_frames.py on the HIP path.  What their pixels must be is derived by hand and asserted on the oracle
(tests/test_canonical_frames.py); here the product renders the same poses: the exact pixel mode equals the oracle byte for byte, the default (fast)
mode -- planar-tile path included: these square-on views are made of tiles that one face covers -- stays within DESIGN.md's tolerance AND within +-1
of the hand-derived values themselves on the pixels the CPU test checks."""
import numpy as np
import pytest

from canonical import find_isolated_box
from canonical_frames import BOX_COLOR, BOX_HALF, EYE_Y, REST_Y, face_wall, find_wall_env, plane_pixel
from hip_util import make_pair

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]

W = H = 128
N_ENVS = 64


def both(og, hg, e, what):
    og.render()
    ref = og.get_observation(e, 0).copy()
    hg.set_pixel_mode("exact"); hg.render()
    exact = hg.get_observation(e, 0).copy()
    assert np.array_equal(ref, exact), f"{what}: exact mode differs from the oracle at {np.argwhere((ref != exact).any(axis=-1))[:4].tolist()}"
    hg.set_pixel_mode("fast"); hg.render()
    fast = hg.get_observation(e, 0).copy()
    d = np.abs(ref.astype(np.int16) - fast.astype(np.int16)).max(axis=-1)
    assert (d > 1).sum() <= 2 and (d > 0).sum() <= max(4, 5e-4 * d.size), f"{what}: fast mode: {(d > 0).sum()} pixels differ, {(d > 1).sum()} by more than 1"
    return fast


def near(px, want, what, tol=1):
    assert px[3] == 255 and all(abs(int(a) - int(b)) <= tol for a, b in zip(px[:3], want[:3])), f"{what}: got {px.tolist()}, derived {want}"


def test_wall_square_on(hip):
    og, hg = make_pair(N_ENVS, 1, W, H, seed=3)
    e = next(e for e in range(N_ENVS) for s in [og.snapshot(e)] if int(s["draw_walls"]) and int(s["W"]) >= 12 and
             not any(int(o[0]) <= 5 for o in s["objects"][: int(s["num_objects"])]) and not (int(s["bz"][0]) <= 5))
    s = og.snapshot(e)
    for g in (og, hg):
        face_wall(g, e, int(s["W"]) // 2 + 0.5, 3.0)
    fast = both(og, hg, e, "wall square-on")
    wallc, floorc = int(s["wall_color"]), int(s["layout_color"])
    for i, j in [(63, 63), (64, 64), (20, 100), (110, 40), (5, 70), (122, 120)]:      # the hand-derived values, on the product's default path
        near(fast[j, i], plane_pixel(i, j, W, H, 2, -3.0, (0, 0, 1), wallc), f"fast wall pixel ({i}, {j})")
    for i, j in [(64, 5), (30, 10), (100, 21)]:
        near(fast[j, i], plane_pixel(i, j, W, H, 1, -(EYE_Y - 1.0), (0, 1, 0), floorc), f"fast floor pixel ({i}, {j})")
    og.close(); hg.close()


def test_distant_wall_and_sky(hip):
    og, hg = make_pair(N_ENVS, 1, W, H, seed=3)
    hit = find_wall_env(og, N_ENVS, 15, 1.5)
    assert hit
    e, z0 = hit
    for g in (og, hg):
        face_wall(g, e, z0, 12.0)
    fast = both(og, hg, e, "distant wall")
    assert fast[H - 2, 64].tolist() == [0, 0, 0, 255]
    og.close(); hg.close()


def test_box_silhouette(hip):
    og, hg = make_pair(N_ENVS, 1, W, H, seed=3)
    e, (ox, oz) = next((e, b) for e in range(N_ENVS) for b in [find_isolated_box(og.snapshot(e))] if b)
    c, sn = float(np.float32(np.cos(np.pi / 2))), float(np.float32(np.sin(np.pi / 2)))
    for g in (og, hg):
        g.debug_set_agent_pos(e, 0, ox + 0.5 + BOX_HALF + 2.5, REST_Y, oz + 0.5)
        g.debug_set_agent_yaw(e, 0, c, sn)
    fast = both(og, hg, e, "box silhouette")
    for j in (21, 30, 40, 48):
        for i in (56, 60, 64, 71):
            near(fast[j, i], plane_pixel(i, j, W, H, 2, -2.5, (0, 0, 1), BOX_COLOR), f"fast box face ({i}, {j})")
    og.close(); hg.close()
