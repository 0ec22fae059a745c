"""CPU model of the raster's tile culling + the planar-tile classification (design aid, not product code).

Steps oracle TowerBuilding envs with random actions, then for every frame: the primitive list and conservative screen rectangles as
mv_frame.h builds them, a brute-force per-pixel winner (primitive, entry axis), and per 16 x (4 NP) tile
  * the survivors of the rectangle culling,
  * whether all of the tile's pixels see ONE face of ONE world box (upper bound of a planar path),
  * what the half-plane classification (4 tile corners x the face's 4 edges + the facing test, with a margin) decides.
Usage: python scripts/model_tile_classes.py [envs] [steps] [NP]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import oracle_lib  # noqa: E402
from megaverse_amd.rollout import action_masks, sample_actions  # noqa: E402

TAN_X = np.float32(1.19175359)
TAN_Y = np.float32(1.19175359 / (128.0 / 72.0))
CLIP_W = 0.005
OBJ_HALF = 0.39


def camera(ag):
    eye = np.array([ag["pos"][0], ag["pos"][1] + 0.05 + 0.41, ag["pos"][2]], np.float64)
    sp, cp = np.sin(ag["pitch"]), np.cos(ag["pitch"])
    m00, m02, m20, m22 = [float(x) for x in ag["basis"]]
    c = np.array([[m00, m02 * sp, m02 * cp], [0, cp, -sp], [m20, m22 * sp, m22 * cp]], np.float64)
    return eye, c


def prims_of(s):
    """world boxes (lo, hi) and camera-frame boxes"""
    world, camf = [], []
    for i in range(int(s["num_boxes"])):
        b = s["boxes"][i]
        if b[6] & 2:
            world.append((b[0:3].astype(float), b[3:6].astype(float)))
    bz = s["bz"]
    world.append((np.array([bz[0], 1.0, bz[2]], float), np.array([bz[1], 1.05, bz[3]], float)))
    for i in range(int(s["num_objects"])):
        o = s["objects"][i]
        if o[3] <= 0:
            c = o[0:3].astype(float) + 0.5
            world.append((c - OBJ_HALF, c + OBJ_HALF))
        else:
            hh = OBJ_HALF * 0.78
            c = np.array([0.0, -0.74, -1.0])
            camf.append((c - hh, c + hh))
    bw = float(s["bar_half_width"])
    camf.append((np.array([-bw, -0.1325, -0.201]), np.array([bw, -0.1295, -0.199])))
    return world, camf


def screen_rect(lo, hi, eye, c, W, H, camframe=False):
    corners = np.array([[(lo, hi)[(k >> a) & 1][a] for a in range(3)] for k in range(8)], float)
    cc = corners if camframe else (corners - eye) @ c   # camera space (c^T (p - eye))
    cw = -cc[:, 2]
    if cw.max() < CLIP_W:
        return None
    pts = []
    for k in range(8):
        if cw[k] >= CLIP_W:
            pts.append((cc[k, 0] / cw[k], cc[k, 1] / cw[k]))
    if cw.min() < CLIP_W:
        for a in range(3):
            for k in range(8):
                d = k | (1 << a)
                if d == k:
                    continue
                if (cw[k] >= CLIP_W) != (cw[d] >= CLIP_W):
                    t = (CLIP_W - cw[k]) / (cw[d] - cw[k])
                    p = cc[k] + t * (cc[d] - cc[k])
                    pts.append((p[0] / CLIP_W, p[1] / CLIP_W))
    pts = np.array(pts)
    xmin, xmax = np.clip([pts[:, 0].min() / TAN_X, pts[:, 0].max() / TAN_X], -4, 4)
    ymin, ymax = np.clip([pts[:, 1].min() / TAN_Y, pts[:, 1].max() / TAN_Y], -4, 4)
    fx0, fx1 = (xmin * 0.5 + 0.5) * W - 1.5, (xmax * 0.5 + 0.5) * W + 0.5
    fy0, fy1 = (ymin * 0.5 + 0.5) * H - 1.5, (ymax * 0.5 + 0.5) * H + 0.5
    if fx1 < 0 or fy1 < 0 or fx0 > W or fy0 > H:
        return None
    return (int(np.floor(max(fx0, 0))), int(np.ceil(min(fx1, W - 1))), int(np.floor(max(fy0, 0))), int(np.ceil(min(fy1, H - 1))))


def ray_boxes(d, lo, hi):
    """d [...,3] directions, box bounds relative to the origin; returns entry t (inf = miss) and entry axis"""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t1, t2 = lo * inv, hi * inv
    tn, tf = np.minimum(t1, t2), np.maximum(t1, t2)
    te, tx = tn.max(-1), tf.min(-1)
    ax = tn.argmax(-1)
    hit = (te <= tx) & (te >= 0.01) & (te <= 120.0)
    return np.where(hit, te, np.inf), ax


def classify_tile(lo, hi, c, dcx, dcy, margin=2e-4):
    """Half-plane classification of one world box (bounds relative to the eye) against a tile whose corner pixel rays are dc = (dcx[a], dcy[b], -1).
    returns ('inside', axis) if every ray of the tile enters through one face, 'miss' if no ray can hit, else 'maybe'"""
    D = np.array([[c @ np.array([dcx[a], dcy[b], -1.0]) for a in range(2)] for b in range(2)]).reshape(4, 3)   # corner directions, world axes
    allmiss = True
    for k in range(3):
        p = lo[k] if lo[k] > 0 else hi[k] if hi[k] < 0 else None
        if p is None:
            continue
        sg = 1.0 if p > 0 else -1.0
        facing = sg * D[:, k]
        if (facing <= 0).all():
            continue   # face never front-facing over the tile
        if not (facing > 0).all():
            allmiss = False
            continue   # mixed: unknown
        inside_all, miss_face = True, False
        for m in range(3):
            if m == k:
                continue
            for side, b in ((0, lo[m]), (1, hi[m])):
                g = sg * (p * D[:, m] - b * D[:, k])
                if side == 1:
                    g = -g
                mg = margin * (abs(p) + abs(b))
                if not (g >= mg).all():
                    inside_all = False
                if (g <= -mg).all():
                    miss_face = True
        if inside_all:
            return ("inside", k)
        if not miss_face:
            allmiss = False
    return ("miss", -1) if allmiss else ("maybe", -1)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    NP = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    W = H = 128
    TW, TH = 16, 4 * NP
    g = oracle_lib.OracleGym("TowerBuilding", W, H, N, 1, 1, False, {})
    g.seed(42)
    g.reset()
    for st in range(steps):
        acts = sample_actions(7, st, N)
        g.set_action_masks(action_masks(acts))
        g.step_norender()
    xs = ((np.arange(W) + 0.5) / W * 2 - 1) * TAN_X
    ys = ((np.arange(H) + 0.5) / H * 2 - 1) * TAN_Y
    stat = dict(tiles=0, empty=0, true_planar=0, planar_single=0, planar_multi=0, general=0, surv_hist=np.zeros(12, int),
                general_surv=0, general_surv_refined=0, other_block=0, nvis=[], bg_pixels=0, classified_tiles=0)
    frame_costs = []
    for e in range(N):
        fc = 0.0
        s = g.snapshot(e)
        eye, c = camera(s["agents"][0])
        world, camf = prims_of(s)
        dc = np.stack(np.broadcast_arrays(xs[None, :], ys[:, None], -1.0), -1)   # [H,W,3]
        dw = dc @ c.T
        # brute force winners
        best = np.full((H, W), np.inf)
        who = np.full((H, W), -1)
        axis = np.full((H, W), -1)
        prl = []
        for lo, hi in world:
            r = screen_rect(lo, hi, eye, c, W, H)
            prl.append(("w", lo - eye, hi - eye, r))
        for lo, hi in camf:
            r = screen_rect(lo, hi, eye, c, W, H, camframe=True)
            prl.append(("c", lo, hi, r))
        prl = [p for p in prl if p[3] is not None]
        stat["nvis"].append(len(prl))
        for i, (kind, lo, hi, r) in enumerate(prl):
            t, ax = ray_boxes(dw if kind == "w" else dc, lo, hi)
            upd = t < best
            best[upd] = t[upd]; who[upd] = i; axis[upd] = ax[upd]
        stat["bg_pixels"] += int((who < 0).sum())
        for ty0 in range(0, H, TH):
            for tx0 in range(0, W, TW):
                tx1, ty1 = tx0 + TW - 1, ty0 + TH - 1
                surv = [i for i, p in enumerate(prl) if p[3][0] <= tx1 and p[3][1] >= tx0 and p[3][2] <= ty1 and p[3][3] >= ty0]
                stat["tiles"] += 1
                if not surv:
                    stat["empty"] += 1
                    fc += 5
                    continue
                stat["surv_hist"][min(len(surv), 11)] += 1
                w_t, a_t = who[ty0:ty0 + TH, tx0:tx0 + TW], axis[ty0:ty0 + TH, tx0:tx0 + TW]
                one = (w_t == w_t[0, 0]).all() and (a_t == a_t[0, 0]).all() and w_t[0, 0] >= 0 and prl[w_t[0, 0]][0] == "w"
                if one:
                    stat["true_planar"] += 1
                # classification
                if any(prl[i][0] == "c" for i in surv):
                    stat["other_block"] += 1
                    stat["general"] += 1
                    stat["general_surv"] += len(surv); stat["general_surv_refined"] += len(surv)
                    fc += 250 + 44 * len(surv) + 60
                    continue
                stat["classified_tiles"] += 1
                res = [classify_tile(prl[i][1], prl[i][2], c, (xs[tx0], xs[tx1]), (ys[ty0], ys[ty1])) for i in surv]
                notmiss = [(i, r) for i, r in zip(surv, res) if r[0] != "miss"]
                if len(notmiss) == 1 and notmiss[0][1][0] == "inside":
                    assert one and w_t[0, 0] == notmiss[0][0] and a_t[0, 0] == notmiss[0][1][1], (e, tx0, ty0)
                    stat["planar_single" if len(surv) == 1 else "planar_multi"] += 1
                    fc += 80
                else:
                    for i, r in zip(surv, res):   # a 'miss' must be a true miss
                        if r[0] == "miss":
                            assert not (w_t == i).any(), (e, tx0, ty0, i)
                    stat["general"] += 1
                    stat["general_surv"] += len(surv)
                    stat["general_surv_refined"] += len(notmiss)
                    fc += (250 + 44 * len(notmiss)) if notmiss else 5
        frame_costs.append(fc)
    fcs = np.array(frame_costs)
    print(f"per-frame cost estimate (VALU per wave-tile units): mean {fcs.mean():.0f} p10 {np.percentile(fcs,10):.0f} p50 {np.percentile(fcs,50):.0f} p90 {np.percentile(fcs,90):.0f} p99 {np.percentile(fcs,99):.0f} max {fcs.max():.0f}")
    T = stat["tiles"]
    print(f"NP={NP} frames={N} tiles={T} empty={stat['empty'] / T:.3f} visible prims/frame mean {np.mean(stat['nvis']):.1f} max {np.max(stat['nvis'])}")
    print(f"background pixel fraction {stat['bg_pixels'] / (N * W * H):.3f}")
    ne = T - stat["empty"]
    print(f"of non-empty tiles: true planar {stat['true_planar'] / ne:.3f}  classified planar single-survivor {stat['planar_single'] / ne:.3f}  "
          f"multi-survivor {stat['planar_multi'] / ne:.3f}  general {stat['general'] / ne:.3f} (camera-frame prim in tile {stat['other_block'] / ne:.3f})")
    print("survivor histogram (non-empty tiles):", (stat["surv_hist"] / ne).round(3).tolist())
    print(f"general tiles: survivors {stat['general_surv'] / max(stat['general'], 1):.2f} -> after 'miss' refinement {stat['general_surv_refined'] / max(stat['general'], 1):.2f}")


if __name__ == "__main__":
    main()
