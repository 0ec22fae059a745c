"""Generates tests/golden/*.npz from the CPU oracle (oracle/libmv_oracle.so).

These fixtures are RESTATEMENT-RELATIVE: the reference's own tests hold no golden vector for this
path and the reference cannot be built here (Bullet/Magnum absent), so the vectors
pin the oracle against silent change and give the HIP path a fixed target at sizes the oracle runs
in seconds.  What IS reference-anchored inside them: the RNG stream (checked separately against the
reference's util.hpp, tests/test_oracle_spec.py).  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib  # noqa: E402
from megaverse_amd.rollout import action_masks, sample_actions  # noqa: E402

SCALAR_FIELDS = ["L", "H", "W", "bz", "layout_color", "wall_color", "draw_walls", "num_objects", "num_boxes", "episode_len",
                 "bz_reward", "highest_tower", "num_frames", "episode_sec", "num_terrain", "num_rewards", "num_platforms", "solved"]


def snap_dict(s, A):
    d = {k: np.asarray(s[k]).copy() for k in SCALAR_FIELDS}
    d["boxes"] = s["boxes"][: int(s["num_boxes"])].copy()
    d["objects"] = s["objects"][: int(s["num_objects"])].copy()
    d["terrain"] = s["terrain"][: int(s["num_terrain"])].copy()
    d["rewards"] = s["rewards"][: int(s["num_rewards"])].copy()
    for f in ("pos", "basis", "pitch", "hv", "vvel", "carrying", "spawn", "total_reward"):
        d["agent_" + f] = np.stack([np.asarray(s["agents"][k][f]) for k in range(A)])
    d["chunk_sum"] = np.asarray(s["chunk"]).astype(np.int64).sum()
    if int(s["hex_num_boxes"]):   # Hex scenarios: the maze's boxes and the collectables, the device's own 32-byte records
        d["hex_boxes"] = s["hex_boxes"][: int(s["hex_num_boxes"])].copy()
        d["hex_objs"] = s["hex_objs"][: int(s["hex_num_objs"])].copy()
    return d


def make(name, N, A, steps, trace_every, W, H, params=None, seed=42, action_seed=1234, scenario="TowerBuilding"):
    g = oracle_lib.OracleGym(scenario, W, H, N, A, 1, False, params)
    g.seed(seed)
    g.reset()
    out = {"N": N, "A": A, "steps": steps, "trace_every": trace_every, "W": W, "H": H, "seed": seed, "action_seed": action_seed,
           "scenario": np.array(scenario)}
    if params:
        out["param_keys"] = np.array(list(params.keys()))
        out["param_vals"] = np.array(list(params.values()), np.float32)
    for e in range(N):
        for k, v in snap_dict(g.snapshot(e), A).items():
            out[f"reset_{e}_{k}"] = v
    out["reset_obs"] = np.stack([g.get_observation(e, a).copy() for e in range(min(N, 4)) for a in range(A)])
    rewards, dones, trace = [], [], []
    for st in range(steps):
        masks = action_masks(sample_actions(action_seed, st, N * A))
        for e in range(N):
            for a in range(A):
                g.set_action_mask(e, a, int(masks[e * A + a]))
        g.step_norender()
        rewards.append(g.get_last_rewards().copy())
        dones.append(np.array([g.is_done(e) for e in range(N)], np.uint8))
        if (st + 1) % trace_every == 0:
            trace.append(np.stack([np.concatenate([np.asarray(g.snapshot(e)["agents"][a]["pos"]) for a in range(A)]) for e in range(N)]))
    out["rewards"] = np.stack(rewards)
    out["dones"] = np.stack(dones)
    out["trace_pos"] = np.stack(trace)
    for e in range(N):
        for k, v in snap_dict(g.snapshot(e), A).items():
            out[f"final_{e}_{k}"] = v
    g.render()
    out["final_obs"] = np.stack([g.get_observation(e, a).copy() for e in range(min(N, 4)) for a in range(A)])
    g.close()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "done; total reward", out["rewards"].sum(), "dones", int(out["dones"].sum()))


if __name__ == "__main__":
    only = sys.argv[1:]
    if only:   # python make_golden.py new  -> only the fixtures that do not exist yet
        _make = make
        def make(name, **kw):  # noqa: E306
            if not os.path.exists(os.path.join(HERE, name + ".npz")):
                _make(name, **kw)
    make("tower_a1", N=6, A=1, steps=1200, trace_every=100, W=64, H=64)
    make("tower_config0", N=2, A=1, steps=900, trace_every=100, W=128, H=128)   # BASELINE.json configs[0]'s shape: num_envs=2, one agent, 128 x 128
    make("tower_a4", N=3, A=4, steps=600, trace_every=100, W=64, H=64)
    make("tower_short_episodes", N=4, A=2, steps=400, trace_every=50, W=32, H=32, params={"episodeLengthSec": -220.0})
    make("obstacles_hard_a2", N=4, A=2, steps=500, trace_every=100, W=64, H=64, scenario="ObstaclesHard", seed=7)
    make("obstacles_easy_a1", N=6, A=1, steps=1200, trace_every=100, W=32, H=32, scenario="ObstaclesEasy", seed=11)
    make("collect_a2", N=5, A=2, steps=1100, trace_every=100, W=64, H=64, scenario="Collect", seed=3)
    make("rearrange_a4", N=4, A=4, steps=1000, trace_every=100, W=64, H=64, scenario="Rearrange", seed=9)
    os.environ.setdefault("BOXOBAN_LEVELS", os.path.join(HERE, "boxoban"))   # synthetic levels in the public Boxoban text format
    make("sokoban_a2", N=6, A=2, steps=1300, trace_every=100, W=64, H=64, scenario="Sokoban", seed=4, action_seed=6)
    make("hex_memory_a2", N=4, A=2, steps=900, trace_every=100, W=48, H=27, scenario="HexMemory", seed=8, action_seed=3, params={"episodeLengthSec": 4.0})
    make("hex_explore_a3", N=4, A=3, steps=500, trace_every=100, W=48, H=27, scenario="HexExplore", seed=13, action_seed=9, params={"episodeLengthSec": 10.0})
