"""helpers shared by the gpu parity tests"""
import numpy as np

import oracle_lib
from megaverse_amd.extension import MegaverseGym
from megaverse_amd.rollout import sample_actions, action_masks

STATE_FIELDS = [f for f in oracle_lib.SNAP.names]


def hip_snapshot(g, env):
    raw = g.debug_snapshot_bytes(env)
    assert raw.size == oracle_lib.SNAP.itemsize, (raw.size, oracle_lib.SNAP.itemsize)
    return raw.view(oracle_lib.SNAP)[0]


def diff_snapshots(a, b, num_agents):
    """list of human-readable differences between two snapshots (bitwise comparison)"""
    out = []
    for name in oracle_lib.SNAP.names:
        if name == "agents":
            for k in range(num_agents):
                for fn in oracle_lib.SNAP_AGENT.names:
                    x, y = np.asarray(a["agents"][k][fn]), np.asarray(b["agents"][k][fn])
                    if x.tobytes() != y.tobytes():
                        out.append(f"agent{k}.{fn}: {x} vs {y}")
        elif name == "boxes":
            n = int(a["num_boxes"])
            if a["boxes"][:n].tobytes() != b["boxes"][:n].tobytes():
                out.append(f"boxes: {a['boxes'][:n].tolist()} vs {b['boxes'][:n].tolist()}")
        elif name == "objects":
            n = int(a["num_objects"])
            if a["objects"][:n].tobytes() != b["objects"][:n].tobytes():
                out.append("objects differ")
        elif name == "terrain":
            n = int(a["num_terrain"])
            if a["terrain"][:n].tobytes() != b["terrain"][:n].tobytes():
                out.append(f"terrain: {a['terrain'][:n].tolist()} vs {b['terrain'][:n].tolist()}")
        elif name == "rewards":
            n = int(a["num_rewards"])
            if a["rewards"][:n].tobytes() != b["rewards"][:n].tobytes():
                out.append(f"rewards: {a['rewards'][:n].tolist()} vs {b['rewards'][:n].tolist()}")
        elif name == "chunk" and int(a["scenario"]) != 0:
            continue   # the Obstacles kernels work on the merged box list, there is no dense chunk in HBM
        else:
            x, y = np.asarray(a[name]), np.asarray(b[name])
            if x.tobytes() != y.tobytes():
                if name == "chunk":
                    idx = np.nonzero(x != y)[0]
                    out.append(f"chunk differs at {idx[:8].tolist()} ({idx.size} cells)")
                else:
                    out.append(f"{name}: {x} vs {y}")
    return out


def make_pair(num_envs, num_agents, w=128, h=128, seed=42, params=None, scenario="TowerBuilding"):
    og = oracle_lib.OracleGym(scenario, w, h, num_envs, num_agents, 1, False, params)
    hg = MegaverseGym(scenario, w, h, num_envs, num_agents, 1, False, params or {})
    og.seed(seed)
    hg.seed(seed)
    og.reset()
    hg.reset()
    return og, hg


def set_same_actions(og, hg, num_envs, num_agents, seed, step):
    acts = sample_actions(seed, step, num_envs * num_agents)
    masks = action_masks(acts)
    for e in range(num_envs):
        for a in range(num_agents):
            og.set_action_mask(e, a, int(masks[e * num_agents + a]))
    hg.set_actions_batched(acts)
    return acts
