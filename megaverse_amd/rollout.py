"""Counter-based random policy shared by the device sampler and the host.

``sample_actions(seed, step, agent_ids)`` returns exactly what the HIP kernel behind
``mv_sample_random_actions`` writes (megaverse_amd/csrc/mv_api.hip: sample_actions_kernel): i.i.d.
uniform per head over the sizes [3,3,3,2,2,3] (= action_space.sample() in the reference,
megaverse/megaverse_env.py:110-112), as a pure function of (seed, step, global agent index, head).
"""
import numpy as np

ACTION_SPACE_SIZES = (3, 3, 3, 2, 2, 3)
_M = np.uint64(0xFFFFFFFF)


def _fmix32(h):
    h = np.asarray(h, dtype=np.uint64) & _M
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M
    h ^= h >> np.uint64(16)
    return h


def sample_actions(seed, step, num_agents, agent_offset=0):
    """-> int32 [num_agents, 6] multi-discrete actions for global agents agent_offset.."""
    seed = np.uint64(int(seed) & 0xFFFFFFFF)
    step = np.uint64(int(step) & 0xFFFFFFFF)
    ids = (np.arange(num_agents, dtype=np.uint64) + np.uint64(agent_offset)) & _M
    s = _fmix32(seed ^ _fmix32((step + np.uint64(0x9E3779B9)) & _M))
    base = _fmix32(s ^ ((ids * np.uint64(0x85EBCA6B) + np.uint64(1)) & _M))
    out = np.empty((num_agents, 6), np.int32)
    for k, size in enumerate(ACTION_SPACE_SIZES):
        h = _fmix32((base + np.uint64(k) * np.uint64(0xC2B2AE35)) & _M)
        out[:, k] = ((h * np.uint64(size)) >> np.uint64(32)).astype(np.int32)
    return out


def sample_single_bit_masks(seed, step, num_agents, agent_offset=0):
    """-> int32 [num_agents] Action bitmasks of the reference's own benchmark policy, Action(1 << randRange(0, NumActions)) with
    NumActions = 11 (src/apps/megaverse_test_app.cpp:140-147); the device twin is mv_actions.h: sampled_single_bit_mask."""
    seed = np.uint64(int(seed) & 0xFFFFFFFF)
    step = np.uint64(int(step) & 0xFFFFFFFF)
    ids = (np.arange(num_agents, dtype=np.uint64) + np.uint64(agent_offset)) & _M
    s = _fmix32(seed ^ _fmix32((step + np.uint64(0x9E3779B9)) & _M))
    base = _fmix32(s ^ ((ids * np.uint64(0x85EBCA6B) + np.uint64(1)) & _M))
    h = _fmix32((base + np.uint64(6) * np.uint64(0xC2B2AE35)) & _M)
    return (np.int32(1) << ((h * np.uint64(11)) >> np.uint64(32)).astype(np.int32)).astype(np.int32)


def action_masks(actions):
    """multi-discrete [.., 6] -> Action bitmasks (reference: megaverse.cpp:100-116)."""
    a = np.asarray(actions, dtype=np.int64)
    masks = np.zeros(a.shape[:-1], np.int32)
    idx = 0
    for i, size in enumerate(ACTION_SPACE_SIZES):
        nz = a[..., i] > 0
        masks[nz] |= (1 << (idx + a[..., i][nz])).astype(np.int32)
        idx += size - 1
    return masks
