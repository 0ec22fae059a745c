"""Env sharding across the GPUs of one node (one process per GPU, torch.distributed / RCCL).

The reference has no communication backend: multi-GPU there means independent OS processes pinned to
GPUs by Sample-Factory (megaverse_rl/runs/performance_benchmark_all_envs.py:5-7).  Envs are fully
independent (each Env owns its world and rng, bindings/megaverse.cpp:54-55), so the path shards by
contiguous env blocks with NO data-path collective.  What is exchanged, when asked for:
  * the per-step scalars (rewards [N*A] f32, dones [N] u8): one small all-gather;
  * optionally the observation slab, one RCCL all-gather into a caller-provided tensor, for a
    consumer that lives on a single GPU (numbers and why this is off by default: DESIGN.md).
Seeding is job-wide: MegaverseGym(..., env_offset, total_envs) draws the master seed stream for all
envs and keeps its slice, so the union of the shards is bit-identical to one big single-process gym.
"""
import numpy as np


def shard_range(rank, world_size, total_envs):
    """contiguous block partition, remainder spread over the first ranks -> (offset, count)"""
    base, rem = divmod(int(total_envs), int(world_size))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def env_seeds(master_seed, total_envs):
    """per-env seeds of MegaverseGym::seed (megaverse.cpp:60-69) for the whole job: a host
    restatement of mt19937 + libstdc++'s uniform_int_distribution (Lemire) used only by tests to
    check the sharded seeding; the library itself uses <random> directly (mv_api.hip: mv_seed)."""
    mt = np.zeros(624, np.uint64)
    mt[0] = np.uint64(master_seed & 0xFFFFFFFF)
    for i in range(1, 624):
        mt[i] = (np.uint64(1812433253) * (mt[i - 1] ^ (mt[i - 1] >> np.uint64(30))) + np.uint64(i)) & np.uint64(0xFFFFFFFF)
    state = {"mt": mt, "idx": 624}

    def twist():
        m = state["mt"]
        for k in range(624):
            y = (m[k] & np.uint64(0x80000000)) | (m[(k + 1) % 624] & np.uint64(0x7FFFFFFF))
            v = m[(k + 397) % 624] ^ (y >> np.uint64(1))
            if y & np.uint64(1):
                v ^= np.uint64(0x9908B0DF)
            m[k] = v
        state["idx"] = 0

    def nxt():
        if state["idx"] >= 624:
            twist()
        y = int(state["mt"][state["idx"]])
        state["idx"] += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    rng_range = 1 << 30
    out = []
    for _ in range(total_envs):
        product = nxt() * rng_range
        low = product & 0xFFFFFFFF
        if low < rng_range:
            threshold = ((1 << 32) - rng_range) % rng_range
            while low < threshold:
                product = nxt() * rng_range
                low = product & 0xFFFFFFFF
        out.append(product >> 32)
    return np.array(out, np.int64)


def all_gather_scalars(rewards, dones, group=None):
    """rewards: float32 tensor [n_local*A], dones: uint8 tensor [n_local]; equal shard sizes.
    Returns the job-wide (rewards, dones) on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    packed = torch.cat([rewards.view(torch.uint8).flatten(), dones.flatten()])
    out = torch.empty(world * packed.numel(), dtype=torch.uint8, device=packed.device)
    dist.all_gather_into_tensor(out, packed, group=group)
    out = out.view(world, packed.numel())
    nr = rewards.numel() * 4
    return out[:, :nr].contiguous().view(torch.float32).flatten(), out[:, nr:].contiguous().flatten()


def all_gather_observations(local_obs, out=None, group=None):
    """local_obs: uint8 [n_local*A, H, W, 4] -> job-wide slab [world*n_local*A, H, W, 4] (RCCL on GPUs)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * local_obs.shape[0],) + tuple(local_obs.shape[1:]), dtype=local_obs.dtype, device=local_obs.device)
    dist.all_gather_into_tensor(out, local_obs.contiguous(), group=group)
    return out
