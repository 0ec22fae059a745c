"""CPU: the N>1 path with world_size 2 on the gloo backend (no GPU): shard ranges, job-wide seeds,
the scalar all-gather and the observation all-gather assemble exactly the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from megaverse_amd import distributed


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_envs, agents, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        off, cnt = distributed.shard_range(rank, world, total_envs)
        seeds = distributed.env_seeds(42, total_envs)[off:off + cnt]
        # stand-in shard outputs derived from the global env index, as the HIP gym would produce
        env_ids = torch.arange(off, off + cnt)
        rewards = (env_ids.repeat_interleave(agents).float() * 0.5 + rank)
        dones = (env_ids % 3 == 0).to(torch.uint8)
        obs = (env_ids.repeat_interleave(agents).view(-1, 1, 1, 1) % 251).to(torch.uint8).expand(-1, 4, 8, 4).contiguous()
        r_all, d_all = distributed.all_gather_scalars(rewards, dones)
        o_all = distributed.all_gather_observations(obs)
        q.put((rank, seeds.tolist(), r_all.tolist(), d_all.tolist(), o_all[:, 0, 0, 0].tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_shards_compose():
    world, total, agents = 2, 12, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, agents, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    all_seeds = distributed.env_seeds(42, total).tolist()
    assert res[0][1] + res[1][1] == all_seeds
    ids = np.arange(total)
    want_r = np.concatenate([np.repeat(ids[:6], agents) * 0.5 + 0, np.repeat(ids[6:], agents) * 0.5 + 1]).tolist()
    for _, _, r_all, d_all, o_all in res:            # every rank sees the job-wide result, in env order
        assert r_all == pytest.approx(want_r)
        assert d_all == (ids % 3 == 0).astype(int).tolist()
        assert o_all == (np.repeat(ids, agents) % 251).tolist()
