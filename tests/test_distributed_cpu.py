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


@pytest.mark.timeout(240)
@pytest.mark.parametrize("mode,fmt,batch", [("allgather", "rgba", 4), ("p2p", "rgba", 4), ("p2p", "rgb", 1), ("allgather", "rgb", 3)])
def test_bench_self_spawns_its_ranks_and_gathers_dry_run(mode, fmt, batch):
    """`python bench.py --gpus 2` (no torchrun around it, the way the driver runs N=1) must start its own ranks; --dry-run swaps
    the gym for a CPU stand-in and RCCL for gloo, everything else -- launcher, double-buffered gather pipeline, both timed legs,
    the JSON line -- is the code the GPU run executes.  The run itself asserts that every rank's shard arrived in rank order."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    # batch > 1: the gather-on leg steps in batched calls, ONE collective per call (the call's half of a ring of 2 x batch slabs; 9 steps = calls of 4, 4, 1 --
    # the last one short); batch 1: one collective per tick
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "9", "--warmup", "5", "--batch", str(batch),
                          "--envs-per-gpu", "4", "--obs", "8", "8", "--gather", mode, "--gather-format", fmt], capture_output=True, text=True, timeout=200, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert ("point-to-point" in rec["gather"]["collective"]) == (mode == "p2p")
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True and rec["steps"] == 9 and rec["scaling"] == "weak"
    assert rec["gather"]["ticks_per_collective"] == batch and rec["config"]["ticks_per_call"] == batch
    assert rec["gather"]["bytes_received_per_gpu_per_collective"] == batch * rec["gather"]["bytes_received_per_gpu_per_step"]
    assert rec["config"]["gather_obs"] is True and rec["value"] > 0 and rec["value_no_gather"] > 0
    assert rec["gather"]["bytes_received_per_gpu_per_step"] == 4 * 8 * 8 * (3 if fmt == "rgb" else 4)


def test_bench_single_rank_does_not_respawn():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run", "--steps", "3", "--warmup", "1", "--envs-per-gpu", "2",
                          "--obs", "8", "8"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["config"]["gather_obs"] is False and "value_no_gather" not in rec
