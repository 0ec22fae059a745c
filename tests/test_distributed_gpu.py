"""The N > 1 path with REAL gyms on a one-GPU box: `python bench.py --gpus 2 --single-device` starts its own two ranks (the launcher the
driver's 8-GPU run goes through), each rank creates a real MegaverseGym shard on device 0 (env_offset / total_envs: contiguous env blocks,
job-wide seed and action streams), ObsGather runs its double-buffered pipeline -- over gloo, from host copies of the slabs: two ranks
cannot share one GPU under RCCL -- and --check-gather makes rank 0 replay the whole job in ONE gym of twice the size and compare its
observation slab with the gathered one, byte for byte.  Nothing is measured here; it takes the stand-in gym out of the only multi-rank
evidence an 8-GPU-less pool allows (tests/test_distributed_cpu.py runs the same code with a stand-in on CPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scenario,agents,mode,fmt,batch", [("TowerBuilding", 2, "allgather", "rgba", 0), ("ObstaclesHard", 1, "p2p", "rgba", 0), ("TowerBuilding", 1, "p2p", "rgb", 1),
                                                            ("TowerBuilding", 1, "allgather", "rgb", 4)])
def test_two_ranks_of_real_gyms_gather_the_single_gym_slab(hip, scenario, agents, mode, fmt, batch):
    """batch 0: what the library recommends (8 ticks per call at this size: the gather-on leg runs batched calls, one collective per call, 9 steps = a call of 8
    and a call of 1); 4: calls of 4, 4, 1; 1: tick by tick, one collective per tick"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MV_PIXEL_MODE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--check-gather", "--scenario", scenario,
                          "--agents", str(agents), "--envs-per-gpu", "12", "--obs", "48", "32", "--steps", "9", "--warmup", "4", "--no-cpu-baseline",
                          "--profile-steps", "0", "--no-extra-legs", "--gather", mode, "--gather-format", fmt, "--batch", str(batch)], capture_output=True, text=True, timeout=500, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["single_device"] is True and rec["config"]["gather_obs"] is True
    assert rec["gather_check"] is True, rec
    assert rec["gather"]["bytes_received_per_gpu_per_step"] == 12 * agents * 48 * 32 * (3 if fmt == "rgb" else 4)
    assert rec["value"] > 0 and rec["value_no_gather"] > 0
    # the line states what it ran under and its own efficiencies: per-GPU rate / the rate of rank 0 alone in the same process group
    assert rec["distributed"]["world_size"] == 2 and rec["distributed"]["backend"] == "gloo"   # (gloo: two ranks cannot share one GPU under RCCL)
    assert rec["value_solo_rank0"] > 0 and 0 < rec["value_efficiency"] and 0 < rec["value_no_gather_efficiency"]
    assert abs(rec["value_efficiency"] - rec["value"] / 2 / rec["value_solo_rank0"]) < 1e-9
    assert rec["gather"]["xgmi_bound_ms_per_step"] > 0
    assert rec["gather"]["ticks_per_collective"] == (batch if batch else 8) and rec["gather"]["collectives_per_call"] == 1
