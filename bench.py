#!/usr/bin/env python
"""bench.py -- agent observations/sec of the batched TowerBuilding step() on MI355X.

Contract (driver): python bench.py --gpus N --steps K --warmup W.  Prints ONE JSON line on rank 0.
N > 1: one rank per GPU over RCCL.  Launched through torch.distributed.run the ranks are taken from the
environment; launched plainly (`python bench.py --gpus 8`) the script re-executes itself under
torch.distributed.run on 127.0.0.1 with a free port.

A "step" is one pass of the hot path over one batch: sample random actions (device, counter-based) -> step
kernel (physics + scenario logic + auto-reset of finished episodes) -> frame setup -> frame sort -> raster kernel
(128x128 RGBA8 first-person observation per agent written into the HBM slab).
Workload at N=1: BASELINE.json configs[1] = TowerBuilding, num_envs=1024, num_agents_per_env=1, obs 128x128.
N>1: weak scaling, 1024 envs per GPU, envs sharded by contiguous blocks with job-wide seeds (a sharded run
simulates exactly the envs the single-process run of N*1024 would) and -- north_star's layout -- ONE data-path
collective: the RCCL all-gather of the observation slabs, issued on a communication stream -- ONE collective
per batched call: a call of k ticks (mv_step_n, the same launches as the N=1 headline) renders into one half of a
ring of 2k slabs while the other half, the previous call's k slabs, travels (--batch 1: one tick and one collective
at a time).  `value` is the rate WITH the gather; `value_no_gather` (same line, the identical calls) is the rate
when every GPU's consumer reads its own shard (the reference's multi-GPU mode).

Timing: the K timed steps carry no instrumentation.  The per-kernel figures of `roofline` / `roofline_physics`
come from a second, untimed loop with HIP events on the gym's stream (mv_profile_begin).
"""
import argparse
import json
import os
import socket
import sys
import time

# HIP deals a process's streams over 4 hardware queues unless told otherwise; a gym holds up to six (the caller's, simulation, copy, episode draws, two pass
# streams) and the transparency legs make more gyms: streams that land on one queue serialise (r11a / r11b: the legs that make NEW gyms after the main one ran at half
# their rate -- MegaverseEnv.step_device 7.5 M obs/s against 16.7 M -- until the process had 8 queues; the main loop's rate does not move).  Before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "agent observations/sec (whole node), TowerBuilding 128x128 obs, random policy"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
XGMI_PEAK_GBS = 7 * 153.0      # 7 point-to-point links x ~153 GB/s per GPU
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2   # wave64 VALU instructions per second: 1024 SIMD-32s, 2 cycles per instruction (MI355X_MICROARCH.md)


def kernel_sources_sha16():
    """sha256 (first 16 hex digits) of the kernel sources: profiles/pmc_traffic.json carries the hash of the tree its counter passes ran on, and
    the bench line prints counter-derived figures only for the very same sources (the GPU box has no .git to ask for a revision)"""
    import hashlib
    d = os.path.join(ROOT, "megaverse_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def host_threads():
    """-> (threads the CPU baseline may use, note): the CPUs this process may run on, capped by the container's CPU-time quota (cgroup cpu.max /
    cfs_quota_us): a box that shows 256 CPUs under a quota of 16 CPUs' time runs 256 pinned threads SLOWER than one (measured, r04s: physics
    309 k steps/s on 1 thread, 2.8 M on 16, 94 k on 256)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"{n} CPUs visible"
    quota = None
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            quota = float(a) / float(b)
    except Exception:  # noqa: BLE001
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if q > 0:
                quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except Exception:  # noqa: BLE001
            pass
    if quota:
        note += f", cgroup CPU quota {quota:g}"
        n = max(1, min(n, int(quota + 0.5)))
    return max(1, n), note


def cpu_baseline(scenario, obs_w, obs_h, n_env, agents, policy="multidiscrete", full=False):
    """Oracle (CPU restatement, kind 'port') timed on this box's host cores on bounded samples of the same workload: same scenario /
    obs size / seed / action stream.  The oracle runs as BASELINE.md 3 plans it: the reference's persistent worker pool (vector_env.cpp:16-40,
    71-87: static block partition, the caller takes block 0 and spins on an atomic count), threads pinned one per allowed CPU (MVO_PIN=1),
    the tile-culled software raster (mvo_set_raster(1): the brute-force checker's image byte for byte, tests/test_oracle_properties.py).
    Legs, each the median of three repetitions after 20 warm-up ticks, each bounded in time so that the default bench run stays within minutes:
      all_threads        the full step (physics + logic + auto-reset + software raster) of all n_env envs on every host thread -- this is `value`;
      one_thread         the same on ONE pinned thread, on the first few envs of the batch (per-core figure);
      physics_only       mvo_step_norender on all threads; physics_one_thread the same on one thread, all n_env envs: their ratio is printed as
                         physics_scaling = rate(all) / (threads x rate(1)) (the serial auto-reset section and the memory system bound it);
      raster_only        mvo_render on all threads (a software raster -- NOT the reference's GL renderer; a baseline, not a target).
    Actions go in through ONE batched call per tick.
    full (--cpu-baseline-full): the all-threads leg exactly as BASELINE.md section 3 states it -- 100 warm-up steps, 2 000 measured steps, median of 5 runs,
    no time bound (minutes of CPU work); the default is the bounded sample the bench contract asks for (seconds), and the `sample` string says which ran."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    from megaverse_amd.rollout import sample_actions, action_masks, sample_single_bit_masks
    os.environ["MVO_PIN"] = "1"
    threads, quota_note = host_threads()

    def masks_of(step, n):
        if policy == "single-bit":
            return sample_single_bit_masks(1234, step, n * agents)
        return action_masks(sample_actions(1234, step, n * agents))

    def leg(n, T, what, budget_s, reps=3, pin=None, warm=20, max_steps=700):
        old = None
        if pin is not None and hasattr(os, "sched_setaffinity"):
            old = os.sched_getaffinity(0)
            os.sched_setaffinity(0, {sorted(old)[pin % len(old)]})
        try:
            g = oracle_lib.OracleGym(scenario, obs_w, obs_h, n, agents, T)
            g.set_raster(True)
            g.seed(42)
            g.reset()
            masks = [masks_of(st, n) for st in range(warm + 64)]   # (drawn ahead: the numpy action sampler is not part of what is timed)
            for st in range(warm):
                g.set_action_masks(masks[st]); g.step_norender()
            rates, steps_total, t_total, st = [], 0, 0.0, warm
            for _ in range(reps):
                steps, t0 = 0, time.perf_counter()
                while True:
                    if what != "raster":
                        g.set_action_masks(masks[warm + (st % 64)])
                    if what == "full":
                        g.step()
                    elif what == "physics":
                        g.step_norender()
                    else:
                        g.render()
                    steps += 1
                    st += 1
                    el = time.perf_counter() - t0
                    if (budget_s is not None and el > budget_s / reps) or steps >= max_steps:
                        break
                rates.append(n * agents * steps / el)
                steps_total += steps
                t_total += el
            g.close()
        finally:
            if old is not None:
                os.sched_setaffinity(0, old)
        return {"value": float(np.median(rates)), "unit": "agent observations/sec" if what != "physics" else "agent steps/sec", "threads": T, "envs": n,
                "steps": steps_total, "seconds": round(t_total, 2), "median_of": reps, "warmup_ticks": warm}

    n1 = max(1, min(n_env, 4))
    legs = {"all_threads": leg(n_env, threads, "full", None, reps=5, warm=100, max_steps=2000) if full else leg(n_env, threads, "full", 8.0),
            "one_thread": leg(n1, 1, "full", 4.0, pin=0),
            "physics_only": leg(n_env, threads, "physics", 3.0),
            "physics_one_thread": leg(n_env, 1, "physics", 3.0, pin=0),
            "raster_only": leg(n_env, threads, "raster", 4.0)}
    a = legs["all_threads"]
    out = {"value": a["value"], "unit": "agent observations/sec", "cores": threads, "kind": "port",
           "physics_scaling": legs["physics_only"]["value"] / (threads * legs["physics_one_thread"]["value"]),
           "sample": f"oracle (CPU restatement of VectorEnv::step with a tile-culled software raster; NOT the reference binary) {scenario} num_envs={n_env} agents={agents} "
                     f"obs {obs_w}x{obs_h}, policy {policy}: {a['steps']} steps in {a['seconds']} s on {threads} pinned threads ({quota_note}; persistent worker pool, static block "
                     f"partition, caller participates: vector_env.cpp:16-40,65-87), {a['warmup_ticks']} warm-up ticks, median of {a['median_of']}"
                     f"{' (BASELINE.md section 3 in full: --cpu-baseline-full)' if full else ' (a bounded sample: the plan of BASELINE.md section 3 in full -- 100 warm-up, 2 000 steps, median of 5 -- runs with --cpu-baseline-full)'}; "
                     f"one_thread = {n1} envs pinned to one core"}
    out.update(legs)
    return out


def cpu_baseline_mixed(scenarios, obs_w, obs_h, n_env, agents):
    """configs[4]: the oracle has one scenario per gym, like the reference (megaverse_env.py:27-39): one oracle gym per scenario with
    n_env / len(scenarios) envs each, stepped one after the other on all host threads; median of three repetitions of the whole round"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    from megaverse_amd.rollout import sample_actions, action_masks
    threads, _ = host_threads()
    per = n_env // len(scenarios)
    gyms = []
    for name in scenarios:
        g = oracle_lib.OracleGym(name, obs_w, obs_h, per, agents, threads)
        g.set_raster(True)   # (the tile-culled software raster: the brute-force checker's image, several times faster)
        g.seed(42)
        g.reset()
        gyms.append(g)
    rates, st, steps_total, t_total = [], 0, 0, 0.0
    for _ in range(3):
        steps, t0 = 0, time.perf_counter()
        while True:
            masks = action_masks(sample_actions(1234, st, per * agents))
            for g in gyms:
                g.set_action_masks(masks)
                g.step()
            steps += 1
            st += 1
            el = time.perf_counter() - t0
            if el > 4.0 or steps >= 400:
                break
        rates.append(per * len(scenarios) * agents * steps / el)
        steps_total += steps
        t_total += el
    for g in gyms:
        g.close()
    return {"value": float(np.median(rates)), "unit": "agent observations/sec", "cores": threads, "kind": "port",
            "sample": f"oracle (CPU restatement, software raster; NOT the reference binary): {len(scenarios)} gyms ({', '.join(scenarios)}) x {per} envs, agents={agents}, "
                      f"obs {obs_w}x{obs_h}, stepped one after the other on {threads} threads: {steps_total} rounds in {t_total:.1f} s, median of 3"}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` -> the same command line, one rank per GPU, through torch.distributed.run"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


class ObsGather:
    """All-gather of the observation slab, double buffered: step t renders into slab t % 2; the collective for slab b runs on a
    communication stream (ordered after the raster by an event) while the compute stream goes on with step t + 1 into the other
    slab; before slab b is rendered into again the compute stream waits for its gather.  CPU / gloo (dry run): same bookkeeping
    with async work handles."""

    def __init__(self, dist, torch, local_shape, world, device, cuda, mode="allgather", channels=4, local=None):
        self.dist, self.torch, self.cuda, self.mode, self.world = dist, torch, cuda, mode, world
        self.rank = dist.get_rank() if world > 1 else 0
        self.n_local = local_shape[0]
        # local: the two buffers the gym renders into, handed in -- the two halves of a batched call's output ring, k slabs each: ONE collective per call of k
        # ticks
        self.local = local if local is not None else [torch.zeros(local_shape, dtype=torch.uint8, device=device) for _ in range(2)]
        # channels = 3: what travels is R, G, B (alpha is 255 everywhere): a packing kernel on the communication stream, 3/4 of the link traffic
        self.channels = channels
        sent_shape = tuple(local_shape[:-1]) + (channels,)
        self.sent = self.local if channels == 4 else [torch.zeros(sent_shape, dtype=torch.uint8, device=device) for _ in range(2)]
        self.out = [torch.zeros((world * local_shape[0],) + sent_shape[1:], dtype=torch.uint8, device=device) for _ in range(2)]
        self.work = [None, None]
        if cuda:
            self.comm = torch.cuda.Stream(device=device)
            self.rendered = [torch.cuda.Event() for _ in range(2)]
            self.gathered = [None, None]

    def before_render(self, b):
        """slab b is about to be overwritten: its previous gather must have read it"""
        if self.cuda:
            if self.gathered[b] is not None:
                self.torch.cuda.current_stream().wait_event(self.gathered[b])
        elif self.work[b] is not None:
            for w in self.work[b]:
                w.wait()
            self.work[b] = None

    def _collective(self, b):
        """all_gather_into_tensor (RCCL picks the algorithm: a ring is bound by ONE xGMI link), or -- mode "p2p" -- the direct shape SURVEY.md
        8e asks for: every GPU sends its shard to each of its peers and receives theirs, one grouped batch of point-to-point operations
        (ncclGroupStart / ncclSend / ncclRecv under RCCL), so that all 7 links of the fully connected node carry one shard each"""
        if self.channels != 4:
            self.sent[b].copy_(self.local[b][..., :self.channels])
        if self.mode != "p2p":
            return [self.dist.all_gather_into_tensor(self.out[b], self.sent[b], async_op=True)]
        n, out = self.n_local, self.out[b]
        out[self.rank * n:(self.rank + 1) * n].copy_(self.sent[b], non_blocking=True)
        ops = []
        for k in range(1, self.world):   # (staggered peers: rank r sends to r + k while it receives from r - k)
            to, frm = (self.rank + k) % self.world, (self.rank - k) % self.world
            ops.append(self.dist.P2POp(self.dist.isend, self.sent[b], to))
            ops.append(self.dist.P2POp(self.dist.irecv, out[frm * n:(frm + 1) * n], frm))
        return self.dist.batch_isend_irecv(ops)

    def after_render(self, b):
        if self.cuda:
            cur = self.torch.cuda.current_stream()
            self.rendered[b].record(cur)
            with self.torch.cuda.stream(self.comm):
                self.comm.wait_event(self.rendered[b])
                for w in self._collective(b):
                    w.wait()   # orders the communication stream after the collective; the host does not block
                ev = self.torch.cuda.Event()
                ev.record(self.comm)
                self.gathered[b] = ev
        else:
            self.work[b] = self._collective(b)

    def drain(self):
        if self.cuda:
            self.comm.synchronize()
        else:
            for b in range(2):
                if self.work[b] is not None:
                    for w in self.work[b]:
                        w.wait()
                    self.work[b] = None


class DryGym:
    """--dry-run stand-in for the gym (no device): fills the slab with a (rank, step) pattern so that the launcher, the gather
    pipeline and the JSON line can be exercised on CPU with gloo (tests/test_distributed_cpu.py)."""

    def __init__(self, rank):
        self.rank, self.buf, self.i = rank, None, 0

    def set_obs_tensor(self, t): self.buf = t
    def sample_random_actions(self, seed, i): self.i = i
    def step(self): self.buf.fill_((self.rank * 31 + self.i) % 251)
    def close(self): pass
    # batched calls into an output ring (the gather-on leg of N > 1 runs them like the N = 1 headline: one collective per call)
    def recommended_ticks_per_call(self): return 4
    def recommended_pass_overlap(self): return False
    def set_ring_tensor(self, ring): self.ring, self.tick = ring, 0

    def step_n(self, k, policy, seed, first):
        for j in range(k):
            self.ring[self.tick % self.ring.shape[0]].fill_((self.rank * 31 + first + j) % 251)
            self.tick += 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=1024)
    ap.add_argument("--agents", type=int, default=1)
    ap.add_argument("--scenario", default="TowerBuilding",
                    help="TowerBuilding (headline), Obstacles{Easy,Medium,Hard,Walls,Steps,Lava}, Collect, Rearrange, Sokoban, HexMemory, HexExplore, Empty, or Mixed (configs[4])")
    ap.add_argument("--obs", type=int, nargs=2, default=[128, 128], metavar=("W", "H"))
    ap.add_argument("--no-gather-obs", action="store_true", help="N>1: skip the gather-on leg (value = the no-gather rate)")
    ap.add_argument("--gather", default="allgather", choices=["allgather", "p2p"],
                    help="N>1: how the observation slabs are assembled: one all_gather_into_tensor (RCCL's choice of algorithm) or grouped point-to-point "
                         "sends / receives, one shard per peer link (the fully connected xGMI shape)")
    ap.add_argument("--gather-format", default="rgb", choices=["rgba", "rgb"],
                    help="N>1: what the gather moves: the slab as rendered, or R, G, B packed on the communication stream (alpha is 255 everywhere): 3/4 of the bytes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="the CPU baseline's all-threads leg as BASELINE.md section 3 states it: 100 warm-up steps, 2 000 measured steps, median of 5 runs (minutes); default: a bounded sample")
    ap.add_argument("--pixels", default="fast", choices=["fast", "exact"], help="observation arithmetic (DESIGN.md 'pixel tolerance')")
    ap.add_argument("--policy", default="multidiscrete", choices=["multidiscrete", "single-bit"],
                    help="random policy: uniform per head (action_space.sample(), megaverse_env.py:110-112) or the reference benchmark's "
                         "Action(1 << randRange(0, 11)) (megaverse_test_app.cpp:140-147)")
    ap.add_argument("--pass-overlap", choices=("auto", "on", "off"), default="auto",
                    help="output ring two calls deep, the observation passes of consecutive calls overlap (mv_set_pass_overlap).  auto: what mv_recommended_pass_overlap says -- "
                         "every scenario but Empty; TowerBuilding with one or two agents per env from 512 frames per tick on (measured r10za/b/c, M obs/s with / without: "
                         "TowerBuilding 1024 envs 34.4 / 32.5, 512 envs 26.9 / 25.4, 512 x 4 agents 24.5 / 28.8; Rearrange 28.6 / 25.6; Collect 16.9 / 16.6; HexMemory 9.8 / 9.5; "
                         "in round 4, r07: TowerBuilding 23.4 / 24.0 -- the passes were not yet what a call waits for)")
    ap.add_argument("--batch", type=int, default=0,
                    help="ticks per stepping call (mv_step_n): every tick is stepped and rendered in full, the two stream hand-overs are paid once "
                         "per call; 1 = one mv_step per tick; 0 (default) = 16 (8 where a call's 16 observation slabs would exceed ~1 GB), the first calls after a synchronisation 2, 4 and 6 ticks (the observation "
                         "passes of a call start when its ticks are stepped: short first calls fill the pipeline sooner; 20-step runs: 20.0-20.25 M obs/s "
                         "against 19.0 M with 2 ticks per call throughout).  N>1: calls of that many ticks throughout, with the gather on one collective per call (the call's half of a ring of 2 x batch slabs)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the single-step / unpipelined / closed-loop transparency legs")
    ap.add_argument("--profile-steps", type=int, default=256, help="steps of the untimed per-kernel profile loop (HIP events on the gym's stream)")
    ap.add_argument("--dry-run", action="store_true", help="CPU / gloo run of the launcher + gather pipeline with a stand-in gym")
    ap.add_argument("--single-device", action="store_true",
                    help="N>1 on a box with ONE GPU (tests): every rank runs its real env shard on device 0, the observation slabs are gathered over gloo from "
                         "host copies -- the launcher, the sharding (env_offset / total_envs) and ObsGather as in a real run, no xGMI")
    ap.add_argument("--check-gather", action="store_true",
                    help="after the timed loop rank 0 replays the whole job in ONE gym of N x envs-per-gpu envs and compares its slab with the gathered one")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args.gpus)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dry = args.dry_run
    single = args.single_device and not dry
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if single:
        local_rank = 0
    device = "cpu" if dry else f"cuda:{local_rank}"
    if not dry:
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry or single:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(device))

    if args.scenario.lower() in ("sokoban", "mixed"):   # level files: $BOXOBAN_LEVELS, else the synthetic Boxoban-format set the tests use
        os.environ.setdefault("BOXOBAN_LEVELS", os.path.join(ROOT, "tests", "golden", "boxoban"))
    W, H = args.obs
    n_env, A = args.envs_per_gpu, args.agents
    mixed = args.scenario.lower() in ("mixed", "mixed4")
    frames = n_env * A
    # ticks per stepping call: 16 -- one tail of the observation launch per 16 ticks -- where the observation passes are what a call waits for, 8 where the step
    # launch is (it grows per tick with the call's length).  Measured, 16 against 8 (M obs/s, `profiles/r08y_*`, `r08p`): TowerBuilding 1024 envs 28.5 / 26.8,
    # ObstaclesHard 1024 24.5 / 22.7, Rearrange 23.7 / 21.5, Collect 15.0 / 14.5; 512 envs 16.2 / 19.3, ObstaclesHard 512 13.2 / 18.8, Sokoban 19.9 / 25.9, 512
    # envs x 4 agents 19.5 / 26.0, 4096 envs 28.4 / 30.5.  Hence: 1024 ... 2047 frames, not Sokoban. The rule itself lives behind the ABI
    # (mv_recommended_ticks_per_call, include/megaverse_hip.h): --batch 0 asks the library, after the gym exists; an explicit --batch sizes the gym's slot
    # groups for it (MV_PIPE_BATCH, read by mv_create).
    if args.batch > 0:
        os.environ.setdefault("MV_PIPE_BATCH", str(max(8, min(16, args.batch))))
    if dry:
        gym = DryGym(rank)
    elif mixed:   # BASELINE.json configs[4]: the eight MEGAVERSE8 scenarios dealt round-robin by env index
        from megaverse_amd.multitask import MEGAVERSE_IN_SCOPE, MultiTaskGym
        # --scenario Mixed: the reference's eight-scenario multi-task set (megaverse_env.py:11-20); Mixed4: the four BASELINE.md section 3 row 5 names
        mixed_set = ["TowerBuilding", "ObstaclesEasy", "ObstaclesHard", "Collect"] if args.scenario.lower() == "mixed4" else MEGAVERSE_IN_SCOPE
        gym = MultiTaskGym(mixed_set, W, H, n_env, A, 0, {}, device=local_rank, env_offset=rank * n_env, total_envs=world * n_env)
        gym.set_pixel_mode(args.pixels)
    else:
        from megaverse_amd.extension import MegaverseGym
        # 0 = episode-feeder threads: this rank's share of the host's cores (host-generated scenarios)
        gym = MegaverseGym(args.scenario, W, H, n_env, A, 0, False, {},
                           device=local_rank, env_offset=rank * n_env, total_envs=world * n_env)
        gym.set_stream(torch.cuda.current_stream().cuda_stream)
        gym.set_pixel_mode(args.pixels)
        gym.set_sample_policy(args.policy)

    if dry:
        batch = args.batch if args.batch > 0 else 8
    else:
        batch = min(args.batch, int(os.environ["MV_PIPE_BATCH"])) if args.batch > 0 else gym.recommended_ticks_per_call()

    do_gather = world > 1 and not args.no_gather_obs
    # batched stepping: tick j of a call renders into slab j of a ring, so that all `batch` observations of a call exist side by side when it
    # is done (a k-step rollout buffer) -- the working set of the observation writes is batch x one slab, not one slab written over and over
    batched = batch > 1 and (not dry or world > 1)
    # N > 1 with the gather on: the SAME batched calls as the N = 1 headline -- a call renders its k ticks into one half of a ring of 2 k
    # slabs while the collective for the other half (the previous call's k slabs) runs on the communication stream: one collective per call, not one per tick
    gather_batched = do_gather and batched and not mixed
    # (slabs of the output ring: a call never holds more ticks than `batch`; two calls deep, the passes of consecutive calls overlap: mv_set_pass_overlap)
    # (the rule: mv_recommended_pass_overlap)
    # -- for a region of at least four full calls: a region that starts from an empty pipeline and ends after a handful of short calls (the driver's 20-step form)
    # gains nothing from a second pass stream and, box by box, loses up to 8 % to it (r10zc / r11b, on / off: 25.5 25.3 25.6 23.8 / 25.3 25.1 25.1 25.4, and 23.5
    # 23.5 23.6 23.1 23.8 24.5 / 25.3 25.4 25.3 25.3 25.1 25.1 M obs/s)
    pass_overlap = batched and not mixed and (args.pass_overlap == "on" or (args.pass_overlap == "auto" and gym.recommended_pass_overlap() and args.steps >= 4 * batch))
    if gather_batched:
        pass_overlap = False   # (the ring's two halves belong to the gather pipeline)
    ring_slots = 2 * batch if gather_batched else (max(batch, 8) * (2 if pass_overlap else 1)) if batched else 1
    ring = torch.zeros((ring_slots, frames, H, W, 4), dtype=torch.uint8, device=device) if batched and not mixed else None   # (Mixed: one slab, no ring)
    gather = None
    if world > 1:
        gdev, gch = "cpu" if single else device, 3 if args.gather_format == "rgb" else 4
        if gather_batched:   # a buffer = one half of the ring: `batch` slabs (single-device tests: host copies of the halves are what gloo gathers)
            halves = None if single else [ring[h * batch:(h + 1) * batch].view(batch * frames, H, W, 4) for h in range(2)]
            gather = ObsGather(dist, torch, (batch * frames, H, W, 4), world, gdev, cuda=not dry and not single, mode=args.gather, channels=gch, local=halves)
        else:
            gather = ObsGather(dist, torch, (frames, H, W, 4), world, gdev, cuda=not dry and not single, mode=args.gather, channels=gch)
    if gather and not gather_batched and not single:
        slabs = gather.local
    elif gather and single and not gather_batched:   # the gym renders into device slabs; their host copies are what gloo gathers
        slabs = [torch.zeros((frames, H, W, 4), dtype=torch.uint8, device=device) for _ in range(2)]
    else:
        slabs = [torch.zeros((frames, H, W, 4), dtype=torch.uint8, device=device)]
    # (rewards and dones get rings, too: a k-step rollout buffer holds all three -- and overlapped passes require it, include/megaverse_hip.h)
    ring_rew = torch.zeros((ring_slots, frames), dtype=torch.float32, device=device) if ring is not None else None
    ring_done = torch.zeros((ring_slots, n_env), dtype=torch.uint8, device=device) if ring is not None else None

    def set_ring():   # (also restarts the ring at entry 0: a region of batched gather calls begins with half 0)
        if dry:
            gym.set_ring_tensor(ring)
        elif mixed:   # one set of rings per scenario (MultiTaskGym.set_output_ring): a batched group call is then two launches
            gym.set_output_ring(ring_slots)
        else:
            gym.set_output_ring(ring_slots, ring.data_ptr(), ring_rew.data_ptr(), ring_done.data_ptr())

    def bind(b):
        if dry:
            gym.set_obs_tensor(slabs[b])
        elif mixed:
            gym.attach_tensor(slabs[b])
        else:
            gym.set_obs_buffer(slabs[b].data_ptr())

    bind(0)
    if not dry:
        gym.seed(42)
        gym.reset()

    def one_step(i, with_gather):
        b = i & 1 if with_gather else 0
        if with_gather:
            gather.before_render(b)
            bind(b)
        gym.sample_random_actions(1234, i)
        gym.step()
        if with_gather:
            if mixed and not gym.union:
                gym.synchronize()   # (round-2 scheme, one stream per scenario: join them before the collective reads the slab)
            if single:
                gather.local[b].copy_(slabs[b])   # (device -> host on the gym's stream, synchronous)
            gather.after_render(b)

    def fence():
        if not dry:
            torch.cuda.synchronize()
        if gather:
            gather.drain()
        if world > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    last_call = {}   # the last gathered call of a batched gather region: ring half, ticks, first step index

    def run_steps(first, n, with_gather, use_batch):
        if use_batch and with_gather:
            # one collective per CALL: call c renders its k ticks into half c & 1 of the ring (entries (c & 1) k ...: the ring restarts with every region) while
            # the collective of the previous call's half runs on the communication stream; the step / observation launches are the N = 1 headline's
            set_ring()
            i = c = 0
            while i < n:
                kk = min(batch, n - i)
                h = c & 1
                gather.before_render(h)
                gym.step_n(kk, args.policy, 1234, first + i)
                if single:
                    gather.local[h].copy_(ring[h * batch:(h + 1) * batch].view(batch * frames, H, W, 4))   # (device -> host on the gym's stream, synchronous)
                gather.after_render(h)
                last_call.update(half=h, ticks=kk, first=first + i)
                i += kk
                c += 1
            return
        if use_batch and not with_gather:
            i = 0
            # A region starts right after a synchronisation, with an empty pipeline: its first observation pass can only start when the first call's
            # ticks are stepped, so the first calls are short -- 2 ticks, then 4, then 6, then --batch (measured on 20-step runs, three each, r07c:
            # 2,4,6: 20.8-21.0 M obs/s; 1,3: 20.3-21.6; 1,3,4,4: 20.9; 3,8: 20.3-20.9; 2,2,4,4: 20.1-20.2; r05k: 19.0-19.1 M with 2 ticks per call throughout
            # and 19.2-19.7 M with 8).  MV_BENCH_CALL_SCHEDULE=a,b,...: other first calls.
            # (N > 1: calls of `batch` ticks throughout, like the gather-on leg)
            sched = [int(x) for x in os.environ.get("MV_BENCH_CALL_SCHEDULE", "2,4,6" if world == 1 else "").split(",") if x]
            while i < n:
                k = min(sched.pop(0) if sched else batch, n - i, max(batch, 8))
                gym.step_n(k, args.policy, 1234, first + i)
                i += k
        else:
            for i in range(n):
                one_step(first + i, with_gather)

    def timed(first, with_gather, use_batch):
        fence()
        t0 = time.perf_counter()
        run_steps(first, args.steps, with_gather, use_batch)
        fence()
        el = time.perf_counter() - t0
        t = torch.tensor([el], dtype=torch.float64, device="cpu" if (dry or single) else device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    step0 = 0
    main_batched = batched and (not do_gather or gather_batched)
    if main_batched and not gather_batched and (ring is not None or mixed):
        set_ring()
        if pass_overlap:
            gym.set_pass_overlap(True)
    run_steps(step0, args.warmup, do_gather, main_batched)
    step0 += args.warmup
    elapsed = timed(step0, do_gather, main_batched)        # THE timed region: exactly --steps steps, no instrumentation
    step0 += args.steps
    last_gathered = step0 - 1
    elapsed_no_gather = None
    if do_gather:                            # second leg, same step count, observations stay on the producing GPU
        bind(0)
        if batched and (ring is not None or mixed):
            set_ring()
        elapsed_no_gather = timed(step0, False, batched)
        step0 += args.steps
    # N > 1: the one-GPU rate measured INSIDE this process group -- rank 0 steps alone while the other ranks wait at the barrier -- so that the
    # line can state its own efficiencies (per-GPU rate / solo rate) instead of leaving them to a comparison with another run
    elapsed_solo = None
    if world > 1 and not dry:
        fence()
        if rank == 0:
            t0 = time.perf_counter()
            run_steps(step0, args.steps, False, batched)
            torch.cuda.synchronize()
            elapsed_solo = time.perf_counter() - t0
        step0 += args.steps
        fence()

    # ---- per-kernel profile: a separate, untimed loop with HIP events around every kernel, each interval on one stream
    prof = prof_alone = None
    if not dry and args.profile_steps > 0:
        fence()
        gym.profile_begin(args.profile_steps)
        run_steps(step0, args.profile_steps, False, batched)
        prof = gym.profile_end()
        if mixed:
            prof = prof[0]   # (the union launches are timed on the group leader's events)
        step0 += args.profile_steps
        # the same launches ALONE on the chip (the physics kernel's figures "alone"): pipelining off -- a call's step launches, then its observation
        # launch, one after the other on the caller's stream --, the same calls, the same events
        if batched and not mixed and world == 1 and gym.pipelining():
            gym.set_pipelining(False)
            fence()
            na = min(args.profile_steps, 4 * batch)
            gym.profile_begin(na)
            run_steps(step0, na, False, batched)
            prof_alone = gym.profile_end()
            step0 += na
            gym.set_pipelining(True)
            fence()
    mixed_ring_checksum = 0
    if batched and mixed and getattr(gym, "ring_obs", None):
        mixed_ring_checksum = sum(int(r[:, ::97].to(torch.int64).sum().item()) for r in gym.ring_obs)
    if batched and not dry and (ring is not None or mixed):
        if pass_overlap:
            gym.set_pass_overlap(False)   # (the transparency legs below run without it, and without its two streams)
        gym.set_output_ring(0)
        bind(0)

    # ---- transparency legs (N = 1), same step count each, none of them is `value`:
    #   single_step   one mv_step per tick (the r02 headline mode: every tick hands over between the two streams);
    #   unpipelined   step kernel and raster back to back on one stream;
    #   closed_loop   a policy in the loop: a device-side policy reads the observations of tick t and produces the actions of tick t + 1
    #                 (torch ops on the gym's stream + mv_set_actions_device): nothing can overlap, this is what an RL learner gets
    extra = {}
    pipelined = bool(not dry and (gym.gyms[0].pipelining() if mixed else gym.pipelining()))
    if mixed and world == 1 and not args.no_extra_legs and batched:   # Mixed: one group step per tick (the hand-overs paid every tick)
        run_steps(step0, min(args.warmup, 20), False, False); step0 += min(args.warmup, 20)
        extra["single_step"] = timed(step0, False, False); step0 += args.steps
    if not dry and not mixed and world == 1 and not args.no_extra_legs:
        wu = min(args.warmup, 20)
        if batched and pipelined:
            run_steps(step0, wu, False, False); step0 += wu
            extra["single_step"] = timed(step0, False, False); step0 += args.steps
        if pipelined:
            gym.set_pipelining(False)
            run_steps(step0, wu, False, False); step0 += wu
            extra["unpipelined"] = timed(step0, False, False); step0 += args.steps
            gym.set_pipelining(True)
        sizes = torch.tensor([3, 3, 3, 2, 2, 3], dtype=torch.int32, device=device)
        acts = torch.zeros((frames, 6), dtype=torch.int32, device=device)
        feat = slabs[0].view(frames, -1)[:, 37:37 + 6 * 97:97]   # six bytes of every frame

        def policy_step(i):
            torch.remainder(feat, sizes, out=acts)   # "policy": ONE kernel, a function of the observation just rendered (uint8 pixels -> int32 actions)
            gym.set_actions_device(acts.data_ptr())
            gym.step()

        for i in range(wu):
            policy_step(step0 + i)
        step0 += wu
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            policy_step(step0 + i)
        host_enqueue_ms_loop = (time.perf_counter() - t0) / args.steps * 1e3
        fence()
        extra["closed_loop"] = time.perf_counter() - t0
        step0 += args.steps

        # closed_loop_double_buffered: the same policy-in-the-loop, with the envs in two halves that take turns (Sample Factory's double-buffered
        # sampling, which is how the reference's own learner drives it: while one half's observations are rendered and go through the policy,
        # the other half steps).  Two gyms of n_env / 2 envs, each with its own stream; a half's chain policy -> step -> raster stays in order
        # on its stream, the two chains overlap on the device.  The same number of agent observations per tick as every other leg.
        if n_env % 2 == 0 and n_env >= 2:
            from megaverse_amd.extension import MegaverseGym as _G
            half_frames = frames // 2
            halves = []
            for hidx in range(2):
                st = torch.cuda.Stream(device=device)
                hg = _G(args.scenario, W, H, n_env // 2, A, 8, False, {}, device=local_rank, env_offset=hidx * (n_env // 2), total_envs=n_env)
                hg.set_stream(st.cuda_stream)
                hg.set_pixel_mode(args.pixels)
                slab = torch.zeros((half_frames, H, W, 4), dtype=torch.uint8, device=device)
                hg.set_obs_buffer(slab.data_ptr())
                hg.seed(42 + hidx); hg.reset()
                halves.append((hg, st, slab, slab.view(half_frames, -1)[:, 37:37 + 6 * 97:97], torch.zeros((half_frames, 6), dtype=torch.int32, device=device)))
            torch.cuda.synchronize()

            def half_step(hidx, i):
                hg, st, _, hfeat, hacts = halves[hidx]
                torch.cuda.set_stream(st)
                torch.remainder(hfeat, sizes, out=hacts)
                hg.set_actions_device(hacts.data_ptr())
                hg.step()

            main_stream = torch.cuda.current_stream()
            for i in range(wu):
                half_step(0, step0 + i); half_step(1, step0 + i)
            step0 += wu
            torch.cuda.set_stream(main_stream)
            fence()
            # One host thread enqueues both halves in turn.  The host is NOT the bound: a half-step costs it 16 us (r10a: scripts/probe_host_cost.py, the device
            # kept idle); `host_enqueue_ms_per_step_*` equals the leg's time because the library bounds the host's run-ahead (the status read-back).  The two
            # chains fall into step with each other -- both policies, both step kernels, then both observation passes side by side -- so the pair behaves like
            # one gym (r03c timeline).  Forcing the stagger costs more than it gives: every cross-queue dependency on a chain is 10-40 us on this part (the
            # halves' passes on one stream: 7.3 M, r10e / r10f; passes taking turns through events: r03).  A host thread per half: 9.2 M against 13.5 M (r08v).
            t0 = time.perf_counter()
            for i in range(args.steps):
                half_step(0, step0 + i); half_step(1, step0 + i)
            t_enq = time.perf_counter() - t0
            torch.cuda.set_stream(main_stream)
            fence()
            extra["closed_loop_double_buffered"] = time.perf_counter() - t0
            host_enqueue_ms = t_enq / args.steps * 1e3   # (when this is close to the leg's ms per step, the Python loop is the bound, not the device)
            step0 += args.steps
            checksum_halves = sum(int(h[2][::97].to(torch.int64).sum().item()) for h in halves)
            for h in halves:
                h[0].close()

    # env_step / env_step_batched: what a drop-in user of the reference's Python class calls (megaverse/megaverse_env.py:132-162 <->
    # megaverse_amd/megaverse_env.py).  env_step: MegaverseEnv.step(list of per-agent actions) -> (list of per-agent (3, H, W) numpy frames, rewards, dones,
    # infos) -- the reference's exact shape, which here includes ONE device-to-host copy of the whole RGBA slab per step over PCIe (the reference's
    # getObservation is a view of host memory, megaverse.cpp:139-143: its frames never leave the host) and the per-agent list building; env_step_batched:
    # step_batched(device action tensor) -> (device view of the slab, rewards, dones): observations stay in HBM, two small read-backs per step.  Actions are
    # drawn ahead (the sampler is not the surface).
    if not dry and not mixed and world == 1 and not args.no_extra_legs:
        import numpy as np
        from megaverse_amd.megaverse_env import MegaverseEnv
        env = MegaverseEnv(args.scenario, n_env, A, 0, False, None, img_w=W, img_h=H, device=local_rank)
        env.env.set_pixel_mode(args.pixels)
        env.seed(42)
        env.reset()
        rng = np.random.RandomState(7)
        sizes_np = np.array([3, 3, 3, 2, 2, 3])
        host_acts = [[tuple(int(v) for v in row) for row in (rng.randint(0, 1 << 30, size=(frames, 6)) % sizes_np)] for _ in range(4)]
        dev_acts = [torch.from_numpy((rng.randint(0, 1 << 30, size=(frames, 6)) % sizes_np).astype(np.int32)).to(device) for _ in range(4)]
        n_env_steps = max(10, min(args.steps, 100))
        for i in range(5):
            env.step(host_acts[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_env_steps):
            obs_list, rew, dn, inf = env.step(host_acts[i % 4])
        torch.cuda.synchronize()
        extra_env = {"env_step": (time.perf_counter() - t0, n_env_steps)}
        assert len(obs_list) == frames and obs_list[0].shape == (3, H, W) and len(dn) == frames and len(inf) == frames
        for i in range(5):
            env.step_batched(dev_acts[i % 4])
        torch.cuda.synchronize()
        n_b = max(10, min(args.steps, 400))
        t0 = time.perf_counter()
        for i in range(n_b):
            obs_t, rew, dn = env.step_batched(dev_acts[i % 4])
        torch.cuda.synchronize()
        extra_env["env_step_batched"] = (time.perf_counter() - t0, n_b)
        assert tuple(obs_t.shape) == (frames, 3, H, W) and obs_t.is_cuda
        # env_step_device: step_device(device action tensor) -> three device tensors, no host synchronisation: the Python class with a policy on the device
        for i in range(5):
            env.step_device(dev_acts[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_b):
            obs_t, rew_t, dn_t = env.step_device(dev_acts[i % 4])
        torch.cuda.synchronize()
        extra_env["env_step_device"] = (time.perf_counter() - t0, n_b)
        assert obs_t.is_cuda and rew_t.is_cuda and dn_t.is_cuda and tuple(rew_t.shape) == (frames,) and tuple(dn_t.shape) == (n_env,)
        env.close()
    else:
        extra_env = {}

    checksum = int(slabs[0][::97].to(torch.int64).sum().item())   # touch the result so nothing is optimised away
    if ring is not None:
        checksum += int(ring[:, ::97].to(torch.int64).sum().item())
    checksum += mixed_ring_checksum
    if dry and do_gather and gather_batched:   # every rank's half-ring of the last gathered call must have arrived, in rank order, tick by tick
        g0 = gather.out[last_call["half"]].view(world, batch, frames, H, W, -1)
        for r in range(world):
            for j in range(last_call["ticks"]):
                assert int(g0[r, j, 0, 0, 0, 0]) == (r * 31 + last_call["first"] + j) % 251, "gathered ring half does not hold rank %d's tick %d" % (r, j)
    elif dry and do_gather:   # every rank's shard of the last gathered step must have arrived, in rank order
        g0 = gather.out[last_gathered & 1]
        for r in range(world):
            assert int(g0[r * frames, 0, 0, 0]) == (r * 31 + last_gathered) % 251, "gathered slab does not hold rank %d's shard" % r

    gather_check = None
    if args.check_gather and do_gather and not dry and not mixed and rank == 0:
        # the whole job in one gym: world x envs_per_gpu envs, the same master seed, the same (seed, step) action stream
        from megaverse_amd.extension import MegaverseGym as _G
        big = _G(args.scenario, W, H, world * n_env, A, 8, False, {}, device=local_rank)
        big.set_pixel_mode(args.pixels)
        big.set_sample_policy(args.policy)
        whole = torch.zeros((world * frames, H, W, 4), dtype=torch.uint8, device=device)
        big.set_obs_buffer(whole.data_ptr())
        big.seed(42); big.reset()
        for i in range(last_gathered + 1):
            big.sample_random_actions(1234, i); big.step()
        big.synchronize(); torch.cuda.synchronize()
        if gather_batched:   # the last tick of the last gathered call: slab (ticks - 1) of every rank's half
            got = gather.out[last_call["half"]].view(world, batch, frames, H, W, -1)[:, last_call["ticks"] - 1].reshape(world * frames, H, W, -1).cpu()
        else:
            got = gather.out[last_gathered & 1].cpu()
        gather_check = bool(torch.equal(got, whole.cpu()[..., :got.shape[-1]])) and int(got[..., :3].max()) > 0
        big.close()

    if rank == 0:
        total_obs = world * frames * args.steps
        obst = args.scenario.lower().startswith("obstacles")
        collect = args.scenario.lower() == "collect"
        # algorithmic bytes (DESIGN.md "kernels").  Raster, per frame: the RGBA8 frame written once + the frame's scene (header 128 B,
        # layout boxes, 80 movable boxes 320 B, agents 128 B each).  Obstacles: 128 layout boxes 4096 B + 16 terrain boxes 512 B + 16
        # reward objects 64 B; Collect: ~75 merged slabs on average (3000 generated landscapes) * 32 B + 96 diamonds * 4 B
        scene_bytes = (4096 + 512 + 64) if obst else (75 * 32 + 96 * 4) if collect else 512
        if mixed:   # mean over the eight scenarios: TowerBuilding 512, 2 x Obstacles 4672, Collect 2784, Sokoban ~1300 (slabs + cell map), Rearrange ~600,
            scene_bytes = (512 + 2 * 4672 + 2784 + 1300 + 600 + 2 * 700 * 32) // 8   # 2 x Hex ~700 boxes x 32 B
        bytes_per_frame = W * H * 4 + 128 + scene_bytes + 320 + 128 * A
        # physics kernel (tick + frame setup, one launch), per env: header R+W + scene + movable boxes R+W + per agent (state R+W, action,
        # reward, objective) + done, + per frame the list the raster reads: 800 B header + ~30 visible primitives x 40 B
        step_bytes_per_env = 2 * 128 + scene_bytes + 2 * 320 + A * (2 * 128 + 4 + 4 + 4) + 1 + A * (800 + 30 * 40)
        headline = args.scenario == "TowerBuilding" and (W, H) == (128, 128) and args.policy == "multidiscrete"
        line = {
            "metric": METRIC if headline else METRIC.replace("TowerBuilding 128x128", f"{args.scenario} {W}x{H}").replace("random policy", f"random policy ({args.policy})"),
            "value": total_obs / elapsed, "unit": "agent observations/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scenario} num_envs={n_env} per GPU x {world} GPU(s), num_agents_per_env={A}, obs {W}x{H} RGBA8, "
                                   f"uniform random {args.policy} actions (device, counter-based), natural auto-resets, master seed 42",
                       "envs_per_gpu": n_env, "agents_per_env": A, "obs": [W, H], "gather_obs": bool(do_gather), "pixels": args.pixels, "policy": args.policy,
                       # the step kernels run ahead of the observation passes on a stream of their own (every tick is still
                       # stepped and rendered in full); ticks_per_call > 1: mv_step_n, the streams hand over once per call, tick j of a call
                       # leaves its observations in slab j of a ring of that many slabs
                       "pipelined": pipelined, "ticks_per_call": batch if main_batched else 1,
                       **({"ring_slots": ring_slots, "overlapped_passes": bool(pass_overlap)} if main_batched else {}),
                       "hip_hardware_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       # who draws the episodes (mv_host_generator_threads): the host feeder's threads, or 0 = on the device (TowerBuilding; Collect where the
                       # rank's share of the host is under three cores, or MV_COLLECT_DEVICE_GEN=1)
                       **({"host_generator_threads": gym.host_generator_threads()} if hasattr(gym, "host_generator_threads") else {}),
                       **({"first_calls": os.environ.get("MV_BENCH_CALL_SCHEDULE", "2,4,6") + " ticks, then ticks_per_call (every tick stepped and rendered in full)"} if main_batched else {}),
                       # (a batched group call is two launches where all of the group's envs are resident at once -- up to 1024 -- else two per tick:
                       # mv_api_step.hip, groupBatch)
                       **({"launches_per_call": 2 * ((batch + 7) // 8) if main_batched and n_env <= 1024 else None, "launches_per_tick": None if main_batched and n_env <= 1024 else 2,
                           "scenarios": ", ".join(gym.scenarios) + " dealt round-robin by env index (one gym per scenario, stepped as one mv_group: "
                                        + ("one step launch and one observation launch per batched CALL, every scenario's ticks in its own rollout rings)" if main_batched and n_env <= 1024
                                           else "one step launch and one raster launch per tick)")} if mixed else {}),
                       "parallelism": f"env-shard x{world}"},
        }
        for key, el in extra.items():
            line["value_" + key] = total_obs / el
            line["ms_per_step_" + key] = el / args.steps * 1e3
        for key, (el, n) in extra_env.items():   # (their own step counts: the reference-shaped leg moves 64 MB over PCIe per step)
            line["value_" + key] = frames * n / el
            line["ms_per_step_" + key] = el / n * 1e3
        if extra_env:
            line["env_step_note"] = ("value_env_step = megaverse_amd.MegaverseEnv.step(list of actions): the reference's Python surface, incl. one device-to-host copy of the RGBA slab "
                                     "(%.0f MB over PCIe per step) and the per-agent lists; value_env_step_batched = step_batched(device action tensor): observations stay in HBM, rewards and dones are read back; "
                                     "value_env_step_device = step_device(device action tensor): all three outputs stay on the device, no host synchronisation" % (frames * H * W * 4 / 1e6))
        if "closed_loop" in extra:
            line["host_enqueue_ms_per_step_closed_loop"] = host_enqueue_ms_loop
        if "closed_loop_double_buffered" in extra:
            line["host_enqueue_ms_per_step_closed_loop_double_buffered"] = host_enqueue_ms
        if dry:
            line["dry_run"] = True
        if single:
            line["single_device"] = True
        if gather_check is not None:
            line["gather_check"] = gather_check
        if world > 1:
            line["distributed"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks_per_gpu": 1 if not single else world}
        if elapsed_solo is not None:
            solo = frames * args.steps / elapsed_solo
            line["value_solo_rank0"] = solo   # one GPU of this node alone, same process group (the others idle at a barrier)
            line["value_efficiency"] = (total_obs / elapsed) / world / solo
            if elapsed_no_gather is not None:
                line["value_no_gather_efficiency"] = (total_obs / elapsed_no_gather) / world / solo
        if do_gather:
            slab_bytes = frames * H * W * gather.channels
            line["value_no_gather"] = total_obs / elapsed_no_gather
            # north_star: "near-linear env-shard scaling" -- the envs shard without any exchange, so the claim about the SIMULATOR is read against
            # value_no_gather (every rank's observations stay in its own HBM, where a per-GPU learner consumes them: the reference's own 8-GPU recipe,
            # performance_benchmark_all_envs.py:3-21); `value` adds north_star's "RCCL all-gather to assemble the final observation batch", whose cost is the
            # links' (gather.link_bound_*): at these rates it, not the simulator, bounds `value` from about two GPUs on
            line["scaling_claim_read_against"] = "value_no_gather (simulator, shards only); value = the same steps with every rank receiving the whole batch over xGMI"
            line["ms_per_step_no_gather"] = elapsed_no_gather / args.steps * 1e3
            line["gather"] = {"collective": ("all_gather_into_tensor (RCCL)" if args.gather == "allgather" else "grouped isend / irecv, one shard per peer (RCCL point-to-point)") +
                                            ", double-buffered on a communication stream",
                              "bytes_received_per_gpu_per_step": (world - 1) * slab_bytes,
                              "achieved_GBps_per_gpu": (world - 1) * slab_bytes / (elapsed / args.steps) / 1e9,
                              "xgmi_peak_GBps_per_gpu": XGMI_PEAK_GBS,
                              # what the links allow: every GPU receives (world - 1) shards per step over its 7 point-to-point links
                              "xgmi_bound_ms_per_step": (world - 1) * slab_bytes / (XGMI_PEAK_GBS * 1e9) * 1e3,
                              # the gather-on leg runs the N = 1 headline's batched calls: ONE collective per call of ticks_per_call ticks (the call's half of a
                              # ring of 2 x ticks_per_call slabs) beside the next call's launches; tick by tick (Mixed, --batch 1): one collective per tick
                              "collectives_per_call": 1, "ticks_per_collective": batch if gather_batched else 1,
                              "bytes_received_per_gpu_per_collective": (world - 1) * slab_bytes * (batch if gather_batched else 1),
                              "xgmi_bound_ms_per_collective": (world - 1) * slab_bytes * (batch if gather_batched else 1) / (XGMI_PEAK_GBS * 1e9) * 1e3,
                              "measured_on": "gloo / host copies (single-device test run)" if single else "CPU stand-in (dry run)" if dry else "RCCL over xGMI"}
        if prof is not None:
            traffic = traffic_step = valu = lds = None
            try:   # per-launch PMC figures from the committed rocprofv3 passes (profiles/), only for the profiled config
                pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                pmc_stale = pt.get("source_sha16") != kernel_sources_sha16()   # counters of other sources are not this run's: traffic / valu stay null
                if not pmc_stale and pt["config"] == {"envs_per_gpu": n_env, "agents_per_env": A, "obs": [W, H]} and args.scenario == "TowerBuilding":
                    traffic = pt["kernels"].get("raster", {}).get("traffic_bytes_per_launch")
                    traffic_step = pt["kernels"].get("step", {}).get("traffic_bytes_per_launch")
                    valu = pt["kernels"].get("raster", {}).get("valu")
                    lds = pt["kernels"].get("raster", {}).get("lds")
            except Exception:  # noqa: BLE001
                pass
            raster_ms, step_ms = prof["raster"][0], prof["step"][0]
            achieved = bytes_per_frame * frames / (raster_ms * 1e-3) / 1e9 if raster_ms > 0 else 0.0
            achieved_step = step_bytes_per_env * n_env / (step_ms * 1e-3) / 1e9 if step_ms > 0 else 0.0
            long_list = args.scenario.lower() in ("collect", "hexmemory", "hexexplore")
            # the k ticks of a call as ONE step launch (step_ticks_kernel / step_<scenario>_ticks_kernel: one agent per env, TowerBuilding also with several)
            # and its k observation passes as ONE launch (raster_fast_batch_kernel, raster_glist_batch_kernel for the long lists): mv_api.hip, canMultiTick /
            # canBatchRaster
            batch_step = batched and not mixed and (A == 1 or args.scenario == "TowerBuilding") and os.environ.get("MV_STEP_TICKS", "1") != "0"
            fam = {"towerbuilding": "", "collect": "collect_", "rearrange": "rearrange_", "sokoban": "sokoban_", "hexmemory": "hex_", "hexexplore": "hex_"}.get(args.scenario.lower(), "obstacles_")
            ticks_kernel_name = "step_ticks_agents_kernel" if A > 1 else "step_%sticks_kernel" % fam
            batch_raster = batch_step and args.pixels == "fast" and os.environ.get("MV_RASTER_BATCH", "8") != "0"
            line["roofline"] = {"bound": "valu", "kernel": (("mv::raster_union_batch_kernel (the k x n observation passes of a group call in one launch: short-list and long-list bodies; per tick)" if batched and n_env <= 1024 else "mv::raster_union_all_kernel (one launch per tick for all scenarios: short-list and long-list bodies)") if mixed else
                                                          "mv::%s (the %d observation passes of a call in one launch; every figure here is PER TICK)" % ("raster_glist_batch_kernel" if long_list else "raster_fast_batch_kernel", batch) if batch_raster
                                                          else "mv::raster_glist_kernel" if long_list else "mv::raster_fast_kernel") if args.pixels == "fast" else "mv::raster_kernel",
                                "ticks_per_launch": batch if batch_raster else 1,
                                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                                "avg_launch_ms": raster_ms, "launches_timed": prof["raster"][1], "algorithmic_bytes_per_launch": bytes_per_frame * frames,
                                "note": "dominant kernel; achieved / peak / frac are the HBM figures (algorithmic bytes per tick / launch time per tick against 8 TB/s); "
                                        "the kernel is bound by VALU issue, see `valu` (its HBM traffic is the pixels: with the non-temporal pixel stores of round 4 the write counter reads "
                                        "~1.3x the algorithmic bytes -- half-line writes no longer merge in L2 -- against 1.0x with plain stores, which were 2.5 % slower); launch time from HIP events "
                                        "(same stream, around the launch the product runs: a batched call's one launch / its ticks) in a separate untimed loop; "
                                        "traffic = HBM bytes per tick from rocprofv3 PMC passes of the same kernel sources (null otherwise)"}
            if valu and raster_ms > 0:   # wave64 VALU instructions issue over 2 cycles on a SIMD-32; 256 CUs x 4 SIMDs x 2.4 GHz
                insts = valu["valu_insts_per_launch"]
                line["roofline"]["valu"] = {"insts_per_launch": insts, "salu_insts_per_launch": valu.get("salu_insts_per_launch"),
                                            "insts_per_64px_tile": insts / (frames * W * H / 64.0),
                                            "issue_peak_insts_per_s": VALU_ISSUE_PEAK, "frac_of_issue_peak": insts / (raster_ms * 1e-3) / VALU_ISSUE_PEAK,
                                            # scripts/probe_valu.hip on this part (profiles/README.md): a SIMD with 8 waves of back-to-back v_fma / v_mul
                                            # retires one per 1.11 ns (the clock under an all-VALU load is below 2.4 GHz); min / max / cndmask / max3 cost more
                                            "probed_ns_per_fma_per_simd": 1.11, "frac_of_probed_fma_rate": insts * 1.11e-9 / 1024.0 / (raster_ms * 1e-3),
                                            # counters: quad-cycles the vector ALUs were busy, over the chip's 1024 SIMDs x the launch's clocks at 2.4 GHz
                                            # (rocprofv3's derived VALUBusy of the same passes: 91 % for this kernel alone on the chip, profiles/r06s_*)
                                            "valu_busy_frac_at_2.4GHz": (valu["active_inst_valu_quadcycles"] * 4.0 / 1024.0 / (raster_ms * 1e-3 * 2.4e9)) if valu.get("active_inst_valu_quadcycles") else None,
                                            "source": valu.get("source")}
            # north_star: "LDS hit rate on the raster tile" -- an LDS access has no miss, only bank-conflict replays: the counters of the committed SQ passes,
            # per tick
            if lds:
                line["roofline"]["lds"] = dict(lds, note="LDS instructions per tick (wave-level), quad-cycles the LDS pipe was busy / replaying bank conflicts; hit rate = 1 - conflict_frac")
            line["roofline_physics"] = {"bound": "latency", "kernel": (("mv::step_union_ticks_kernel (the k ticks of all scenarios in one launch; per tick)" if batched and n_env <= 1024 else "mv::step_union_kernel") if mixed else "mv::%s (the %d ticks of a call in %d launch%s; per tick)" % (ticks_kernel_name, batch, (batch + 7) // 8, "es of 8" if batch > 8 else "") if batch_step else "mv::step_kernel") +
                                                              " (voxel physics + scenario logic + auto-reset + frame setup)", "ticks_per_launch": min(batch, 8) if batch_step else 1, "achieved": achieved_step,
                                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_step / HBM_PEAK_GBS, "traffic": traffic_step,
                                        "avg_launch_ms": step_ms, "launches_timed": prof["step"][1], "algorithmic_bytes_per_launch": step_bytes_per_env * n_env,
                                        # the same launch priced with SURVEY 8(d)'s per-env figure (17.9 KB for one agent, 18.7 KB for four: the 16 KB voxel
                                        # chunk counted as read every tick -- which this kernel does NOT do: it works from the box lists, DESIGN.md header)
                                        "survey_bytes_per_env": 17900 + (A - 1) * 267,
                                        "frac_survey_bytes": ((17900 + (A - 1) * 267) * n_env / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if step_ms else None,
                                        "note": "north_star's >=40 % HBM target names this kernel and is NOT met in either convention (`frac`: the bytes the kernel moves; "
                                                "`frac_survey_bytes`: SURVEY 8(d)'s 17.9 KB per env, chunk included): its working set is 1.7 KB per env (the 16 KB voxel chunk "
                                                "is not streamed), so it is latency-bound, not bandwidth-bound: one wave per env, launch length = slowest wave (DESIGN.md 3.1); "
                                                "when pipelined it runs concurrently with the previous ticks' rasters, which stretches its launches.  40 % of 8 TB/s would mean "
                                                "a 1024-env tick in 1.2 us of streaming for 3.8 MB -- the tick's dependent chain of casts is ~10 us whatever the bandwidth"}
            if prof_alone and prof_alone["step"][0] > 0 and batch_step:
                sa = prof_alone["step"][0]
                line["roofline_physics"]["alone"] = {"avg_launch_ms": sa, "frac": step_bytes_per_env * n_env / (sa * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                     "frac_survey_bytes": (17900 + (A - 1) * 267) * n_env / (sa * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                     "note": "the same multi-tick launch with nothing beside it (pipelining off for a few calls after the timed region), per tick, "
                                                             "between HIP events (a kernel trace reads ~4 us per tick less: r12zb, 114 us per 8 ticks); the kernel flavour the "
                                                             "library's rule picks for this gym -- one wave per env at 1024 envs; the two-wave software-pipelined kernel, "
                                                             "MV_STEP_PIPE=1, is a quarter shorter alone"}
            if args.pixels == "exact":
                line["kernels"] = {"publish_and_frame_sort": {"avg_launch_ms": prof["setup"][0], "note": "exact pixel mode only; same-stream interval"}}
        line["checksum"] = checksum
        if world == 1 and not args.no_cpu_baseline and not dry:
            if mixed:
                # (the same scenario set the GPU gym ran: Mixed4's four, not the eight)
                line["cpu_baseline"] = cpu_baseline_mixed(list(gym.scenarios), W, H, n_env, A)
            else:
                line["cpu_baseline"] = cpu_baseline(args.scenario, W, H, n_env, A, args.policy, full=args.cpu_baseline_full)
        print(json.dumps(line), flush=True)

    gym.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
