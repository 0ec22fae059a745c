"""Host cost of one half-step of the policy-in-the-loop legs (bench.py: closed_loop_double_buffered), piece by piece: what one Python thread pays to
enqueue `policy -> set_actions_device -> step` for a gym of `envs` envs, with the device kept idle enough that nothing is back-pressure.
usage: python scripts/probe_host_cost.py [envs] [iters]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from megaverse_amd.extension import MegaverseGym  # noqa: E402

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
W = H = 128
dev = torch.device("cuda:0")
st = torch.cuda.Stream(device=dev)
g = MegaverseGym("TowerBuilding", W, H, envs, 1, 8, False, {}, device=0)
g.set_stream(st.cuda_stream)
g.set_pixel_mode("fast")
slab = torch.zeros((envs, H, W, 4), dtype=torch.uint8, device=dev)
g.set_obs_buffer(slab.data_ptr())
g.seed(42); g.reset()
sizes = torch.tensor([3, 3, 3, 2, 2, 3], dtype=torch.int32, device=dev)
acts = torch.zeros((envs, 6), dtype=torch.int32, device=dev)
feat = slab.view(envs, -1)[:, 37:37 + 6 * 97:97]
torch.cuda.synchronize()


def timed(name, fn, chunk=20):
    """host seconds per call, the device drained every `chunk` calls outside the clock"""
    total = 0.0
    n = 0
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    while n < iters:
        t0 = time.perf_counter()
        for _ in range(chunk):
            fn()
        total += time.perf_counter() - t0
        n += chunk
        torch.cuda.synchronize()
    print(f"{name:40s} {total / n * 1e6:8.2f} us per call", flush=True)
    return total / n


main = torch.cuda.current_stream()
timed("torch.cuda.set_stream", lambda: torch.cuda.set_stream(st))
torch.cuda.set_stream(st)
timed("torch.remainder(feat, sizes, out=acts)", lambda: torch.remainder(feat, sizes, out=acts))
ptr = acts.data_ptr()
timed("acts.data_ptr()", lambda: acts.data_ptr())
timed("set_actions_device", lambda: g.set_actions_device(ptr))


def step_only():
    g.set_actions_device(ptr)
    g.step()


t_step = timed("set_actions_device + step", step_only)


def half_step():
    torch.cuda.set_stream(st)
    torch.remainder(feat, sizes, out=acts)
    g.set_actions_device(acts.data_ptr())
    g.step()


timed("whole half-step", half_step)
timed("whole half-step, drained every 200", half_step, chunk=200)
if hasattr(g, "step_after_policy"):
    timed("step_after_policy(ptr)", lambda: g.step_after_policy(ptr))
# the open-loop single step for comparison (actions drawn in the kernel: the pipelined path, two streams)
timed("step_n(1, sampled)", lambda: g.step_n(1, "multidiscrete", 7, 0))
torch.cuda.set_stream(main)
g.close()
