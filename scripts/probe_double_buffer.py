"""closed loop in two halves (bench.py's closed_loop_double_buffered leg on its own, for a kernel trace):
   python scripts/probe_double_buffer.py [envs] [ticks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaverse_amd.extension import MegaverseGym

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = "cuda:0"
sizes = torch.tensor([3, 3, 3, 2, 2, 3], dtype=torch.int32, device=dev)
halves = []
for h in range(2):
    st = torch.cuda.Stream(device=dev, priority=(-1 if h == 0 else 0) if os.environ.get("PRIO", "0") == "1" else 0)
    g = MegaverseGym("TowerBuilding", 128, 128, n // 2, 1, 8, False, {}, env_offset=h * (n // 2), total_envs=n)
    g.set_stream(st.cuda_stream)
    g.set_pixel_mode("fast")
    slab = torch.zeros((n // 2, 128, 128, 4), dtype=torch.uint8, device=dev)
    g.set_obs_buffer(slab.data_ptr())
    g.seed(42 + h); g.reset()
    halves.append((g, st, slab.view(n // 2, -1)[:, 37:37 + 6 * 97:97], torch.zeros((n // 2, 6), dtype=torch.int32, device=dev), slab))
torch.cuda.synchronize()
main = torch.cuda.current_stream()


main = torch.cuda.current_stream()
TOKEN = os.environ.get("TOKEN", "0") == "1"   # the two halves' observation passes take turns (events), so that one half steps while the other renders
done = [torch.cuda.Event(), torch.cuda.Event()]
done[1].record(main)


def half(h):
    g, st, feat, acts, _ = halves[h]
    torch.cuda.set_stream(st)
    torch.remainder(feat, sizes, out=acts)
    g.set_actions_device(acts.data_ptr())
    if TOKEN:
        g.step_no_render()
        st.wait_event(done[1 - h])
        g.render()
        done[h].record(st)
    else:
        g.step()


for i in range(30):
    half(0); half(1)
torch.cuda.set_stream(main); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(ticks):
    half(0); half(1)
t1 = time.perf_counter()
torch.cuda.set_stream(main); torch.cuda.synchronize()
t2 = time.perf_counter()
print("double-buffered closed loop: %.2f us per tick of %d envs (%.2f M obs/s); host enqueue %.2f us per tick" % ((t2 - t0) / ticks * 1e6, n, n * ticks / (t2 - t0) / 1e6, (t1 - t0) / ticks * 1e6))
for h in halves:
    h[0].close()
