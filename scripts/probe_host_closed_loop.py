"""host cost of the pieces of a closed-loop tick (policy kernel, mv_set_actions_device, mv_step): enqueue time per call, device idle-limited
(small gym, so that the device is never the bound)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaverse_amd.extension import MegaverseGym

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = "cuda:0"
st = torch.cuda.Stream(device=dev)
g = MegaverseGym("TowerBuilding", 128, 128, n, 1, 8, False, {})
g.set_stream(st.cuda_stream)
g.set_pixel_mode("fast")
slab = torch.zeros((n, 128, 128, 4), dtype=torch.uint8, device=dev)
g.set_obs_buffer(slab.data_ptr())
g.seed(42); g.reset()
sizes = torch.tensor([3, 3, 3, 2, 2, 3], dtype=torch.int32, device=dev)
acts = torch.zeros((n, 6), dtype=torch.int32, device=dev)
feat = slab.view(n, -1)[:, 37:37 + 6 * 97:97]
main = torch.cuda.current_stream()
K = int(os.environ.get("K", 300))


def t(name, f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        f(i)
    el = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("%-34s %.2f us per call (enqueue)" % (name, el / K * 1e6))


t("torch.cuda.set_stream", lambda i: torch.cuda.set_stream(st))
torch.cuda.set_stream(st)
t("torch.remainder(out=)", lambda i: torch.remainder(feat, sizes, out=acts))
t("set_actions_device", lambda i: g.set_actions_device(acts.data_ptr()))
t("acts.data_ptr()", lambda i: acts.data_ptr())
t("step (closed loop: actions set)", lambda i: (g.set_actions_device(acts.data_ptr()), g.step()))
t("sample_random_actions + step", lambda i: (g.sample_random_actions(1, i), g.step()))
t("step_no_render", lambda i: (g.set_actions_device(acts.data_ptr()), g.step_no_render()))
t("render", lambda i: g.render())
torch.cuda.set_stream(main)
g.close()
