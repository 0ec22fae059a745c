"""where the step kernel's time goes: the tick alone (mv_step_no_render) and the frame setup alone (mv_render's stand-alone kernel), to be run under
rocprofv3 --kernel-trace --stats:  python scripts/probe_tick.py [scenario] [envs] [agents]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from megaverse_amd.extension import MegaverseGym  # noqa: E402

scenario = sys.argv[1] if len(sys.argv) > 1 else "TowerBuilding"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
a = int(sys.argv[3]) if len(sys.argv) > 3 else 1
os.environ.setdefault("BOXOBAN_LEVELS", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "boxoban"))
g = MegaverseGym(scenario, 128, 128, n, a, 8, False, {})
g.set_pixel_mode("fast")
g.set_pipelining(False)
mode = os.environ.get("PROBE_MODE", "both")   # tick | fused | both
g.seed(42)
g.reset()
for st in range(int(os.environ.get("PROBE_WARM", "0"))):   # (with MV_TICK_TIMING_SKIP set to the same number: statistics of the steady state)
    g.sample_random_actions(1234, 100000 + st)
    g.step_no_render()
for st in range(300 if mode != "fused" else 0):
    g.sample_random_actions(1234, st)
    g.step_no_render()
g.synchronize()
for st in range(100 if mode == "both" else 0):
    g.render()
g.synchronize()
for st in range(300, 500 if mode != "tick" else 300):
    g.sample_random_actions(1234, st)
    g.step()
g.synchronize()
g.close()
