import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from megaverse_amd.extension import MegaverseGym
from megaverse_amd.rollout import sample_actions
A = int(sys.argv[1]); seed = int(sys.argv[2]); N = 12
hg = MegaverseGym("Rearrange", 32, 32, N, A, 1, False, {}); hg.seed(seed); hg.reset()
for st in range(1000):
    hg.set_actions_batched(sample_actions(300 + seed, st, N * A))
    hg.step_no_render()
    hg.synchronize()
    print(st, flush=True)
print("done")
