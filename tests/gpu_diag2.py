import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hip_util import *
t00 = time.time()
def P(*a): print(f"[{time.time()-t00:7.2f}]", *a, flush=True)
N, A = 16, 1
og, hg = make_pair(N, A, 128, 128, seed=42)
for st in range(60):
    acts = set_same_actions(og, hg, N, A, 1234, st)
    og.step_norender()
    if st >= 24: P("launch step", st)
    hg.step_no_render()
    hg.synchronize()
    if st >= 24:
        P("  synced", st)
        for e in range(N):
            dd = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
            if dd: P("  diff env", e, dd[:6])
P("done")
