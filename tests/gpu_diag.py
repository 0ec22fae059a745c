import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hip_util import *
t00 = time.time()
def P(*a): print(f"[{time.time()-t00:7.2f}]", *a, flush=True)
N, A = int(sys.argv[1]), int(sys.argv[2])
og, hg = make_pair(N, A, 128, 128, seed=42)
P("pair made")
for st in range(int(sys.argv[3])):
    acts = set_same_actions(og, hg, N, A, 1234, st)
    if st < 3 or st % 100 == 0: P("step", st, "acts", acts.tolist())
    og.step_norender()
    hg.step_no_render()
    hg.synchronize()
    if st < 3 or st % 100 == 0:
        P("  synced")
        for e in range(N):
            dd = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
            if dd: P("  diff env", e, dd[:6])
P("done")
