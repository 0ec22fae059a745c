"""First-light script for the GPU box (not a pytest file): prints parity + timing facts."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ctypes as C
import oracle_lib
from hip_util import *
import megaverse_amd.extension as ext

lib = ext.load_library()
rng = np.random.default_rng(0)

# 1. fp32 exactness of / sqrt, no fma contraction
n = 1 << 16
a = (rng.standard_normal(n) * 10 ** rng.uniform(-6, 6, n)).astype(np.float32); b = (rng.standard_normal(n) * 10 ** rng.uniform(-6, 6, n)).astype(np.float32)
out = np.empty(n, np.float32)
assert lib.mv_debug_math(0, 0, a.ctypes.data, b.ctypes.data, n, out.ctypes.data) == 0, lib.mv_last_error()
print("div exact:", np.array_equal(out.view(np.uint32), (a / b).view(np.uint32)))
lib.mv_debug_math(0, 1, np.abs(a).ctypes.data, None, n, out.ctypes.data)
print("sqrt exact:", np.array_equal(out.view(np.uint32), np.sqrt(np.abs(a)).view(np.uint32)))
lib.mv_debug_math(0, 3, a.ctypes.data, b.ctypes.data, n, out.ctypes.data)
print("no-fma exact:", np.array_equal(out.view(np.uint32), ((a * b).astype(np.float32) + a).view(np.uint32)))
x = rng.uniform(-7, 7, n).astype(np.float32); out2 = np.empty(2 * n, np.float32)
lib.mv_debug_math(0, 2, x.ctypes.data, None, n, out2.ctypes.data)
L = oracle_lib.lib(); s = C.c_float(); c = C.c_float(); ok = True
for i in range(0, n, 37):
    L.mvo_sincos(float(x[i]), C.byref(s), C.byref(c))
    ok &= (np.float32(s.value).view(np.uint32) == out2[2*i].view(np.uint32)) and (np.float32(c.value).view(np.uint32) == out2[2*i+1].view(np.uint32))
print("sincos exact:", bool(ok), "max err vs libm", np.abs(out2[0::2] - np.sin(x.astype(np.float64))).max())

# 2. rng streams
m = 2000
o = np.empty(m, np.uint32); lib.mv_debug_rng(0, 5489, 0, None, None, m, o.ctypes.data)
print("mt19937 first:", o[:3], "oracle:", [L.mvo_mt19937_nth(5489, k) for k in (1, 2, 3)], "1999th eq:", o[1998] == L.mvo_mt19937_nth(5489, 1999))
lo = rng.integers(-5, 5, m).astype(np.int32); hi = (lo + rng.integers(1, 1000, m)).astype(np.int32)
d = np.empty(m, np.int32); r = np.empty(m, np.int32)
lib.mv_debug_rng(0, 42, 1, lo.ctypes.data, hi.ctypes.data, m, d.ctypes.data); L.mvo_rand_range_seq(42, lo.ctypes.data, hi.ctypes.data, m, r.ctypes.data)
print("rand_range eq:", np.array_equal(d, r))
df = np.empty(m, np.float32); rf = np.empty(m, np.float32)
lib.mv_debug_rng(0, 7, 2, None, None, m, df.ctypes.data); L.mvo_frand_seq(7, m, rf.ctypes.data)
print("frand eq:", np.array_equal(df.view(np.uint32), rf.view(np.uint32)))
for nn in (1, 2, 7, 594, 600):
    ds = np.empty(nn, np.int32); rs = np.empty(nn, np.int32)
    lib.mv_debug_rng(0, 99, 3, None, None, nn, ds.ctypes.data); L.mvo_shuffle_iota(99, nn, rs.ctypes.data)
    print("shuffle", nn, np.array_equal(ds, rs))

# 3. reset parity
for A in (1, 4):
    N = 16
    og, hg = make_pair(N, A, 128, 128, seed=42)
    bad = 0
    for e in range(N):
        dd = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
        if dd:
            bad += 1
            if bad <= 2: print("reset diff env", e, dd[:6])
    print(f"reset parity A={A}: {N-bad}/{N} envs equal")
    # pixels after reset
    og_obs = np.stack([og.get_observation(e, a) for e in range(2) for a in range(A)])
    hg_obs = np.stack([hg.get_observation(e, a) for e in range(2) for a in range(A)])
    print("  reset pixels equal:", np.array_equal(og_obs, hg_obs), "ndiff", int((og_obs != hg_obs).sum()))
    # 4. rollout parity
    steps = 1500 if A == 1 else 600
    first_bad = None
    t0 = time.time()
    for st in range(steps):
        set_same_actions(og, hg, N, A, 1234, st)
        og.step_norender(); hg.step_no_render()
        if st % 50 == 49 or st < 5:
            for e in range(N):
                dd = diff_snapshots(og.snapshot(e), hip_snapshot(hg, e), A)
                if dd and first_bad is None:
                    first_bad = (st, e, dd[:8])
            ro = og.get_last_rewards(); rh = hg.get_rewards_array()
            if not np.array_equal(ro.view(np.uint32), rh.view(np.uint32)) and first_bad is None: first_bad = (st, 'rewards', ro, rh)
            if first_bad: break
    print(f"rollout parity A={A} steps={steps}: first mismatch = {first_bad}  ({time.time()-t0:.1f}s)")
    og.render(); hg.render()
    og_obs = np.stack([og.get_observation(e, a) for e in range(N) for a in range(A)])
    hg_obs = np.stack([hg.get_observation(e, a) for e in range(N) for a in range(A)])
    print("  post-rollout pixels equal:", np.array_equal(og_obs, hg_obs), "ndiff", int((og_obs != hg_obs).sum()), "of", og_obs.size)
    if not np.array_equal(og_obs, hg_obs):
        f = np.nonzero((og_obs != hg_obs).reshape(N*A, -1).any(1))[0]; print("  frames differing:", f[:10])
    og.close(); hg.close()

# 5. timing
import torch
for N in (1024,):
    hg = MegaverseGym("TowerBuilding", 128, 128, N, 1, 1, False, {})
    hg.seed(42); hg.reset(); hg.synchronize()
    for st in range(20):
        hg.sample_random_actions(1234, st); hg.step()
    hg.synchronize(); t0 = time.time(); K = 200
    for st in range(K):
        hg.sample_random_actions(1234, 20 + st); hg.step()
    hg.synchronize(); dt = time.time() - t0
    print(f"N={N}: {dt/K*1e3:.3f} ms/step  {N*K/dt:,.0f} obs/s")
    t0 = time.time()
    for st in range(K):
        hg.sample_random_actions(1234, 220 + st); hg.step_no_render()
    hg.synchronize(); dt = time.time() - t0
    print(f"   no-render: {dt/K*1e3:.3f} ms/step")
    hg.close()
