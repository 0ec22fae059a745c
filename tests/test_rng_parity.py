"""GPU: the device restatement of the RNG helpers (megaverse_amd/csrc/mv_rng.h) against the oracle's
libstdc++ calls AND the reference's util.hpp compiled in place (oracle/_ref), bit for bit."""
import numpy as np
import pytest

import oracle_lib

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def test_device_mt19937_known_answer(hip):
    n = 10000
    out = np.empty(n, np.uint32)
    assert hip.load_library().mv_debug_rng(0, 5489, 0, None, None, n, out.ctypes.data) == 0
    assert out[-1] == 4123659995            # C++ standard [rand.predef]
    assert out[0] == oracle_lib.lib().mvo_mt19937_nth(5489, 1)


@pytest.mark.parametrize("seed", [0, 42, 4294967295])
def test_device_rand_range_and_frand(hip, seed):
    lib, L = hip.load_library(), oracle_lib.lib()
    rng = np.random.default_rng(seed)
    n = 3000
    lo = rng.integers(-50, 50, n).astype(np.int32)
    hi = (lo + rng.integers(1, 2000, n)).astype(np.int32)
    lo[:20], hi[:20] = 0, 1 << 30
    d, o = np.empty(n, np.int32), np.empty(n, np.int32)
    assert lib.mv_debug_rng(0, seed, 1, lo.ctypes.data, hi.ctypes.data, n, d.ctypes.data) == 0
    L.mvo_rand_range_seq(seed, lo.ctypes.data, hi.ctypes.data, n, o.ctypes.data)
    assert np.array_equal(d, o)
    ref = oracle_lib.ref_lib()
    if ref is not None:
        r = np.empty(n, np.int32)
        ref.mvref_rand_range_seq(seed, lo.ctypes.data, hi.ctypes.data, n, r.ctypes.data)
        assert np.array_equal(d, r)
    df, of = np.empty(n, np.float32), np.empty(n, np.float32)
    assert lib.mv_debug_rng(0, seed, 2, None, None, n, df.ctypes.data) == 0
    L.mvo_frand_seq(seed, n, of.ctypes.data)
    assert np.array_equal(df.view(np.uint32), of.view(np.uint32))
    if ref is not None:
        rf = np.empty(n, np.float32)
        ref.mvref_frand_seq(seed, n, rf.ctypes.data)
        assert np.array_equal(df.view(np.uint32), rf.view(np.uint32))


@pytest.mark.parametrize("n", [1, 2, 3, 10, 99, 100, 594, 1023, 4096])
def test_device_shuffle_matches_std_shuffle(hip, n):
    d, o = np.empty(n, np.int32), np.empty(n, np.int32)
    assert hip.load_library().mv_debug_rng(0, 99, 3, None, None, n, d.ctypes.data) == 0
    oracle_lib.lib().mvo_shuffle_iota(99, n, o.ctypes.data)
    assert np.array_equal(d, o)
    assert sorted(d.tolist()) == list(range(n))
