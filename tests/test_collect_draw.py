"""The Collect generator that runs on the device (megaverse_amd/csrc/mv_collect_draw.h), compiled for the CPU (mv_debug_collect_draw_host: no GPU involved),
against the product's host generator (mv_gen_collect.cpp through mv_debug_generate_episode, which test_host_generators.py holds against the oracle and
test_oracle_collect.py against the reference's perlin_noise.hpp): the same env seed -> byte-identical episodes -- slabs, height map, spawn cells and rotations,
movable boxes, diamonds and their values, colours, episode length -- over consecutive episodes of one env's stream.  The host generator draws through
libstdc++'s mt19937 / minstd_rand0 / uniform_int_distribution / shuffle / sort; the device code restates those algorithms, so this is where the restatement is
pinned (the unstable std::sort's order of equal keys decides which cells the diamonds take)."""
import numpy as np
import pytest

from megaverse_amd import extension as ext
from test_host_generators import COLLECT_BLOB, generate


def draw_host(agents, env_seed, n, base_len=60.0):
    lib = ext.load_library()
    size = lib.mv_debug_collect_draw_host(agents, env_seed, n, base_len, None, 0)
    assert size == COLLECT_BLOB.itemsize
    buf = np.zeros(1, COLLECT_BLOB)
    assert lib.mv_debug_collect_draw_host(agents, env_seed, n, base_len, buf.ctypes.data, size) == size
    return buf[0]


def assert_same_episode(got, want, agents, what):
    for f in ("num_boxes", "num_objects", "num_rewards", "num_positive", "layout_color", "wall_color", "episode_len", "pad"):
        assert got[f] == want[f], (what, f, got[f], want[f])
    assert (got["dim"] == want["dim"]).all(), (what, got["dim"], want["dim"])
    assert (got["spawn"] == want["spawn"]).all(), what
    assert (got["yaw_frand"][:agents].view(np.uint32) == want["yaw_frand"][:agents].view(np.uint32)).all(), what
    assert (got["heightmap"] == want["heightmap"]).all(), what
    for name, count in (("objects", "num_objects"), ("rewards", "num_rewards"), ("boxes", "num_boxes")):
        n = int(want[count])
        assert got[name][:n].tobytes() == want[name][:n].tobytes(), (what, name)


@pytest.mark.parametrize("agents", [1, 2, 8])
def test_device_code_on_the_host_draws_the_host_generators_episodes(agents):
    rng = np.random.default_rng(1234 + agents)
    seeds = [0, 1, 42, (1 << 30) - 1] + [int(s) for s in rng.integers(0, 1 << 30, 60 if agents == 1 else 24)]
    for s in seeds:
        for n in (1, 2, 3):
            got, want = draw_host(agents, s, n), generate("Collect", agents, s, n)
            assert got["seq"] == n
            assert_same_episode(got, want, agents, (s, n))


def test_long_env_stream_and_episode_length_parameter():
    # one env's stream far down (every episode re-seeds from the one before), and the base episode length
    for n in (10, 57):
        assert_same_episode(draw_host(3, 777, n), generate("Collect", 3, 777, n), 3, n)
    got, want = draw_host(1, 5, 1, 10.0), generate("Collect", 1, 5, 1, 10.0)
    assert got["episode_len"] == want["episode_len"] and got["episode_len"] < 60.0
