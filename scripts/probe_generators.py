"""Episodes per second ONE host thread generates, per host-generated scenario (the feeder's bound).  CPU only: the generators are
reached through the C ABI's test hook mv_debug_generate_episode(scenario, agents, env_seed, n, ...), which generates the first n episodes of an
env's stream; the time of n = N against n = 1 is N - 1 episodes.

    python scripts/probe_generators.py [N]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megaverse_amd import extension as ext  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    lib = ext.load_library()
    print("%-16s %12s %14s" % ("scenario", "us/episode", "episodes/s"))
    for scenario in ("ObstaclesEasy", "ObstaclesMedium", "ObstaclesHard", "Empty", "Collect", "Rearrange", "HexMemory", "HexExplore"):
        size = lib.mv_debug_generate_episode(scenario.encode(), 1, 12345, 1, 60.0, None, 0)
        if size <= 0:
            print("%-16s not available" % scenario)
            continue
        buf = (ext.C.c_uint8 * size)() if hasattr(ext, "C") else __import__("ctypes").create_string_buffer(size)
        best = None
        for _ in range(3):
            t0 = time.perf_counter(); lib.mv_debug_generate_episode(scenario.encode(), 1, 12345, 1, 60.0, buf, size)
            t1 = time.perf_counter(); lib.mv_debug_generate_episode(scenario.encode(), 1, 12345, n, 60.0, buf, size)
            t2 = time.perf_counter()
            per = ((t2 - t1) - (t1 - t0)) / (n - 1)
            best = per if best is None else min(best, per)
        print("%-16s %12.1f %14.0f" % (scenario, best * 1e6, 1.0 / best))


if __name__ == "__main__":
    main()
