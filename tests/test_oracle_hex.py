"""CPU tests of the Hex scenarios' building blocks in oracle/ (SURVEY.md 8f-4).  The honeycomb maze generator (src/libs/mazes in the
reference tree: vendored, dependency-free) is PINNED: the restatement is compared with the library itself, compiled in place into
oracle/_ref/libmv_ref_mazes.so with a seeded subclass of its Kruskal (the reference seeds it from std::random_device)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib

REF = os.path.join(oracle_lib.ORACLE_DIR, "_ref", "libmv_ref_mazes.so")


def maze(fn, size, seed):
    cells = C.c_int(0)
    n = fn(size, seed, C.byref(cells), None, None, None, None, None)
    counts = np.zeros(cells.value, np.int32); to = np.zeros(n, np.int32); xy = np.zeros((n, 4), np.float64)
    centers = np.zeros((cells.value, 2), np.float64); bounds = np.zeros(4, np.float64)
    assert fn(size, seed, C.byref(cells), counts.ctypes.data, to.ctypes.data, xy.ctypes.data, centers.ctypes.data, bounds.ctypes.data) == n
    return cells.value, counts, to, xy, centers, bounds


def bind(lib, name):
    fn = getattr(lib, name)
    fn.argtypes = [C.c_int, C.c_uint, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    fn.restype = C.c_int
    return fn


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libmv_ref_mazes.so is built from /root/reference (not present here)")
@pytest.mark.parametrize("size", [2, 3, 5, 8, 10])
def test_honeycomb_maze_matches_the_reference_library(size):
    ref, mine = bind(C.CDLL(REF), "mvref_hex_maze"), bind(oracle_lib.lib(), "mvo_hex_maze")
    for seed in (0, 1, 12345, 2 ** 31 + 7):
        a, b = maze(ref, size, seed), maze(mine, size, seed)
        assert a[0] == b[0] == 3 * size * (size - 1) + 1
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])     # borders left per cell and their neighbours, list order included
        assert np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])     # cell centres and bounds, bit for bit
        # border end points: the restatement uses a literal table for the seven angles' cos / sin (a compiler is free to call sincos()
        # for cos(): one ulp of a double in a cancelling sum), the library whatever its build calls
        assert np.abs(a[3] - b[3]).max() < 1e-15


@pytest.mark.parametrize("size", [2, 4, 7])
def test_honeycomb_maze_is_a_spanning_tree(size):
    cells, counts, to, xy, centers, bounds = maze(bind(oracle_lib.lib(), "mvo_hex_maze"), size, 99)
    # borders left between cells = inner edges - (cells - 1) removed by the spanning tree, each listed on both sides
    inner = int((to >= 0).sum()) // 2
    total_inner = (6 * cells - 6 * (2 * size - 1)) // 2       # every cell has 6 neighbours except along the rim
    assert inner == total_inner - (cells - 1)
    # passages (removed borders) connect everything: union-find over pairs of adjacent cells that do NOT share a remaining border
    start = np.concatenate([[0], np.cumsum(counts)])
    walls = {(i, int(t)) for i in range(cells) for t in to[start[i]:start[i + 1]] if t >= 0}
    parent = list(range(cells))
    def root(u):
        while parent[u] != u:
            parent[u] = parent[parent[u]]; u = parent[u]
        return u
    d = np.linalg.norm(centers[:, None, :] - centers[None, :, :], axis=-1)
    for i in range(cells):
        for j in range(i + 1, cells):
            if abs(d[i, j] - np.sqrt(3)) < 1e-9 and (i, j) not in walls:
                parent[root(i)] = root(j)
    assert len({root(i) for i in range(cells)}) == 1
    assert np.allclose(bounds, [-np.sqrt(3) * (size - 0.5), -(1.5 * size - 0.5), np.sqrt(3) * (size - 0.5), 1.5 * size - 0.5])
