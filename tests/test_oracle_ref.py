"""Pins the oracle's open list, neighbour order and RNG to the REFERENCE'S OWN CODE:
oracle/_ref/libppref.so is built (oracle/Makefile `ref`) from /root/reference's
utils/frontier.h, utils/grid.cpp, utils/random.h, utils/maths.h -- the only
reference files that compile in this image without Eigen/flann.  The prebuilt .so
travels to the GPU box; these tests are skipped if it is absent."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

REF_SO = os.path.join(O.ORACLE_DIR, "_ref", "libppref.so")
if not os.path.exists(REF_SO) and os.path.isdir("/root/reference/planner/src"):
    subprocess.call(["make", "-C", O.ORACLE_DIR, "ref"])
pytestmark = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(scope="module")
def ref():
    L = C.CDLL(REF_SO)
    L.ref_modulo.restype = C.c_double
    L.ref_modulo.argtypes = [C.c_double, C.c_double]
    return L


def _random_ops(rng, n, levels):
    costs = rng.randint(0, levels, size=n).astype(np.float64) / 4.0
    ops, pushed, inq = [], 0, 0
    while pushed < n or inq > 0:
        if pushed < n and (inq == 0 or rng.rand() < 0.55):
            ops.append(pushed)
            pushed += 1
            inq += 1
        else:
            ops.append(-1)
            inq -= 1
    return np.array(ops, dtype=np.int32), costs


def test_pop_order_matches_reference_frontier(ref):
    rng = np.random.RandomState(7)
    for trial in range(40):
        ops, costs = _random_ops(rng, 300, levels=3 + trial % 9)
        popped = np.empty(len(ops), dtype=np.int32)
        n = ref.ref_frontier_replay(C.c_int(len(ops)), O.iptr(ops), O.dptr(costs), O.iptr(popped))
        want = popped[:n]
        assert np.array_equal(O.frontier_replay(ops, costs, 0), want)  # (cost,-seq) heap
        assert np.array_equal(O.frontier_replay(ops, costs, 1), want)  # literal restatement


def test_reference_test_frontier_binary_passes():
    exe = os.path.join(O.ORACLE_DIR, "_ref", "test_frontier")
    if not os.path.exists(exe):
        pytest.skip("reference test binary not built")
    assert subprocess.call([exe]) == 0


def test_neighbor_order_matches_reference(ref):
    for rows, cols in ((5, 7), (1, 1), (2, 2), (3, 1)):
        for r in range(-1, rows + 1):
            for c in range(-1, cols + 1):
                n = C.c_int()
                rc = np.empty((8, 2), dtype=np.int32)
                ref.ref_neighbors(C.c_int(r), C.c_int(c), C.c_int(rows), C.c_int(cols), C.byref(n), O.iptr(rc))
                mine = O.neighbors(r, c, rows, cols)
                assert np.array_equal(mine, rc[:n.value]), (r, c, rows, cols)


def test_rng_matches_reference_random(ref):
    ref.ref_rng_uniform.argtypes = [C.c_ulonglong, C.c_longlong, C.c_double, C.c_double, C.POINTER(C.c_double)]
    for seed in (0, 1, 12345, 2**63 + 5):
        for lb, ub in ((0.0, 1.0), (-51.2, 51.2)):
            out = np.empty(1000)
            ref.ref_rng_uniform(C.c_ulonglong(seed), C.c_longlong(1000), C.c_double(lb), C.c_double(ub), O.dptr(out))
            assert np.array_equal(O.rng_uniform(seed, 1000, lb, ub), out)


def test_modulo_matches_reference(ref):
    import math
    rng = np.random.RandomState(3)
    for v in rng.uniform(-20, 20, 200):
        m = ref.ref_modulo(float(v), 2 * math.pi)
        x = math.fmod(v, 2 * math.pi)
        if x < 0:
            x += 2 * math.pi
        assert m == x
