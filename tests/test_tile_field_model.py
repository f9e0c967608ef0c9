"""CPU: the tile-fixpoint statement of ObstaclesHeuristic::Update (tests/cpp/model_tile_field.cpp -- the algorithm the GPU kernel
k_wavefront_tiles runs, pathplanning_amd/csrc/pp_wavefront_tiles.hip) against the oracle's sequential restatement of
algo/heuristics.cpp:106-153.  The model is test infrastructure: it pins the ALGORITHM (any order of solving tiles reaches the
reference's field, bit for bit, unless a straight / diagonal tie is reported) on the CPU, where no GPU is needed; the kernel itself is
compared with the oracle in tests/test_gpu_wavefront_tiles.py and tests/test_gpu_parity.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cpp", "model_tile_field.cpp")
SO = os.path.join(HERE, "cpp", "_model_tile_field.so")


@pytest.fixture(scope="module")
def model():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", SO, SRC])
    lib = C.CDLL(SO)

    def run(occ8, goal_row, goal_col, max_visits=64):
        rows, cols = occ8.shape
        cost = np.empty((rows, cols), dtype=np.float32)
        st = np.zeros(16, dtype=np.int64)
        rc = lib.pp_model_tile_field(occ8.ctypes.data_as(C.POINTER(C.c_uint8)), rows, cols, int(goal_row), int(goal_col), cost.ctypes.data_as(C.POINTER(C.c_float)),
                                     st.ctypes.data_as(C.POINTER(C.c_int64)), max_visits)
        return cost, dict(visits=int(st[0]), rounds=int(st[1]), passes=int(st[2]), cells=int(st[3]), tie=int(st[5]), unsettled=int(st[6]), tiles=int(st[7]), max_visits=int(st[8])), rc
    return run


def check(model, w, occ, goals):
    occ8 = np.ascontiguousarray((occ >= 0).astype(np.uint8))
    out = []
    for g in goals:
        cell = np.asarray(w.to_cell(g)).reshape(-1)
        ref, explored = w.obstacle_heuristic(g)
        cost, st, rc = model(occ8, cell[0], cell[1])
        assert rc == 0 and st["tie"] == 0 and st["unsettled"] == 0, (g, st)
        assert np.array_equal(ref.view(np.uint32), cost.view(np.uint32)), (g, int((ref.view(np.uint32) != cost.view(np.uint32)).sum()))
        assert np.array_equal(np.isfinite(cost), explored.astype(bool))
        out.append(st)
    return out


def test_benchmark_style_map_and_work_figures(model):
    """512 x 512, 12 rectangle outlines (SURVEY 8d layout): identical fields, and the work figures the kernel is sized with -- about 1.2
    visits per tile (1.0 on an empty map: tiles are taken in cost order, each after its upwind neighbours) and ~90 rounds per visit."""
    w = O.synthetic_world(512, 12, 7)
    rng = np.random.RandomState(1)
    st = check(model, w, w.occ(), rng.uniform(-25.6, 25.6, (6, 2)))
    visits = sum(s["visits"] for s in st) / sum(s["tiles"] for s in st)
    assert 1.0 <= visits < 1.6, visits
    w0 = O.World(12.8, 12.8, 0.1)
    occ0 = np.full((w0.rows, w0.cols), -1, dtype=np.int32)
    w0.set_occ(occ0)
    st0 = check(model, w0, occ0, [(0.3, -4.0), (-12.79, 12.79)])
    assert all(s["visits"] == s["tiles"] for s in st0), st0


@pytest.mark.parametrize("density", [0.05, 0.28, 0.4])
def test_cluttered_maps(model, density):
    """Random clutter (enclosed pockets, the corner rule at every other cell, fronts meeting inside tiles): identical fields."""
    rng = np.random.RandomState(int(density * 100))
    w = O.World(9.6, 12.8, 0.1)
    occ = np.where(rng.uniform(size=(w.rows, w.cols)) < density, 0, -1).astype(np.int32)
    w.set_occ(occ)
    goals = list(rng.uniform([-9.6, -12.8], [9.6, 12.8], (4, 2)))
    occupied = np.argwhere(occ >= 0)
    r, c = occupied[len(occupied) // 3]  # a goal on an occupied cell is pushed all the same (heuristics.cpp:119-121)
    goals.append((-9.6 + (r + 0.5) * 0.1, -12.8 + (c + 0.5) * 0.1))
    check(model, w, occ, goals)


def test_ragged_grid_and_goal_outside(model):
    w = O.World(10.0, 3.35, 0.1)  # 200 x 67 cells: partial tiles on both edges
    rng = np.random.RandomState(3)
    occ = np.where(rng.uniform(size=(w.rows, w.cols)) < 0.1, 0, -1).astype(np.int32)
    w.set_occ(occ)
    check(model, w, occ, [(-9.99, -3.3), (9.95, 3.3), (0.0, 0.0)])
    cost, st, rc = model(np.ascontiguousarray((occ >= 0).astype(np.uint8)), -1, -1)  # WorldPositionToGridCell failed: the field stays +inf
    assert rc == 0 and np.isinf(cost).all()
