"""BASELINE config 1: grid A* on a 128x128 occupancy map through the pybind11 module -- `initialize_size(64, 64)` at
0.5 m, 4 rectangle outlines, (2,2) -> (120,120), Euclidean edge cost and heuristic as Python callables per edge, exactly as
interfaces/python/scripts/example_a_star_grid.py:46-66 drives the reference.  CPU only ("no GPU" by the north star): the
product side is pathplanning_amd/host/a_star.hpp; the checker is the oracle's restatement of algo/a_star.h +
a_star_n2.cpp + bidirectional_a_star.h.  Compared: path cells, cost, expansion order (hence the explored set)."""
import importlib
import math
import os
import sys
import time

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nav():
    from pathplanning_amd import build
    build.build_pyplanning(verbose=False)
    sys.path.insert(0, os.path.join(ROOT, "pathplanning_amd", "lib"))
    return importlib.import_module("pyplanning")


def euclid(a, b):
    return math.sqrt((a.row - b.row) ** 2 + (a.col - b.col) ** 2)


def config1_world():
    w = O.World(32.0, 32.0, 0.5)
    assert (w.rows, w.cols) == (128, 128)
    # the example script's four rectangles (example_a_star_grid.py:18-43), scaled from its 20 m map to the 64 m one
    s = 3.2
    w.add_rectangle(10.0 * s, 1.0, [2.0 * s, 0.0, -math.pi / 4.0])
    w.add_rectangle(10.0 * s, 1.0, [0.0, 7.0 * s, -math.pi / 4.0])
    w.add_rectangle(10.0 * s, 1.0, [-8.0 * s, 5.0 * s, math.pi / 2.0])
    w.add_rectangle(14.0 * s, 1.0, [5.0 * s, -5.0 * s, 0.0])
    return w


def product_map(nav, w):
    m = nav.OccupancyMap(float(w.resolution))
    m.initialize_size(w.rows * float(w.resolution), w.cols * float(w.resolution))
    assert (m.rows(), m.columns()) == (w.rows, w.cols)
    m.set_grids(w.occ(), np.full((w.rows, w.cols), 2**31 - 1, np.int32), np.zeros((w.rows, w.cols), np.float32))
    return m


def cells(seq):
    return np.array([(c.row, c.col) for c in seq], dtype=np.int32).reshape(-1, 2)


def test_config1_unidirectional_matches_oracle(nav):
    w = config1_world()
    m = product_map(nav, w)
    init, goal = (2, 2), (120, 120)
    algo = nav.AStarN2()
    algo.set_init_state(nav.GridCellPosition(*init))
    algo.set_goal_state(nav.GridCellPosition(*goal))
    assert algo.search_path() == nav.Status.FAILURE  # not initialised (a_star.h:330-333)
    assert algo.initialize(nav.AStarStatePropagatorFcnN2(m, euclid), nav.AStarHeuristicFcnN2(euclid))
    t0 = time.time()
    assert algo.search_path() == nav.Status.SUCCESS
    ms = (time.time() - t0) * 1e3
    want = O.grid_astar(w, init, goal)
    assert want["status"] == 0
    assert np.array_equal(cells(algo.get_path()), want["path"])
    assert algo.get_optimal_cost() == want["cost"]
    assert np.array_equal(cells(algo.get_expansion_order()), want["explored"])
    explored = {(c.row, c.col) for c in algo.get_explored_states()}
    assert explored == {tuple(c) for c in want["explored"]} | {init}
    print("config 1: %.1f ms through pybind11 callbacks, %d cells explored, cost %.6f" % (ms, len(explored), algo.get_optimal_cost()))


def test_config1_bidirectional_matches_oracle_and_repeats_the_meeting_cell(nav):
    w = config1_world()
    m = product_map(nav, w)
    init, goal = (2, 2), (120, 120)
    prop = nav.AStarStatePropagatorFcnN2(m, euclid)
    h = nav.AStarHeuristicFcnN2(euclid)
    # as in the example (:61-66, :103-109): the unidirectional run leaves its goal in the shared heuristic object
    uni = nav.AStarN2()
    uni.set_init_state(nav.GridCellPosition(*init))
    uni.set_goal_state(nav.GridCellPosition(*goal))
    uni.initialize(prop, h)
    assert uni.search_path() == nav.Status.SUCCESS
    hf, hr = nav.BidirectionalAStarN2.get_average_heuristic_pair(h, h)
    bi = nav.BidirectionalAStarN2()
    bi.set_init_state(nav.GridCellPosition(*init))
    bi.set_goal_state(nav.GridCellPosition(*goal))
    assert bi.initialize(prop, prop, hf, hr)
    assert bi.search_path() == nav.Status.SUCCESS
    want = O.grid_astar(w, init, goal, bidirectional=True, inner_goal_f=goal, inner_goal_r=goal)
    assert want["status"] == 0
    path = cells(bi.get_path())
    assert np.array_equal(path, want["path"])
    assert bi.get_optimal_cost() == want["cost"]
    fo, ro = bi.get_expansion_orders()
    assert np.array_equal(cells(fo), want["explored"]) and np.array_equal(cells(ro), want["explored_reverse"])
    dup = sum(1 for i in range(1, len(path)) if tuple(path[i]) == tuple(path[i - 1]))
    assert dup == 1
    assert abs(bi.get_optimal_cost() - uni.get_optimal_cost()) < 1e-9
    fe, re_ = bi.get_explored_states()
    assert {(c.row, c.col) for c in fe} == {tuple(c) for c in want["explored"]} | {init}
    assert {(c.row, c.col) for c in re_} == {tuple(c) for c in want["explored_reverse"]} | {goal}


def test_example_script_layout_and_python_subclasses(nav):
    """40x40 layout of the example script; heuristic and propagator subclassed in Python (the trampolines of pyplanning.cpp:130-151)."""
    w = O.World(10.0, 10.0, 0.5)
    w.add_rectangle(10.0, 1.0, [2.0, 0.0, -math.pi / 4.0])
    w.add_rectangle(10.0, 1.0, [0.0, 7.0, -math.pi / 4.0])
    w.add_rectangle(10.0, 1.0, [-8.0, 5.0, math.pi / 2.0])
    w.add_rectangle(14.0, 1.0, [5.0, -5.0, 0.0])
    m = product_map(nav, w)

    class H(nav.AStarHeuristicN2):
        def __init__(self):
            super().__init__()
            self.goal = None

        # the trampolines look overrides up by the C++ method name (PYBIND11_OVERRIDE_PURE, as the reference's do)
        def SetGoal(self, g):
            self.goal = g

        def GetHeuristicValue(self, s):
            return euclid(s, self.goal)

    algo = nav.AStarN2()
    algo.set_init_state(nav.GridCellPosition(1, 1))
    algo.set_goal_state(nav.GridCellPosition(35, 35))
    h = H()  # the Python object must outlive the search: the C++ side only holds the C++ half
    assert algo.initialize(nav.AStarStatePropagatorFcnN2(m, euclid), h)
    assert algo.search_path() == nav.Status.SUCCESS
    want = O.grid_astar(w, (1, 1), (35, 35))
    assert np.array_equal(cells(algo.get_path()), want["path"]) and algo.get_optimal_cost() == want["cost"]
    assert np.array_equal(cells(algo.get_expansion_order()), want["explored"])
    # unreachable goal: enclosed cell -> Failure after exhausting the reachable cells, infinite cost, empty path
    occ = w.occ().copy()
    occ[19:22, 19:22] = 0
    occ[20, 20] = -1
    w.set_occ(occ)
    m.set_grids(occ, np.zeros_like(occ), np.zeros(occ.shape, np.float32))
    algo.set_goal_state(nav.GridCellPosition(20, 20))
    assert algo.search_path() == nav.Status.FAILURE
    assert algo.get_path() == [] and algo.get_optimal_cost() == math.inf
    assert O.grid_astar(w, (1, 1), (20, 20))["status"] != 0


def test_cpp_mirror_of_reference_test_a_star(tmp_path):
    """planner/tests/test_a_star.cpp on the host engine, asserts live (tests/cpp/test_a_star.cpp)."""
    import subprocess
    from pathplanning_amd import build
    build.build(verbose=False)
    exe = str(tmp_path / "test_a_star")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "test_a_star.cpp"), "-o", exe,
                           "-L" + build.LIB_DIR, "-lpphip", "-Wl,-rpath," + build.LIB_DIR])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
