"""-m gpu: batched grid A* / bidirectional grid A* on the device (pp_grid_astar_batch) against the
oracle's restatement of algo/a_star.h + a_star_n2.cpp + bidirectional_a_star.h: status, cost, path cells and the whole expansion
order (hence the explored set) must be identical -- integer / index work, bit-exact; costs are sums of correctly rounded square
roots in the same order, compared with ==."""
import math

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def device_map(w, ctx=None):
    import pathplanning_amd as pa
    ctx = ctx or pa.Context(0)
    ms = pa.OccupancyMapSet.from_bounds(ctx, w.lb, w.ub, float(w.resolution))
    assert (ms.rows, ms.cols) == (w.rows, w.cols)
    ms.upload_occupancy(w.occ())
    return ctx, ms


def config1_world():
    # BASELINE config 1: 128 x 128 cells, the example script's four rectangles scaled to the 64 m map (tests/test_grid_astar_config1.py)
    w = O.World(32.0, 32.0, 0.5)
    s = 3.2
    w.add_rectangle(10.0 * s, 1.0, [2.0 * s, 0.0, -math.pi / 4.0])
    w.add_rectangle(10.0 * s, 1.0, [0.0, 7.0 * s, -math.pi / 4.0])
    w.add_rectangle(10.0 * s, 1.0, [-8.0 * s, 5.0 * s, math.pi / 2.0])
    w.add_rectangle(14.0 * s, 1.0, [5.0 * s, -5.0 * s, 0.0])
    return w


def same(got, want, bidirectional):
    assert got["status"] == want["status"], (got["status"], want["status"])
    assert np.array_equal(got["expanded"], want["explored"])
    if bidirectional:
        assert np.array_equal(got["expanded_reverse"], want["explored_reverse"])
    if want["status"] == 0:
        assert got["cost"] == want["cost"], (got["cost"], want["cost"])
        assert np.array_equal(got["path"], want["path"])
    else:
        assert math.isinf(got["cost"]) and len(got["path"]) == 0


def test_config1_on_the_device_uni_and_bidirectional():
    import pathplanning_amd as pa
    w = config1_world()
    ctx, ms = device_map(w)
    g = pa.GridAStarBatch(ms)
    init, goal = (2, 2), (120, 120)
    uni = g.search_batch([init], [goal])[0]
    same(uni, O.grid_astar(w, init, goal), False)
    assert uni["status"] == 0 and tuple(uni["path"][0]) == init and tuple(uni["path"][-1]) == goal
    # as the example script runs it: both wrapped heuristics still hold the unidirectional run's goal
    bi = g.search_batch([init], [goal], bidirectional=True, inner_goals=[[*goal, *goal]])[0]
    same(bi, O.grid_astar(w, init, goal, bidirectional=True, inner_goal_f=goal, inner_goal_r=goal), True)
    dup = sum(1 for i in range(1, len(bi["path"])) if tuple(bi["path"][i]) == tuple(bi["path"][i - 1]))
    assert dup == 1  # the meeting cell appears twice
    assert abs(bi["cost"] - uni["cost"]) < 1e-9
    # and as a caller means it: forward heuristic -> goal, reverse heuristic -> init (the default)
    bi2 = g.search_batch([init], [goal], bidirectional=True)[0]
    same(bi2, O.grid_astar(w, init, goal, bidirectional=True, inner_goal_f=goal, inner_goal_r=init), True)


def random_free_cells(w, rng, n):
    occ = w.occ()
    free = np.argwhere(occ < 0) if occ.min() < 0 else np.argwhere(occ == 0)
    return free[rng.choice(len(free), n)].astype(np.int32)


@pytest.mark.parametrize("bidirectional", [False, True])
def test_batch_of_random_queries_matches_the_oracle(bidirectional):
    import pathplanning_amd as pa
    w = O.synthetic_world(192, 6, 17)
    ctx, ms = device_map(w)
    rng = np.random.RandomState(4)
    n = 96
    inits, goals = random_free_cells(w, rng, n), random_free_cells(w, rng, n)
    inits[5] = goals[5]  # init == goal: found at the first pop (uni) / first steps (bidirectional)
    res = pa.GridAStarBatch(ms).search_batch(inits, goals, bidirectional=bidirectional)
    n_ok = 0
    for q in range(n):
        want = O.grid_astar(w, inits[q], goals[q], bidirectional=bidirectional, inner_goal_f=goals[q], inner_goal_r=inits[q])
        same(res[q], want, bidirectional)
        n_ok += want["status"] == 0
    assert n_ok >= n // 2


def test_walled_in_goal_occupied_cells_and_shortcuts():
    """A goal enclosed by a closed outline: the search explores everything it can reach and fails (a_star.h:345).  Occupied init:
    the root is pushed regardless (a_star.h:350-364).  A cluttered map makes the open list replace nodes (ProcessPossibleShortcut)."""
    import pathplanning_amd as pa
    w = O.World(6.4, 6.4, 0.1)  # 128 x 128
    w.add_rectangle(3.0, 3.0, [0.0, 0.0, 0.0])  # closed outline around the centre
    rng = np.random.RandomState(2)
    for _ in range(40):
        w.add_rectangle(rng.uniform(0.3, 1.2), rng.uniform(0.1, 0.4), [rng.uniform(-6, 6), rng.uniform(-6, 6), rng.uniform(-3, 3)])
    ctx, ms = device_map(w)
    occ = w.occ()
    occupied = np.argwhere(occ >= 0) if occ.min() < 0 else np.argwhere(occ != 0)
    inside = (64, 64)
    assert occ[inside] < 0 or occ[inside] == 0
    inits = np.array([(2, 2), (2, 2), tuple(occupied[0]), (125, 3), (64, 64)], dtype=np.int32)
    goals = np.array([inside, tuple(occupied[3]), (120, 120), (3, 125), (66, 63)], dtype=np.int32)
    g = pa.GridAStarBatch(ms)
    for bidirectional in (False, True):
        res = g.search_batch(inits, goals, bidirectional=bidirectional)
        for q in range(len(inits)):
            want = O.grid_astar(w, inits[q], goals[q], bidirectional=bidirectional, inner_goal_f=goals[q], inner_goal_r=inits[q])
            same(res[q], want, bidirectional)
        if not bidirectional:
            assert res[0]["status"] == -1 and res[0]["n_expanded"] > 1000  # walled-in goal: everything reachable was expanded
            assert res[1]["status"] == -1  # occupied goal
            assert res[4]["status"] == 0


def test_large_map_open_list_beyond_lds_and_buffer_limits():
    """1024 x 1024: the open list outgrows its LDS part (1024 entries), and a path longer than the caller's buffer is reported, not
    written out of bounds."""
    import pathplanning_amd as pa
    from pathplanning_amd import _lib
    w = O.synthetic_world(1024, 24, 5)
    ctx, ms = device_map(w)
    rng = np.random.RandomState(8)
    inits, goals = random_free_cells(w, rng, 6), random_free_cells(w, rng, 6)
    inits[0], goals[0] = (5, 5), (1018, 1015)
    g = pa.GridAStarBatch(ms)
    res = g.search_batch(inits, goals)
    for q in range(len(inits)):
        same(res[q], O.grid_astar(w, inits[q], goals[q]), False)
    assert res[0]["status"] == 0 and res[0]["n_expanded"] > 50000
    with pytest.raises(_lib.PPError):
        g.search_batch(inits[:1], goals[:1], max_path=16)
    with pytest.raises(_lib.PPError):
        g.search_batch([(5, 5)], [(5000, 5)])  # outside the map
