"""-m gpu: device RRT / RRT* (whole loop on the GPU) against the CPU oracle: identical trees."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_common import make_pair

pytestmark = pytest.mark.gpu
SMOKE = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "smoke_cases.json")))


def same_tree(got, want):
    assert got["status"] == want["status"]
    assert got["iterations"] == want["iterations"]
    assert len(got["nodes"]) == len(want["nodes"])
    assert np.array_equal(got["parents"], want["parents"])
    assert np.abs(got["nodes"] - want["nodes"]).max() < 1e-9 if len(got["nodes"]) else True
    fin = np.isfinite(want["costs"])
    assert np.array_equal(np.isfinite(got["costs"]), fin)
    assert np.abs(got["costs"][fin] - want["costs"][fin]).max() < 1e-9
    assert got["n_knn"] == want["n_knn"]
    assert got["n_edge_checks"] == want["n_edge_checks"]
    assert len(got["path"]) == len(want["path"])
    if len(want["path"]):
        assert np.abs(got["path"] - want["path"]).max() < 1e-9


def test_reference_smoke_cases_free_space():
    """planner/tests/test_rrt.cpp / test_rrt_star.cpp configurations."""
    import pathplanning_amd as pa
    ctx = pa.Context(0)
    c = SMOKE["rrt"]
    for seed in range(6):
        r = pa.RRT(ctx, c["bounds"][0], c["bounds"][1])
        r.set_init_state(c["start"])
        r.set_goal_state(c["goal"])
        r.set_seed(seed)
        r.search_path()
        want = O.rrt(None, c["bounds"][0], c["bounds"][1], c["start"], c["goal"], seed, star=False)
        same_tree(r.result, want)
    c = SMOKE["rrt_star"]
    n_ok = 0
    for seed in range(4):
        r = pa.RRTStar(ctx, c["bounds"][0], c["bounds"][1])
        r.set_init_state(c["start"])
        r.set_goal_state(c["goal"])
        r.set_seed(seed)
        st = r.search_path()
        want = O.rrt(None, c["bounds"][0], c["bounds"][1], c["start"], c["goal"], seed, star=True, max_iteration=10000, max_nodes=10000)
        same_tree(r.result, want)
        if st == pa.Status.SUCCESS:
            n_ok += 1
            path = r.get_path()
            assert np.hypot(*(path[0] - np.array(c["start"]))) < c["spatial_tolerance"]
            assert np.hypot(*(path[-1] - np.array(c["goal"]))) < c["spatial_tolerance"]
    assert n_ok >= 1


def test_rrt_star_on_occupancy_map():
    """SURVEY 8(d) config 3 shape (scaled down): R2 occupancy validator, kNN + choose-parent + edge checks."""
    import pathplanning_amd as pa
    w, ms, val, ctx = make_pair(256, 6, 3)
    lb, ub = w.lb[:2], w.ub[:2]
    for seed, star in ((1, True), (2, True), (3, False)):
        cls = pa.RRTStar if star else pa.RRT
        r = cls(ctx, lb, ub, validator=val, max_iteration=6000, max_number_tree_node=6000, max_connection_distance=0.512, goal_bias=0.05)
        r.set_init_state([-11.0, -11.0])
        r.set_goal_state([11.0, 11.0])
        r.set_seed(seed)
        r.search_path()
        want = O.rrt(w, lb, ub, [-11.0, -11.0], [11.0, 11.0], seed, star=star, max_iteration=6000, max_nodes=6000, max_connection=0.512, goal_bias=0.05)
        same_tree(r.result, want)
        assert len(want["nodes"]) > 100


def test_rrt_batch_equals_single_runs_and_oracle():
    """pp_rrt_run_batch: independent problems, one workgroup each; every tree equals the oracle's (and trees beyond
    2048 nodes go through the cell-list index: exact nearest / k-nearest with ties to the lower index)."""
    import pathplanning_amd as pa
    w, ms, val, ctx = make_pair(256, 6, 3)
    lb, ub = w.lb[:2], w.ub[:2]
    inits = np.array([[-11.0, -11.0], [10.0, -9.0], [0.5, 11.5], [-11.0, 11.0]])
    goals = np.array([[11.0, 11.0], [-10.0, 9.0], [3.0, -11.0], [100.0, 100.0]])  # last goal outside: runs every iteration
    seeds = np.array([5, 6, 7, 8], dtype=np.uint64)
    for star in (True, False):
        cls = pa.RRTStar if star else pa.RRT
        r = cls(ctx, lb, ub, validator=val, max_iteration=5000, max_number_tree_node=5000, max_connection_distance=0.512, goal_bias=0.05)
        got = r.search_batch(inits, goals, seeds)
        assert len(got) == 4
        for i in range(4):
            want = O.rrt(w, lb, ub, inits[i], goals[i], int(seeds[i]), star=star, max_iteration=5000, max_nodes=5000, max_connection=0.512, goal_bias=0.05)
            same_tree(got[i], want)
        assert len(got[3]["nodes"]) > 2048


def test_config3_full_size_rrt_star():
    """BASELINE config 3 at full size: RRT* on a 1024 x 1024 map, maxIteration = maxNumberTreeNode = 1e5,
    maxConnectionDistance 2.048 m, goalBias 0.05, k = floor(ln N); the goal sits inside an obstacle's clearance so no
    iteration is skipped.  The whole tree (parents, costs, counters) must equal the oracle's; on top, size-independent
    invariants: parents precede children, cost = parent cost + edge length."""
    import time
    import pathplanning_amd as pa
    from pathplanning_amd import synthetic
    ctx = pa.Context(0)
    m = synthetic.make_map(1024, 24, seed=1)
    ms, val = synthetic.upload(ctx, m)
    ow = O.World(float(m["upper"][0]), float(m["upper"][1]), m["resolution"])
    ow.set_occ(m["occ"])
    ow.set_d2(m["d2"])
    ow.set_pathcost(m["path_cost"])
    lb, ub = np.array(m["lower"][:2]), np.array(m["upper"][:2])
    rng = np.random.RandomState(0)
    while True:
        g = rng.uniform(lb * 0.6, ub * 0.6)
        if not val.is_state_valid(np.array([[g[0], g[1], 0.0]]))[0]:
            break
    N = 100000
    r = pa.RRTStar(ctx, lb, ub, validator=val, max_iteration=N, max_number_tree_node=N, max_connection_distance=2.048, goal_bias=0.05)
    r.set_init_state([-46.0, -46.0])
    r.set_goal_state(g)
    r.set_seed(7)
    t = time.time()
    r.search_path()
    gpu_s = time.time() - t
    got = r.result
    assert got["iterations"] == N + 1  # `count > maxIteration` (rrt_star.h:62, Appendix A Q15)
    nodes, parents, costs = got["nodes"], got["parents"], got["costs"]
    assert len(nodes) > 50000
    idx = np.arange(1, len(nodes))
    assert (parents[idx] < idx).all() and (parents[idx] >= 0).all()
    edge = np.hypot(*(nodes[idx] - nodes[parents[idx]]).T)
    assert np.abs(costs[idx] - (costs[parents[idx]] + edge)).max() < 1e-9
    t = time.time()
    want = O.rrt(ow, lb, ub, [-46.0, -46.0], g, 7, star=True, max_iteration=N, max_nodes=N, max_connection=2.048, goal_bias=0.05)
    cpu_s = time.time() - t
    same_tree(got, want)
    print("config 3: %d iterations, %d nodes; GPU %.2f s, CPU oracle (brute-force kNN, 1 core) %.1f s" % (got["iterations"], len(nodes), gpu_s, cpu_s))


def test_rrt_star_rewire_extension_matches_its_definition():
    """SURVEY 8f rank 4 (beyond the reference, whose RRT* never rewires): star = 2 (rewire over the k nearest) and star = 3 (radius
    near-set) on an occupancy map against the oracle's definition of the same steps: identical parents, costs, counters.  And what
    rewiring is for: tree costs never above, and mostly below, the choose-parent-only tree grown from the same samples."""
    import pathplanning_amd as pa
    w, ms, val, ctx = make_pair(256, 6, 3)
    lb, ub = w.lb[:2], w.ub[:2]
    kw = dict(max_iteration=4000, max_number_tree_node=4000, max_connection_distance=0.512, goal_bias=0.05)
    okw = dict(max_iteration=4000, max_nodes=4000, max_connection=0.512, goal_bias=0.05)
    for seed in (1, 2):
        plain = pa.RRTStar(ctx, lb, ub, validator=val, **kw)
        plain.set_init_state([-11.0, -11.0])
        plain.set_goal_state([100.0, 100.0])  # outside: every iteration runs
        plain.set_seed(seed)
        plain.search_path()
        for mode, extra, oextra in ((2, dict(rewire=True), {}), (3, dict(radius_gamma=6.0), dict(gamma=6.0))):
            r = pa.RRTStar(ctx, lb, ub, validator=val, **kw, **extra)
            r.set_init_state([-11.0, -11.0])
            r.set_goal_state([100.0, 100.0])
            r.set_seed(seed)
            r.search_path()
            want = O.rrt(w, lb, ub, [-11.0, -11.0], [100.0, 100.0], seed, star=mode, **okw, **oextra)
            same_tree(r.result, want)
            got = r.result
            # a tree: every node reaches the root, costs are sums of edge lengths
            idx = np.arange(1, len(got["nodes"]))
            fin = np.isfinite(got["costs"][idx])
            edge = np.hypot(*(got["nodes"][idx] - got["nodes"][got["parents"][idx]]).T)
            assert np.abs(got["costs"][idx][fin] - (got["costs"][got["parents"][idx]][fin] + edge[fin])).max() < 1e-9
            depth_ok = np.zeros(len(got["nodes"]), dtype=bool)
            depth_ok[0] = True
            for _ in range(len(got["nodes"])):
                new = depth_ok | depth_ok[np.maximum(got["parents"], 0)]
                if new.all() or (new == depth_ok).all():
                    depth_ok = new
                    break
                depth_ok = new
            assert depth_ok.all()
            if mode == 2:  # same near-sets as the plain run while the trees coincide; rewiring can only lower costs of shared prefixes
                assert (got["parents"] != plain.result["parents"][:len(got["parents"])]).sum() > 20 or len(got["parents"]) != len(plain.result["parents"])
                assert np.nanmean(got["costs"][np.isfinite(got["costs"])]) <= np.nanmean(plain.result["costs"][np.isfinite(plain.result["costs"])]) + 1e-9
