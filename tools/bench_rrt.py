#!/usr/bin/env python3
"""BASELINE config 3: RRT* on a 1024 x 1024 occupancy map: maxIteration = maxNumberTreeNode = 1e5,
maxConnectionDistance 2.048 m, goalBias 0.05, k = floor(ln N).  Reports samples/s, kNN queries/s, edge checks/s."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathplanning_amd as pa  # noqa: E402
from pathplanning_amd import synthetic  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = pa.Context(0)
m = synthetic.make_map(1024, 24, seed=1)
ms, val = synthetic.upload(ctx, m)
lb, ub = np.array(m["lower"][:2]), np.array(m["upper"][:2])
# a goal inside an obstacle's clearance is never connected: the loop runs all N iterations (a reachable goal ends RRT* at
# the first exact connection, a few hundred iterations -- algo/rrt_star.h:86-95)
cand = synthetic.sample_valid_poses(val, m, 1, seed=3)[0, :2]
rng = np.random.RandomState(0)
while True:
    g = rng.uniform(lb * 0.6, ub * 0.6)
    if not val.is_state_valid(np.array([[g[0], g[1], 0.0]]))[0]:
        break
print("unreachable goal", g)
for star in (True, False):
    cls = pa.RRTStar if star else pa.RRT
    r = cls(ctx, lb, ub, validator=val, max_iteration=N, max_number_tree_node=N, max_connection_distance=2.048, goal_bias=0.05)
    r.set_init_state([-46.0, -46.0])
    r.set_goal_state(g)
    r.set_seed(7)
    t = time.time()
    st = r.search_path()
    dt = time.time() - t
    res = r.result
    print("%s: status %s, %d iterations, %d nodes in %.2f s -> %.0f samples/s, %.0f kNN queries/s, %.0f edge checks/s" % (
        cls.__name__, st, res["iterations"], len(res["nodes"]), dt, res["iterations"] / dt, res["n_knn"] / dt, res["n_edge_checks"] / dt))

# ---- many independent trees at once (one workgroup per tree): the GPU-native use
NB, NI = 512, 20000
rng = np.random.RandomState(1)
inits = synthetic.sample_valid_poses(val, m, NB, seed=11)[:, :2].copy()
goals = np.tile(g, (NB, 1))
seeds = np.arange(NB, dtype=np.uint64) + 100
for star in (True, False):
    cls = pa.RRTStar if star else pa.RRT
    r = cls(ctx, lb, ub, validator=val, max_iteration=NI, max_number_tree_node=NI, max_connection_distance=2.048, goal_bias=0.05)
    t = time.time()
    out = r.search_batch(inits, goals, seeds)
    dt = time.time() - t
    it = sum(o["iterations"] for o in out)
    print("%s batch: %d trees x %d iterations in %.2f s -> %.3g samples/s, %.3g kNN queries/s, %.3g edge checks/s" % (
        cls.__name__, NB, NI, dt, it / dt, sum(o["n_knn"] for o in out) / dt, sum(o["n_edge_checks"] for o in out) / dt))

# ---- CPU oracle (brute-force kNN restatement of the reference loop, one core) on a bounded sample
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
ow = O.World(float(m["upper"][0]), float(m["upper"][1]), m["resolution"])
ow.set_occ(m["occ"])
ow.set_d2(m["d2"])
ow.set_pathcost(m["path_cost"] if "path_cost" in m else np.zeros_like(m["d2"], dtype=np.float32))
t = time.time()
want = O.rrt(ow, lb, ub, [-46.0, -46.0], g, 7, star=True, max_iteration=20000, max_nodes=20000, max_connection=2.048, goal_bias=0.05)
dt = time.time() - t
print("CPU oracle RRT*: %d iterations in %.2f s -> %.0f samples/s (1 core, brute-force kNN)" % (want["iterations"], dt, want["iterations"] / dt))
