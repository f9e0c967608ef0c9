#!/usr/bin/env python3
"""BASELINE config 2: ONE Hybrid-A* query on a 512 x 512 map ((-23.04,-23.04,0) -> (23.04,23.04,0), seed
12345), P = 10 (reference default), P = 74 (numGeneratedMotion = 37, the reference-reachable count next to "72") and P = 72 through
an explicit table of 36 steering angles (pp_planner_set_primitives; the oracle gets the same list in place of m_deltas).
Latency of the device path (wavefront + graph search) beside the CPU oracle on one core."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathplanning_amd as pa  # noqa: E402
from pathplanning_amd import synthetic  # noqa: E402
import oracle_lib as O  # noqa: E402

ctx = pa.Context(0)
m = synthetic.make_map(512, 12, seed=1)
ms, val = synthetic.upload(ctx, m)
ow = O.World(float(m["upper"][0]), float(m["upper"][1]), m["resolution"])
ow.set_occ(m["occ"])
ow.set_d2(m["d2"])
ow.set_pathcost(m["path_cost"])
start = np.array([[-23.04, -23.04, 0.0]])
goal = np.array([[23.04, 23.04, 0.0]])
seeds = np.array([12345], dtype=np.uint64)
import math  # noqa: E402
for ngm in (5, 37, "72 explicit"):
    P = pa.HybridAStarSearchParameters(num_generated_motion=ngm if isinstance(ngm, int) else 5)
    planner = pa.HybridAStarBatch(val, P, max_batch=1, max_nodes=131072)
    deltas = None
    if not isinstance(ngm, int):
        delta_max = math.atan(P.wheelbase / P.min_turning_radius)
        deltas = np.linspace(-delta_max, delta_max, 36)
        planner.set_primitives(deltas)
    planner.initialize()
    for it in range(3):
        t = time.time()
        res = planner.search_batch(start, goal, seeds)
        wall = (time.time() - t) * 1e3
        wf, se = planner.last_timings()
    h = O.Hybrid(ow, O.params_array(num_generated_motion=ngm if isinstance(ngm, int) else 5), table=planner.nonholo_table())
    if deltas is not None:
        h.set_deltas(deltas)
    t = time.time()
    r = h.search(start[0], goal[0], 12345)
    cpu = (time.time() - t) * 1e3
    g = res[0]
    same = g.status == r["status"] and g.n_expanded == len(r["expanded"]) and np.array_equal(planner.get_expanded_of(0), r["expanded"])
    print("P=%d: status %d, %d expansions, cost %.6f | GPU wavefront %.2f ms + search %.2f ms (call %.1f ms) | CPU oracle 1 core %.1f ms | same expansions: %s" % (
        planner.num_primitives, g.status, g.n_expanded, g.cost, wf, se, wall, cpu, same))
