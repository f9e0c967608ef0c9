#!/usr/bin/env python3
"""Where do the device's and the oracle's smoothing descents part?  Runs the queries of tests/test_gpu_postprocess.py's
0.1 m case with the iteration cap swept and prints, per cap, how far apart the two sides' points are -- a smooth exponential
growth from 1e-16 is the descent's own instability, a jump is a discrete event (a point crossing a cell boundary) or a defect.
usage: diag_smoother_divergence.py [query] [spacing]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from gpu_common import make_pair, valid_random_poses  # noqa: E402
import pathplanning_amd as pa  # noqa: E402

q = int(sys.argv[1]) if len(sys.argv) > 1 else 23
spacing = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
w, ms, val, ctx = make_pair(256, 6, 3)
ms.upload_nearest_cells(*O.world_nearest(w))
rng = np.random.RandomState(5)
n = 24
starts, goals = valid_random_poses(rng, w, n), valid_random_poses(rng, w, n)
goals[0] = starts[0]
seeds = np.arange(n, dtype=np.uint64) + 17
planner = pa.HybridAStarBatch(val, pa.HybridAStarSearchParameters(), max_batch=n, max_nodes=32768)
planner.initialize()
res = planner.search_batch(starts, goals, seeds)
h = O.Hybrid(w, O.params_array(), table=planner.nonholo_table())
r = h.search(starts[q], goals[q], int(seeds[q]))
prev = None
for iters in (1, 2, 3, 5, 8, 12, 20, 30, 50, 75, 100, 150, 200, 300, 400, 600, 800, 1000, 1300, 1600, 2000):
    post = planner.postprocess(path_interpolation=spacing, smoother=dict(max_iterations=iters))
    g = planner.get_processed_path(q)
    sp = O.smoother_array(max_curvature=0.5, max_iterations=iters)
    want = O.postprocess(w, r, goals[q], O.params_array(), spacing, sp)
    d = np.abs(g["smoothed"][:, :2] - want["smoothed"][:, :2])
    i = int(np.nanargmax(d.max(axis=1)))
    moved = np.abs(want["smoothed"][:, :2] - want["resampled"][:, :2]).max()
    print("iterations %5d  status dev %2d oracle %2d  max |dev - oracle| %.3e at point %d of %d   (oracle moved its points by up to %.3e)" % (
        iters, post[q].smoothing_status, want["status"], float(np.nanmax(d)), i, len(d), moved))
