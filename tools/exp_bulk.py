#!/usr/bin/env python3
"""Experiment: search throughput on queries of bounded length (no long tail), for the selected search kernel."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathplanning_amd as pa  # noqa: E402
from pathplanning_amd import synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
CAP = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
ctx = pa.Context(0)
m = synthetic.make_map(1024, 24, seed=1)
ms, val = synthetic.upload(ctx, m)
planner = pa.HybridAStarBatch(val, pa.HybridAStarSearchParameters(), max_batch=B, max_nodes=81920)
planner.initialize()
reach = synthetic.reachable_mask(val, m)
starts = synthetic.sample_valid_poses(val, m, B, seed=1000, reachable=reach)
goals = synthetic.sample_valid_poses(val, m, B, seed=2000, reachable=reach)
seeds = np.arange(B, dtype=np.uint64)
res = planner.search_batch(starts, goals, seeds)
ne = np.array([r.n_expanded for r in res])
keep = np.nonzero(ne <= CAP)[0]
idx = np.resize(keep, B)
for it in range(2):
    res2 = planner.search_batch(starts[idx].copy(), goals[idx].copy(), seeds[idx].copy())
    wf, se = planner.last_timings()
tot = ne[idx].sum()
print("rows=%s cap %d: %d queries, %d expansions, search %.1f ms -> %.1f M exp/s, %.1f us/query" % (os.environ.get("PP_SEARCH_ROWS", "0"), CAP, B, tot, se, tot / se / 1e3, se * 1e3 / B))
