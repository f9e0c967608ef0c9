#!/usr/bin/env python3
"""Experiment: search time of one big batch when the queries are issued longest-first (tail hidden)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathplanning_amd as pa  # noqa: E402
from pathplanning_amd import synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
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
wf, se = planner.last_timings()
ne = np.array([r.n_expanded for r in res])
print("unsorted: wavefront %.1f ms search %.1f ms, expansions %d, %.1f M exp/s" % (wf, se, ne.sum(), ne.sum() / se / 1e3))
order = np.argsort(-ne)
for name, o in (("longest first", order), ("by distance", np.argsort(-np.hypot(starts[:, 0] - goals[:, 0], starts[:, 1] - goals[:, 1])))):
    res2 = planner.search_batch(starts[o].copy(), goals[o].copy(), seeds[o].copy())
    wf, se = planner.last_timings()
    ne2 = np.array([r.n_expanded for r in res2])
    assert (ne2 == ne[o]).all()
    print("%s: wavefront %.1f ms search %.1f ms, %.1f M exp/s" % (name, wf, se, ne.sum() / se / 1e3))
