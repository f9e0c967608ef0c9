#!/usr/bin/env python3
"""Diagnostics for the batched Hybrid-A* search: expansion-count distribution, per-kernel time."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathplanning_amd as pa  # noqa: E402
from pathplanning_amd import synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = pa.Context(0)
m = synthetic.make_map(1024, 24, seed=1)
ms, val = synthetic.upload(ctx, m)
planner = pa.HybridAStarBatch(val, pa.HybridAStarSearchParameters(), max_batch=B, max_nodes=81920)
planner.initialize()
reach = synthetic.reachable_mask(val, m)
starts = synthetic.sample_valid_poses(val, m, B, seed=1000, reachable=reach)
goals = synthetic.sample_valid_poses(val, m, B, seed=2000, reachable=reach)
seeds = np.arange(B, dtype=np.uint64)
for it in range(2):
    t = time.time()
    res = planner.search_batch(starts, goals, seeds)
    dt = time.time() - t
    wf, se = planner.last_timings()
    print("iter", it, "wall %.1f ms wavefront %.1f ms search %.1f ms" % (dt * 1e3, wf, se))
ne = np.array([r.n_expanded for r in res])
st = np.array([r.status for r in res])
nn = np.array([r.n_nodes for r in res])
sc = np.array([r.n_state_checks for r in res])
print("status counts", {int(k): int((st == k).sum()) for k in np.unique(st)})
print("n_expanded: mean %.0f median %.0f p90 %.0f p99 %.0f max %d sum %d" % (ne.mean(), np.median(ne), np.percentile(ne, 90), np.percentile(ne, 99), ne.max(), ne.sum()))
print("failed n_expanded:", sorted(ne[st != 0].tolist())[-10:])
print("success n_expanded max:", ne[st == 0].max(), "nodes max", nn.max())
print("search us per expansion of the longest query: %.2f" % (se * 1e3 / ne.max()))
print("aggregate expansions/s: %.3g ; state checks/s %.3g" % (ne.sum() / (se * 1e-3), sc.sum() / (se * 1e-3)))
worst = np.argsort(ne)[-5:]
for q in worst:
    print("q", q, "status", st[q], "exp", ne[q], "start", starts[q], "goal", goals[q], "dist", np.hypot(*(starts[q][:2] - goals[q][:2])))

# ---- phase breakdown (stamped diagnostic build of the kernel)
import ctypes as C
from pathplanning_amd._lib import check, ptr
check(planner.lib.pp_planner_set_profiling(planner.h, 1))
res = planner.search_batch(starts, goals, seeds)
wf, se = planner.last_timings()
cyc = np.zeros((B, 8), dtype=np.uint64)
check(planner.lib.pp_planner_phase_cycles(planner.h, B, ptr(cyc)))
names = ["pop", "load", "heur", "child", "dup", "insert", "write", "rs"]
ne = np.array([r.n_expanded for r in res])
tot = cyc.sum(0).astype(np.float64)
print("profiled search %.1f ms; phase share of wave-cycles (all queries):" % se)
for nm, t in zip(names, tot):
    print("  %-7s %5.1f %%   %.0f cycles/expansion" % (nm, 100 * t / tot.sum(), t / ne.sum()))
qw = int(np.argmax(ne))
print("longest query", qw, "expansions", ne[qw], "cycles/expansion by phase:", {n: int(c / ne[qw]) for n, c in zip(names, cyc[qw])})
