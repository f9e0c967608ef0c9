#!/usr/bin/env python3
"""Which of the bench's queries are the lattice-exhausting ones (status -1 after ~65 k expansions, the tail of every run)?  Clearance
(obstacle distance at the start and goal cells) and field value at the start, failures against everything else."""
import os
import sys

import numpy as np
import torch

torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathplanning_amd as pa  # noqa: E402
from pathplanning_amd import synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = pa.Context(0)
m = synthetic.make_map(1024, 24, seed=1)
ms, val = synthetic.upload(ctx, m)
reach = synthetic.reachable_mask(val, m)
starts = synthetic.sample_valid_poses(val, m, B, seed=1000, reachable=reach)
goals = synthetic.sample_valid_poses(val, m, B, seed=2000, reachable=reach)
seeds = np.arange(B, dtype=np.uint64)
pl = pa.HybridAStarBatch(val, pa.HybridAStarSearchParameters(), max_batch=B, max_nodes=81920)
pl.initialize()
res = pl.search_batch(starts, goals, seeds)
a = np.frombuffer(res, dtype=np.dtype(type(res[0])), count=B)
status, nexp = a["status"].copy(), a["n_expanded"].copy()
res_m = float(ms.resolution)


def cell(p):
    r = np.clip(((p[:, 0] - ms.grid_origin[0]) / res_m).astype(np.int64), 0, ms.rows - 1)
    c = np.clip(((p[:, 1] - ms.grid_origin[1]) / res_m).astype(np.int64), 0, ms.cols - 1)
    return r, c


ds = np.sqrt(m["d2"][cell(starts)].astype(np.float64)) * res_m
dg = np.sqrt(m["d2"][cell(goals)].astype(np.float64)) * res_m
fail = status != 0
print("queries %d, failures %d, expansions of failures: %s" % (B, fail.sum(), np.sort(nexp[fail])[-24:]))
print("clearance at start [m]: failures %s" % np.round(np.sort(ds[fail]), 2))
print("clearance at goal  [m]: failures %s" % np.round(np.sort(dg[fail]), 2))
for nm, d in (("start", ds), ("goal", dg), ("min(start, goal)", np.minimum(ds, dg))):
    print("%-17s all: 10/50/90 %% = %.2f / %.2f / %.2f   failures: 10/50/90 %% = %.2f / %.2f / %.2f" % ((nm,) + tuple(np.percentile(d, (10, 50, 90))) + tuple(np.percentile(d[fail], (10, 50, 90)))))
long_ = nexp >= 20000
print("queries with >= 20 k expansions: %d; their min clearance 10/50/90 %% = %s" % (long_.sum(), np.round(np.percentile(np.minimum(ds, dg)[long_], (10, 50, 90)), 2)))
for thr in (1.2, 1.5, 2.0, 3.0):
    sel = np.minimum(ds, dg) < thr
    print("min clearance < %.1f m: %.1f %% of the queries, %.0f %% of the failures, %.0f %% of the >= 20 k-expansion queries" % (thr, 100 * sel.mean(), 100 * sel[fail].mean(), 100 * sel[long_].mean()))
# the failures no clearance test catches: what do they look like?
from pathplanning_amd.planner import ObstaclesHeuristic  # noqa: E402
odd = np.where(fail & (np.minimum(ds, dg) >= 2.0))[0]
for q in odd:
    f = ObstaclesHeuristic(ms).update(goals[q:q + 1, :2])[0]
    r, c = cell(starts[q:q + 1])
    eu = float(np.hypot(*(starts[q, :2] - goals[q, :2])))
    print("query %d: start %s (clearance %.2f) goal %s (clearance %.2f): field at start %.1f cells = %.1f m, straight line %.1f m, %d expansions" % (
        q, np.round(starts[q], 2), ds[q], np.round(goals[q], 2), dg[q], f[r[0], c[0]], f[r[0], c[0]] * res_m, eu, nexp[q]))
    # clearance along the straight line and around the goal: is the goal in a pocket whose entrance is narrower than the car needs?
    rr, cc = cell(goals[q:q + 1])
    win = m["d2"][max(rr[0] - 40, 0):rr[0] + 41, max(cc[0] - 40, 0):cc[0] + 41]
    print("   obstacle distance within 4 m of the goal: min %.2f m, max %.2f m; field values within 4 m of the goal: max %.1f" % (
        np.sqrt(win.min()) * res_m, np.sqrt(win.max()) * res_m, np.nanmax(np.where(np.isfinite(f[max(rr[0] - 40, 0):rr[0] + 41, max(cc[0] - 40, 0):cc[0] + 41]), f[max(rr[0] - 40, 0):rr[0] + 41, max(cc[0] - 40, 0):cc[0] + 41], np.nan))))
# long successes: does the straight-line distance tell?
eu = np.hypot(starts[:, 0] - goals[:, 0], starts[:, 1] - goals[:, 1])
ok_long = (status == 0) & (nexp >= 20000)
print("successes with >= 20 k expansions: %d; straight-line distance 10/50/90 %% = %s m (all queries: %s)" % (
    ok_long.sum(), np.round(np.percentile(eu[ok_long], (10, 50, 90)), 1), np.round(np.percentile(eu, (10, 50, 90)), 1)))
for d in (60, 70, 80, 90):
    sel = eu > d
    print("distance > %d m: %.1f %% of the queries, %.0f %% of the long successes, %.0f %% of all expansions" % (d, 100 * sel.mean(), 100 * sel[ok_long].mean(), 100 * nexp[sel].sum() / nexp.sum()))
print("rank correlation of expansions with distance: %.2f" % np.corrcoef(np.argsort(np.argsort(eu)), np.argsort(np.argsort(nexp)))[0, 1])
