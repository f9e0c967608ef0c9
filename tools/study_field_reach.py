#!/usr/bin/env python3
"""How much of a goal's obstacle-heuristic field does its Hybrid-A* query READ?  (CPU, oracle only: a study for the pipeline's bounded
fields)  For uniformly drawn start / goal pairs on the bench map: field value at the start cell, the largest field value
over the lattice cells the search expanded (the children it evaluated lie one step further: see `slack`), and the share of the field's
rounds / cells a wavefront stopped at `value(start) + margin` would have processed."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pathplanning_amd import synthetic  # noqa: E402
import oracle_lib as O  # noqa: E402

n_q = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cells = 1024
m = synthetic.make_map(cells, 24, seed=1)
w = O.World(float(m["upper"][0]), float(m["upper"][1]), m["resolution"])
w.set_occ(m["occ"])
w.set_d2(m["d2"])
w.set_pathcost(m["path_cost"])
res = float(np.float32(m["resolution"]))
rng = np.random.RandomState(5)
lo, up = m["lower"], m["upper"]


def sample(k):
    out = np.empty((0, 3))
    while len(out) < k:
        p = np.column_stack([rng.uniform(lo[0], up[0], 4 * k), rng.uniform(lo[1], up[1], 4 * k), rng.uniform(-math.pi, math.pi, 4 * k)])
        out = np.concatenate([out, p[w.is_state_valid(p).astype(bool)]])
    return out[:k]


starts, goals = sample(n_q), sample(n_q)
table, _ = O.nonholo_build(w.lb, w.ub, O.params_array())
h = O.Hybrid(w, O.params_array(), table=table)
rows = []
for q in range(n_q):
    field, _ = w.obstacle_heuristic(goals[q, :2])
    sc = w.to_cell(starts[q:q + 1, :2])[0]
    hs = float(field[sc[0], sc[1]])
    if not np.isfinite(hs):
        continue
    r = h.search(starts[q], goals[q], q)
    ex = r["expanded"]
    # lattice index = pose / spatial resolution truncated toward zero (1 m = 10 grid cells): the cell's centre stands for the poses in it
    xs = ex[:, 0] + 0.5 * np.sign(ex[:, 0])
    ys = ex[:, 1] + 0.5 * np.sign(ex[:, 1])
    gc = w.to_cell(np.column_stack([xs, ys]))
    gx, gy = np.clip(gc[:, 0], 0, cells - 1), np.clip(gc[:, 1], 0, cells - 1)
    d0 = h.discretize(starts[q:q + 1])[0]
    fv = field[gx, gy]
    fv = fv[np.isfinite(fv)]
    top = float(fv.max()) if len(fv) else hs
    fin = field[np.isfinite(field)]
    rows.append((hs, top - hs, float(fin.max()), r["status"], len(ex), float((fin < hs + 48).mean()), tuple(d0[:2]) == tuple(ex[0, :2])))
a = np.array([(x[0], x[1], x[2], x[3], x[4], x[5]) for x in rows])
print("queries %d (anchor ok: %s)" % (len(a), all(x[6] for x in rows)))
print("field at start: mean %.0f  field max: mean %.0f  -> rounds share if stopped at start+48: %.2f" % (a[:, 0].mean(), a[:, 2].mean(), ((a[:, 0] + 48) / a[:, 2]).clip(max=1).mean()))
print("cells share discovered below start+48: mean %.2f" % a[:, 5].mean())
exc = a[:, 1]
for p in (50, 90, 95, 99, 100):
    print("excess (max expanded-cell value - value at start), percentile %3d: %7.1f cells" % (p, np.percentile(exc, p)))
for mg in (16, 32, 48, 64, 96, 128):
    print("margin %3d: queries whose expanded cells stay inside: %.3f" % (mg, (exc + 28 <= mg).mean()))
print("failures (status != 0): %d, their excess: %s" % ((a[:, 3] != 0).sum(), np.round(exc[a[:, 3] != 0], 0)))
