"""Device grid A* (pp_grid_astar_batch) timing: BASELINE config 1 (one query, 128 x 128) and batches on 128^2 / 1024^2 maps, next to
the CPU oracle on one core.  Usage: python tools/bench_grid_astar.py [n_batch]"""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import pathplanning_amd as pa  # noqa: E402


def free_cells(w, rng, n):
    free = np.argwhere(w.occ() < 0)
    return free[rng.choice(len(free), n)].astype(np.int32)


def run(name, w, n, bidirectional, ctx):
    ms = pa.OccupancyMapSet.from_bounds(ctx, w.lb, w.ub, float(w.resolution))
    ms.upload_occupancy(w.occ())
    g = pa.GridAStarBatch(ms)
    rng = np.random.RandomState(1)
    inits, goals = free_cells(w, rng, n), free_cells(w, rng, n)
    g.search_batch(inits[:2], goals[:2], bidirectional=bidirectional, want_expanded=False)  # warm-up
    t0 = time.time()
    res = g.search_batch(inits, goals, bidirectional=bidirectional, want_expanded=False, max_path=4 * (w.rows + w.cols))
    dt = time.time() - t0
    nexp = sum(r["n_expanded"] + r["n_expanded_reverse"] for r in res)
    k = min(n, 16)
    t0 = time.time()
    same = 0
    for q in range(k):
        want = O.grid_astar(w, inits[q], goals[q], bidirectional=bidirectional, inner_goal_f=goals[q], inner_goal_r=inits[q])
        same += int(want["status"] == res[q]["status"] and (want["status"] != 0 or (want["cost"] == res[q]["cost"] and np.array_equal(want["path"], res[q]["path"]))))
    cpu = (time.time() - t0) / k
    print("%-34s %5d queries  %8.1f ms  %9.0f queries/s  %6.2e expansions/s  | oracle 1 core: %7.2f ms/query = %6.0f queries/s; %d/%d identical"
          % (name, n, dt * 1e3, n / dt, nexp / dt, cpu * 1e3, 1.0 / cpu, same, k))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ctx = pa.Context(0)
    w = O.World(32.0, 32.0, 0.5)
    s = 3.2
    for dx, dy, p in ((10.0 * s, 1.0, [2.0 * s, 0.0, -math.pi / 4.0]), (10.0 * s, 1.0, [0.0, 7.0 * s, -math.pi / 4.0]),
                      (10.0 * s, 1.0, [-8.0 * s, 5.0 * s, math.pi / 2.0]), (14.0 * s, 1.0, [5.0 * s, -5.0 * s, 0.0])):
        w.add_rectangle(dx, dy, p)
    ms = pa.OccupancyMapSet.from_bounds(ctx, w.lb, w.ub, 0.5)
    ms.upload_occupancy(w.occ())
    g = pa.GridAStarBatch(ms)
    g.search_batch([(2, 2)], [(120, 120)], want_expanded=False)
    for bi in (False, True):
        t0 = time.time()
        r = g.search_batch([(2, 2)], [(120, 120)], bidirectional=bi, want_expanded=False)[0]
        dt = time.time() - t0
        t0 = time.time()
        O.grid_astar(w, (2, 2), (120, 120), bidirectional=bi, inner_goal_f=(120, 120), inner_goal_r=(2, 2))
        print("config 1 %s: device %.2f ms (%d expansions, incl. workspace allocation), oracle %.2f ms"
              % ("bidirectional" if bi else "unidirectional", dt * 1e3, r["n_expanded"] + r["n_expanded_reverse"], (time.time() - t0) * 1e3))
    run("128^2 (config 1 map), uni", w, n, False, ctx)
    run("128^2 (config 1 map), bidirectional", w, n, True, ctx)
    w2 = O.synthetic_world(1024, 24, 5)
    run("1024^2, 24 outlines, uni", w2, max(n // 8, 8), False, ctx)
    run("1024^2, 24 outlines, bidirectional", w2, max(n // 8, 8), True, ctx)


if __name__ == "__main__":
    main()
