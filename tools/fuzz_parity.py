#!/usr/bin/env python3
"""One-off parity sweep (GPU + oracle): random maps and queries through both search kernels; every query must agree with
the oracle in status, expansion sequence, node / RNG / check counters, cost and path."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathplanning_amd as pa  # noqa: E402
import oracle_lib as O  # noqa: E402
from gpu_common import make_pair, valid_random_poses  # noqa: E402
from test_gpu_hybrid import compare  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 192
total = ok = 0
for cells, nobs, seed in ((256, 6, 11), (512, 14, 12), (384, 20, 13)):
    w, ms, val, ctx = make_pair(cells, nobs, seed)
    rng = np.random.RandomState(seed)
    starts = valid_random_poses(rng, w, N)
    goals = valid_random_poses(rng, w, N)
    seeds = rng.randint(0, 2 ** 31, N).astype(np.uint64)
    for rows in ("0", "1"):
        os.environ["PP_SEARCH_ROWS"] = rows
        os.environ["PP_SEARCH_SUSPEND_AFTER"] = "3000" if rows == "1" else "0"
        planner = pa.HybridAStarBatch(val, pa.HybridAStarSearchParameters(), max_batch=N, max_nodes=65536, search_rows=32)
        planner.initialize()
        res = planner.search_batch(starts, goals, seeds)
        h = O.Hybrid(w, O.params_array(), table=planner.nonholo_table())
        n_ok = compare(planner, res, h, starts, goals, seeds)
        total += N
        ok += N
        print("map %dx%d seed %d, kernel rows=%s: %d queries identical to the oracle (%d solved, max %d expansions)" % (
            cells, cells, seed, rows, N, n_ok, max(r.n_expanded for r in res)))
        planner.close()
print("TOTAL %d/%d" % (ok, total))
