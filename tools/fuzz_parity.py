#!/usr/bin/env python3
"""One-off parity sweep (GPU + oracle): random maps and queries through both search kernels and through the streaming pipeline
(field slots recycled six times); every query must agree with the oracle in status, expansion sequence, node / RNG / check counters,
cost and path."""
import time
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
    # the same queries through the streaming pipeline: N / 6 slots, so every slot is used about six times
    os.environ.pop("PP_SEARCH_ROWS", None)
    os.environ.pop("PP_SEARCH_SUSPEND_AFTER", None)
    pipe = pa.HybridAStarPipeline(val, capacity=max(8, N // 6), max_nodes=65536, search_rows=32, log_expansions=True)
    pipe.initialize()
    h = O.Hybrid(w, O.params_array(), table=pipe.nonholo_table())
    index_of, nxt, done, t0, solved = {}, 0, 0, time.time(), 0
    while done < N:
        if nxt < N and pipe.free_slots() > 0:
            for i, t in enumerate(pipe.submit(starts[nxt:], goals[nxt:], seeds[nxt:])):
                index_of[int(t)] = nxt + i
            nxt = len(index_of)
        tickets, res = pipe.poll(256, release=False)
        for i, t in enumerate(tickets):
            q, g = index_of[int(t)], res[i]
            r = h.search(starts[q], goals[q], int(seeds[q]))
            assert g.status == r["status"] and g.n_expanded == len(r["expanded"]) and np.array_equal(pipe.get_expanded_of(int(t)), r["expanded"]), q
            assert (g.n_nodes, g.n_rng_draws, g.n_rs_attempts, g.n_state_checks, g.n_path_checks) == (r["n_nodes"], r["n_rng_draws"], r["n_rs_attempts"], r["n_state_checks"], r["n_path_checks"]), q
            if r["status"] == 0:
                solved += 1
                path = pipe.get_path_of(int(t))
                assert abs(g.cost - r["cost"]) < 1e-5 and len(path["poses"]) == len(r["path_poses"]) and np.abs(path["poses"] - r["path_poses"]).max() < 1e-5, q
            done += 1
        if len(tickets):
            pipe.release(tickets)
        assert time.time() - t0 < 900, "pipeline stalled"
    pipe.close()
    total += N
    ok += N
    print("map %dx%d seed %d, pipeline (%d slots): %d queries identical to the oracle (%d solved)" % (cells, cells, seed, max(8, N // 6), N, solved))
print("TOTAL %d/%d" % (ok, total))
