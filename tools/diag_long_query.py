"""Diagnostic: one long query of the full-size batch (tests/test_gpu_fullsize.py) -- where does its expansion sequence leave the oracle's?
    python tools/diag_long_query.py <query index>"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_lib as O
import pathplanning_amd as pa
from pathplanning_amd import synthetic

qs = [int(x) for x in sys.argv[1:]] or [776]
B = 4096
ctx = pa.Context(0)
m = synthetic.make_map(1024, 24, seed=1)
ms, val = synthetic.upload(ctx, m)
params = pa.HybridAStarSearchParameters()
planner = pa.HybridAStarBatch(val, params, max_batch=B, max_nodes=81920, search_rows=1024)
planner.initialize()
reach = synthetic.reachable_mask(val, m)
starts = synthetic.sample_valid_poses(val, m, B, seed=1000, reachable=reach)
goals = synthetic.sample_valid_poses(val, m, B, seed=2000, reachable=reach)
seeds = np.arange(B, dtype=np.uint64)
res = planner.search_batch(starts, goals, seeds)
ow = O.World(float(m["upper"][0]), float(m["upper"][1]), m["resolution"])
ow.set_occ(m["occ"])
ow.set_d2(m["d2"])
ow.set_pathcost(m["path_cost"])
h = O.Hybrid(ow, O.params_array(), table=planner.nonholo_table())
single = pa.HybridAStarBatch(val, params, max_batch=1, max_nodes=81920)
single.initialize(planner.nonholo_table())
for q in qs:
    r = h.search(starts[q], goals[q], int(seeds[q]))
    g = planner.get_expanded_of(q)
    o = r["expanded"]
    k = min(len(g), len(o))
    d = np.flatnonzero((g[:k] != o[:k]).any(axis=1))
    print("query", q, "status", res[q].status, r["status"], "n_expanded", res[q].n_expanded, len(o), "rng", res[q].n_rng_draws, r["n_rng_draws"], "rs", res[q].n_rs_attempts, r["n_rs_attempts"],
          "first difference at", int(d[0]) if len(d) else None, "of", k, "differing entries", len(d))
    if len(d):
        i = int(d[0])
        print(" gpu   ", g[max(0, i - 2):i + 4].tolist())
        print(" oracle", o[max(0, i - 2):i + 4].tolist())
        # is it a permutation nearby (order) or different cells?
        print(" same multiset of cells:", sorted(map(tuple, g.tolist())) == sorted(map(tuple, o.tolist())))
    rs = single.search_batch(starts[q:q + 1], goals[q:q + 1], seeds[q:q + 1])
    gs = single.get_expanded_of(0)
    ks = min(len(gs), len(o))
    ds = np.flatnonzero((gs[:ks] != o[:ks]).any(axis=1))
    print(" one-query kernel alone: n_expanded", rs[0].n_expanded, "first difference", int(ds[0]) if len(ds) else None)
    # the two search trees node by node (creation order)
    import ctypes as C
    from pathplanning_amd._lib import check, ptr
    n = rs[0].n_nodes
    gp, gpose, gcost, gdead = np.zeros(n, dtype=np.int32), np.zeros((n, 3)), np.zeros((n, 2)), np.zeros(n, dtype=np.int32)
    check(single.lib.pp_planner_debug_nodes(single.h, 0, n, ptr(gp), ptr(gpose), ptr(gcost), ptr(gdead)))
    L = O.lib()
    on = r["n_nodes"]
    op, opose, ocost, odead = np.zeros(on + 8, dtype=np.int32), np.zeros((on + 8, 3)), np.zeros((on + 8, 2)), np.zeros(on + 8, dtype=np.int32)
    L.ppo_hybrid_nodes(h.h, O.iptr(op), O.dptr(opose), O.dptr(ocost), O.iptr(odead))
    print(" nodes: gpu", n, "oracle", on)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "long_query_%d.npz" % q), gp=gp, gpose=gpose, gcost=gcost, gdead=gdead, gexp=gs, oexp=o, op=op[:on], opose=opose[:on],
                        ocost=ocost[:on], odead=odead[:on], start=starts[q], goal=goals[q])
    k = min(n, int(on))
    bad = np.flatnonzero((gp[:k] != op[:k]) | (np.abs(gpose[:k] - opose[:k]).max(axis=1) > 1e-6) | (np.abs(gcost[:k] - ocost[:k]).max(axis=1) > 1e-6))
    print(" first differing node", int(bad[0]) if len(bad) else None, "of", k, "; dead flags differ at", np.flatnonzero(gdead[:k] != odead[:k])[:10].tolist())
    big = np.argsort(-np.abs(gcost[:k] - ocost[:k]).max(axis=1))[:5]
    for i in list(bad[:3]) + list(big):
        i = int(i)
        print("  node", i, "parent", gp[i], op[i], "pose", gpose[i].tolist(), opose[i].tolist(), "costs", gcost[i].tolist(), ocost[i].tolist(), "dead", gdead[i], odead[i])
