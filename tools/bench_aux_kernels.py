#!/usr/bin/env python3
"""Timings of the kernels around the hot path (VERDICT r02 'measure what round 2 built'): map authoring and GVD::Update in both modes,
post-processing of a 4096-query batch, Reeds-Shepp connect / IsPathValid at 1e6 paths.  Host-side wall times per call (the calls
synchronise); run it under `rocprofv3 --kernel-trace --stats -- python3 tools/bench_aux_kernels.py` for the per-kernel rows.
usage: bench_aux_kernels.py [--big]   (--big adds the 4096^2 / 384-outline map)"""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathplanning_amd as pa  # noqa: E402
from pathplanning_amd import synthetic  # noqa: E402

out = {}
ctx = pa.Context(0)


def timed(f, reps=1):
    f()  # warm (first dispatch, allocations)
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    return (time.perf_counter() - t0) / reps * 1e3, r


def build_map(cells, n_obstacles, seed=1, resolution=0.1):
    half = cells * resolution / 2.0
    lower, upper = np.array([-half, -half, -math.pi]), np.array([half, half, math.pi])
    rng = np.random.RandomState(seed)
    dx, dy = 0.3 * half / 2.0, 0.04 * half / 2.0
    corners = [(dx, dy), (-dx, dy), (-dx, -dy), (dx, -dy)]
    poses = [[*rng.uniform(-0.7 * half, 0.7 * half, 2), rng.uniform(-math.pi, math.pi)] for _ in range(n_obstacles)]

    def outlines():
        ms = pa.OccupancyMapSet.from_bounds(ctx, lower, upper, resolution)
        n = 0
        for k, p in enumerate(poses):
            n += ms.add_polygon(corners, p, k)
        return ms, n
    t_out, (ms, n_cells) = timed(outlines)
    t_edt, _ = timed(lambda: ms.update_gvd(mode=ms.GVD_EXACT_EDT), reps=5)
    t0 = time.perf_counter()
    pops = ms.update_gvd(mode=ms.GVD_REFERENCE_ORDER)
    t_ref = (time.perf_counter() - t0) * 1e3
    # incremental: one more outline, then its removal
    extra = [0.21 * half, -0.33 * half, 0.7]
    t0 = time.perf_counter()
    ms.add_polygon(corners, extra, n_obstacles)
    pops2 = ms.update_gvd(mode=ms.GVD_REFERENCE_ORDER)
    t_add = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    ms.add_polygon(corners, extra, -1)
    pops3 = ms.update_gvd(mode=ms.GVD_REFERENCE_ORDER)
    t_rm = (time.perf_counter() - t0) * 1e3
    n = cells * cells
    return ms, dict(cells=n, outlines=n_obstacles, outline_cells=n_cells, outlines_ms=t_out, gvd_exact_transform_ms=t_edt, exact_transform_GBs=n * 33.0 / (t_edt * 1e-3) / 1e9,
                    gvd_reference_order_first_build_ms=t_ref, brushfire_pops_first_build=pops, add_one_outline_ms=t_add, pops_add=pops2 - pops, remove_it_ms=t_rm, pops_remove=pops3 - pops2,
                    note="exact transform: 33 B per cell algorithmic (occupancy 4 + two labels 8 + two squared distances 8 + edge 1 + path cost 4 + distance 4 + scratch 4); wall time per call incl. its one synchronisation")


ms, out["map_1024_24_outlines"] = build_map(1024, 24)
print(json.dumps(out["map_1024_24_outlines"]), flush=True)
if "--big" in sys.argv:
    ms_big, out["map_4096_384_outlines"] = build_map(4096, 384)
    print(json.dumps(out["map_4096_384_outlines"]), flush=True)
    ms_big.close()

# ---- post-processing of a 4096-query batch on the 1024^2 map (fields in reference order)
val = pa.StateValidatorOccupancyMap(ms)
g = ms.download_gvd()
m = dict(lower=ms.lower, upper=ms.upper, resolution=0.1, occ=ms.download_occupancy(), d2=g["d2"], path_cost=g["path_cost"])
reach = synthetic.reachable_mask(val, m)
B = 4096
starts = synthetic.sample_valid_poses(val, m, B, seed=1000, reachable=reach)
goals = synthetic.sample_valid_poses(val, m, B, seed=2000, reachable=reach)
planner = pa.HybridAStarBatch(val, max_batch=B, max_nodes=81920, search_rows=2048)
planner.initialize()
t0 = time.perf_counter()
res = planner.search_batch(starts, goals, np.arange(B, dtype=np.uint64))
t_search = (time.perf_counter() - t0) * 1e3
for spacing in (0.8, 0.1):
    t_post, post = timed(lambda: planner.postprocess(path_interpolation=spacing))
    pts = sum(p.n_points for p in post)
    its = sum(p.iterations for p in post)
    out["postprocess_4096_queries_spacing_%.1f" % spacing] = dict(ms=t_post, sampled_points=pts, smoother_iterations=its, point_iterations_per_s=float(sum(p.n_points * p.iterations for p in post)) / (t_post * 1e-3),
                                                               search_ms_same_batch=t_search)
    print(json.dumps(out["postprocess_4096_queries_spacing_%.1f" % spacing]), flush=True)
planner.close()

# ---- Reeds-Shepp connect + IsPathValid over 1e6 pose pairs
n = 1_000_000
rng = np.random.RandomState(7)
half = 51.2
a = np.column_stack([rng.uniform(-half, half, n), rng.uniform(-half, half, n), rng.uniform(-math.pi, math.pi, n)])
b = a + np.column_stack([rng.uniform(-15, 15, n), rng.uniform(-15, 15, n), rng.uniform(-math.pi, math.pi, n)])
rs = pa.ReedsSheppPaths(ctx, min_turning_radius=2.0)
t_conn, paths = timed(lambda: rs.connect(a, b))
t_valid, (valid, last) = timed(lambda: val.is_rs_path_valid(paths))
out["reeds_shepp_1e6"] = dict(connect_ms_incl_pcie=t_conn, connect_paths_per_s=n / (t_conn * 1e-3), is_path_valid_ms_incl_pcie=t_valid, path_checks_per_s=n / (t_valid * 1e-3), valid_fraction=float(np.mean(valid)),
                              note="host-buffer entry points: 48 B in + 128 B out (connect), 128 B in + 5 B out (validity) per path cross PCIe inside the timed call; kernel rows in the rocprofv3 summary")
print(json.dumps(out["reeds_shepp_1e6"]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "aux_kernels.json"), "w"), indent=1)
