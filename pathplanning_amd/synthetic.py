"""Synthetic input generator for bench.py / examples (host side, numpy + scipy).

Produces the three grids the hot path READS but does not build ("data the path reads"):
occupancy (int32, -1 free / id occupied), squared obstacle distance (int32, exact EDT) and a
Voronoi-field potential (float32, Dolgov form as in state_validator/gvd.cpp:266-283).  The map
layout follows SURVEY 8(d): K rectangle outlines of 0.3*half x 0.04*half at seeded poses within
+-0.7*half.  This is input synthesis, not the reference's incremental brushfire (out of scope,
SURVEY 8f #1): values are self-consistent but not claimed equal to GVD::Update's.
"""
import math

import numpy as np


def _raster_segment(occ, p0, p1, origin, res, ident):
    n = int(max(abs(p1[0] - p0[0]), abs(p1[1] - p0[1])) / (0.25 * res)) + 2
    t = np.linspace(0.0, 1.0, n)
    x = p0[0] + t * (p1[0] - p0[0])
    y = p0[1] + t * (p1[1] - p0[1])
    r = ((x - origin[0]) / res).astype(np.int64)
    c = ((y - origin[1]) / res).astype(np.int64)
    ok = (r >= 0) & (r < occ.shape[0]) & (c >= 0) & (c < occ.shape[1])
    occ[r[ok], c[ok]] = ident


def make_map(n_cells, n_obstacles, seed, resolution=0.1):
    """Returns dict(lower, upper, resolution, occ, d2, path_cost)."""
    from scipy import ndimage
    res = float(np.float32(resolution))
    half = n_cells * resolution / 2.0
    lower = np.array([-half, -half, -math.pi])
    upper = np.array([half, half, math.pi])
    origin = (-float(np.float32(2 * half)) / 2.0, -float(np.float32(2 * half)) / 2.0)
    occ = np.full((n_cells, n_cells), -1, dtype=np.int32)
    rng = np.random.RandomState(seed)
    for k in range(n_obstacles):
        cx, cy = rng.uniform(-0.7 * half, 0.7 * half, 2)
        th = rng.uniform(-math.pi, math.pi)
        dx, dy = 0.3 * half / 2.0, 0.04 * half / 2.0
        corners = [(dx, dy), (-dx, dy), (-dx, -dy), (dx, -dy)]
        c, s = math.cos(th), math.sin(th)
        pts = [(c * x - s * y + cx, s * x + c * y + cy) for x, y in corners]
        for i in range(4):
            _raster_segment(occ, pts[i], pts[(i + 1) % 4], origin, res, k)
    free = occ < 0
    if free.all():
        d2 = np.full(occ.shape, np.iinfo(np.int32).max, dtype=np.int32)
        path_cost = np.zeros(occ.shape, dtype=np.float32)
    else:
        dist, idx = ndimage.distance_transform_edt(free, return_indices=True)
        d2 = np.rint(dist * dist).astype(np.int64).astype(np.int32)
        # Voronoi edges: cells whose 4-neighbours are closest to a different obstacle
        owner = occ[idx[0], idx[1]]
        edge = np.zeros(occ.shape, dtype=bool)
        edge[:-1, :] |= owner[:-1, :] != owner[1:, :]
        edge[:, :-1] |= owner[:, :-1] != owner[:, 1:]
        edge &= free & (d2 > 1)
        if edge.any():
            vd = ndimage.distance_transform_edt(~edge).astype(np.float32) * np.float32(res)
        else:
            vd = np.full(occ.shape, np.float32(np.sqrt(float(np.iinfo(np.int32).max)) * res), dtype=np.float32)
        od = (np.sqrt(d2.astype(np.float64)) * res).astype(np.float32)
        alpha, dmax = np.float32(20.0), np.float32(30.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            pc = (alpha / (alpha + od)) * (vd / (od + vd)) * (((od - dmax) ** 2) / (dmax ** 2))
        pc = np.where((od >= dmax) | ~np.isfinite(pc), np.float32(0.0), pc).astype(np.float32)
        path_cost = pc
    return dict(lower=lower, upper=upper, resolution=resolution, occ=occ, d2=d2, path_cost=path_cost)


def make_map_product(ctx, n_cells, n_obstacles, seed, resolution=0.1, reference_order=True):
    """The same K rectangle outlines as make_map, but built THROUGH THE PRODUCT: vertices rotated / translated on the host
    (PolygonShape::GetVerticesPosition), outlines rasterised by the device's Bresenham in AddObstacle order (pp_rasterize_cells +
    pp_map_set_cells), GVD::Update by pp_map_update_gvd_ex -- reference_order=True: the reference's own brushfire, its grids bit for
    bit (host replay, seconds); False: the exact Euclidean transform on the device (milliseconds).  Returns (m, info): m = the dict
    make_map returns (grids downloaded from the device), info = build times in ms."""
    import time
    from .planner import OccupancyMapSet
    half = n_cells * resolution / 2.0
    lower = np.array([-half, -half, -math.pi])
    upper = np.array([half, half, math.pi])
    ms = OccupancyMapSet.from_bounds(ctx, lower, upper, resolution)
    assert (ms.rows, ms.cols) == (n_cells, n_cells)
    rng = np.random.RandomState(seed)
    dx, dy = 0.3 * half / 2.0, 0.04 * half / 2.0
    corners = [(dx, dy), (-dx, dy), (-dx, -dy), (dx, -dy)]  # RectangleShape, obstacle.cpp:105-110
    t0 = time.perf_counter()
    for k in range(n_obstacles):
        cx, cy = rng.uniform(-0.7 * half, 0.7 * half, 2)
        th = rng.uniform(-math.pi, math.pi)
        ms.add_polygon(corners, [cx, cy, th], k)
    t1 = time.perf_counter()
    ms.update_gvd(mode=ms.GVD_EXACT_EDT)  # timed for the line; its grids are replaced below when reference_order
    t2 = time.perf_counter()
    if reference_order:
        ms.update_gvd(mode=ms.GVD_REFERENCE_ORDER)
    t3 = time.perf_counter()
    g = ms.download_gvd()
    m = dict(lower=lower, upper=upper, resolution=resolution, occ=ms.download_occupancy(), d2=g["d2"], path_cost=g["path_cost"])
    ms.close()
    info = dict(outlines_ms=(t1 - t0) * 1e3, gvd_exact_transform_ms=(t2 - t1) * 1e3, gvd_reference_order_ms=(t3 - t2) * 1e3 if reference_order else None,
                fields="reference order (host brushfire replay)" if reference_order else "exact transform (device)")
    return m, info


def upload(ctx, m):
    """Creates the device map set + validator from make_map()'s dict."""
    from .planner import OccupancyMapSet, StateValidatorOccupancyMap
    ms = OccupancyMapSet.from_bounds(ctx, m["lower"], m["upper"], m["resolution"])
    assert (ms.rows, ms.cols) == m["occ"].shape, ((ms.rows, ms.cols), m["occ"].shape)
    ms.upload_dist2(m["d2"])
    ms.upload_occupancy(m["occ"])
    ms.upload_path_cost(m["path_cost"])
    return ms, StateValidatorOccupancyMap(ms)


def sample_valid_poses(validator, m, n, seed, reachable=None):
    """Uniform poses inside the bounds that the (GPU) validator accepts.  `reachable`: optional bool
    grid; poses whose cell is not set are rejected (e.g. the pockets enclosed by outline obstacles)."""
    rng = np.random.RandomState(seed)
    out = np.empty((0, 3))
    lo, up = m["lower"], m["upper"]
    ms = validator.map
    res = float(ms.resolution)
    while len(out) < n:
        k = 2 * (n - len(out)) + 64
        p = np.column_stack([rng.uniform(lo[0], up[0], k), rng.uniform(lo[1], up[1], k), rng.uniform(-math.pi, math.pi, k)])
        ok = validator.is_state_valid(p)
        if reachable is not None:
            r = np.clip(((p[:, 0] - ms.grid_origin[0]) / res).astype(np.int64), 0, ms.rows - 1)
            c = np.clip(((p[:, 1] - ms.grid_origin[1]) / res).astype(np.int64), 0, ms.cols - 1)
            ok &= reachable[r, c]
        out = np.concatenate([out, p[ok]])
    return np.ascontiguousarray(out[:n])


def reachable_mask(validator, m, seed=0):
    """Cells connected to the bulk of the free space FOR THE ROBOT: one obstacle-heuristic wavefront
    on the GPU over an occupancy inflated by the validator's safety radius (cells closer than
    minSafeRadius to an obstacle count as occupied).  Used only to pick benchmark queries that have a
    solution; pockets enclosed by outline obstacles or by narrow gaps are dropped."""
    from .planner import ObstaclesHeuristic, OccupancyMapSet
    ms = validator.map
    dist = np.sqrt(m["d2"].astype(np.float64)) * float(ms.resolution)
    inflated = np.where(dist < validator.min_safe_radius, 0, -1).astype(np.int32)
    tmp = OccupancyMapSet(ms.ctx, ms.lower, ms.upper, ms.resolution, ms.rows, ms.cols, ms.grid_origin)
    tmp.upload_occupancy(inflated)
    p = sample_valid_poses(validator, m, 64, seed)
    fields = ObstaclesHeuristic(tmp).update(p[:4, :2])
    tmp.close()
    best = max(fields, key=lambda f: np.isfinite(f).sum())  # the largest component among a few seeds
    return np.isfinite(best)
