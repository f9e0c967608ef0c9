"""Shared helpers for the -m gpu parity tests: an oracle World and the matching device map set."""
import math

import numpy as np

import oracle_lib as O


def make_pair(n_cells, n_obstacles, seed, resolution=0.1, ctx=None):
    """Returns (oracle world, device map set, validator) over identical grids."""
    import pathplanning_amd as pa
    w = O.synthetic_world(n_cells, n_obstacles, seed, resolution)
    ctx = ctx or pa.Context(0)
    ms = pa.OccupancyMapSet.from_bounds(ctx, w.lb, w.ub, resolution)
    assert (ms.rows, ms.cols) == (w.rows, w.cols)
    assert np.allclose(ms.grid_origin, w.origin, rtol=0, atol=0)
    ms.upload_dist2(w.d2())
    ms.upload_occupancy(w.occ())
    ms.upload_path_cost(w.pathcost())
    val = pa.StateValidatorOccupancyMap(ms)
    return w, ms, val, ctx


def random_poses(rng, w, n, margin=1.05):
    half = w.ub[0]
    p = np.empty((n, 3))
    p[:, 0] = rng.uniform(-margin * half, margin * half, n)
    p[:, 1] = rng.uniform(-margin * half, margin * half, n)
    p[:, 2] = rng.uniform(-1.2 * math.pi, 1.2 * math.pi, n)
    return p


def valid_random_poses(rng, w, n):
    out = []
    while len(out) < n:
        p = random_poses(rng, w, 4 * n, margin=0.98)
        p[:, 2] = rng.uniform(-math.pi, math.pi, len(p))
        ok = w.is_state_valid(p).astype(bool)
        out.extend(list(p[ok]))
    return np.array(out[:n])
