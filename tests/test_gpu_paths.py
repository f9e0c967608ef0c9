"""-m gpu: the path value types of the plugin surface through the C ABI against the oracle (SURVEY 8a row a7, and the
IsPathValid overloads of row a2 that round 1 could only reach from inside a search):
PathConnectionReedsShepp::Connect, PathReedsShepp::Interpolate / GetDirection / Truncate (with the reference's Q11 slot reset) /
GetCuspPointRatios, IsPathValid over Reeds-Shepp and PathSE2 paths, and the float distance-grid upload."""
import math

import numpy as np
import pytest

import oracle_lib as O
from gpu_common import make_pair, valid_random_poses

pytestmark = pytest.mark.gpu


def random_pairs(rng, n, span=12.0):
    a = np.column_stack([rng.uniform(-span, span, n), rng.uniform(-span, span, n), rng.uniform(-math.pi, math.pi, n)])
    b = np.column_stack([rng.uniform(-span, span, n), rng.uniform(-span, span, n), rng.uniform(-math.pi, math.pi, n)])
    return a, b


def same_records(got, want, tol=1e-9):
    assert np.array_equal(got["direction"], want["direction"])
    fin = np.isfinite(want["motion_length"])
    assert np.array_equal(np.isfinite(got["motion_length"]), fin)
    used = fin & (want["direction"] != 2)  # Motion::steer of an unused slot is indeterminate in the reference (reeds_shepp.h:55)
    assert np.array_equal(got["steer"][used], want["steer"][used])
    assert np.abs(got["motion_length"][fin] - want["motion_length"][fin]).max() < tol
    assert np.abs(got["final_pose"] - want["final_pose"]).max() < tol
    assert np.abs(got["length"] - want["length"]).max() < tol
    assert np.array_equal(got["min_turning_radius"], want["min_turning_radius"])


def test_connect_matches_oracle():
    import pathplanning_amd as pa
    ctx = pa.Context(0)
    rng = np.random.RandomState(7)
    a, b = random_pairs(rng, 20000)
    a[:4] = b[:4]  # zero-length paths: PathSegment of length 0 -> m_final = m_init
    for rmin, rev, fwd, sw in ((2.0, 1.0, 1.0, 0.0), (1.3, 2.0, 1.0, 0.3)):
        got = pa.ReedsSheppPaths(ctx, rmin, sw, rev, fwd).connect(a, b)
        want = O.rs_connect(a, b, rmin, rev, fwd, sw)
        same = got["word"] == want["word"]
        assert same.mean() > 0.999  # device libm vs glibc: near-ties between two words of equal cost (float compare)
        same_records(got[same], want[same])
        assert np.abs(got["cost"][same] - want["cost"][same]).max() < 1e-4
        assert np.abs(got["cost"] - want["cost"]).max() < 1e-3  # a different word of (nearly) the same cost


def test_interpolate_direction_truncate_cusps_match_oracle():
    import pathplanning_amd as pa
    ctx = pa.Context(0)
    rng = np.random.RandomState(8)
    a, b = random_pairs(rng, 6000)
    rs = pa.ReedsSheppPaths(ctx, 2.0, 0.2, 1.5, 1.0)
    P = O.rs_connect(a, b, 2.0, 1.5, 1.0, 0.2)  # identical inputs for both sides
    for ratios in (rng.uniform(0, 1, len(P)), np.zeros(len(P)), np.ones(len(P))):
        gp, gd = rs.interpolate(P, ratios)
        wp, wd = O.rs_path_interpolate(P, ratios)
        assert np.abs(gp - wp).max() < 1e-9
        assert np.array_equal(gd, wd)
    cut = rng.uniform(0.02, 0.98, len(P))
    T = O.rs_path_truncate(P, cut)
    same_records(rs.truncate(P, cut), T, tol=1e-9)
    # the Q11 slot reset is visible: Interpolate(1) of a truncated path no longer reaches its own m_final
    gp, gd = rs.interpolate(T, 1.0)
    wp, wd = O.rs_path_interpolate(T, 1.0)
    assert np.abs(gp - wp).max() < 1e-9 and np.array_equal(gd, wd)
    assert np.abs(wp[:, :2] - T["final_pose"][:, :2]).max() > 0.1
    fixed = rs.truncate(P, cut, q11=False)
    fp, _ = rs.interpolate(fixed, 1.0)
    assert np.abs(fp - fixed["final_pose"]).max() < 1e-9  # with the intended behaviour it does
    # truncating twice, cusp ratios of whole and truncated paths
    T2 = O.rs_path_truncate(T, 0.5)
    same_records(rs.truncate(T, 0.5), T2)
    for recs in (P, T, T2):
        g, w = rs.get_cusp_point_ratios(recs), O.rs_path_cusps(recs)
        assert [len(x) for x in g] == [len(x) for x in w]
        assert all(np.abs(x - y).max() < 1e-12 for x, y in zip(g, w) if len(x))
    assert max(len(x) for x in O.rs_path_cusps(P)) >= 2


def test_is_path_valid_over_reeds_shepp_and_se2_paths():
    import pathplanning_amd as pa
    w, ms, val, ctx = make_pair(256, 6, 3)
    rng = np.random.RandomState(9)
    n = 4000
    a, b = valid_random_poses(rng, w, n), valid_random_poses(rng, w, n)
    b[: n // 2, :2] = a[: n // 2, :2] + rng.uniform(-4, 4, (n // 2, 2))  # short connections: many stay collision free
    a[:3] = b[:3]
    P = O.rs_connect(a, b, 2.0)
    P = np.concatenate([P, O.rs_path_truncate(P[:1000], rng.uniform(0.1, 0.9, 1000))])
    gv, gl = val.is_rs_path_valid(P)
    wv, wl = O.rs_paths_valid(w, P)
    assert np.array_equal(gv, wv) and np.array_equal(gl, wl)
    assert 0.05 < gv.mean() < 0.95
    gv, gl = val.is_se2_path_valid(a, b)
    wv, wl = O.se2_paths_valid(w, a, b)
    assert np.array_equal(gv, wv) and np.array_equal(gl, wl)
    assert 0.05 < gv.mean() < 0.95
    # other validator tunables
    val.min_safe_radius = 0.4
    val.min_path_interpolation_distance = 0.03
    w.set_validator(0.4, 0.03)
    gv, gl = val.is_rs_path_valid(P[:1500])
    wv, wl = O.rs_paths_valid(w, P[:1500])
    assert np.array_equal(gv, wv) and np.array_equal(gl, wl)


def test_float_distance_grid_upload_equals_the_int_grid_upload():
    """pp_map_upload_distance takes what GetDistanceToNearestObstacle returns (float metres, gvd.h:38); no lossy round trip
    through d2 = lround(d * d): exact for INT_MAX cells (obstacle-free map) and for d2 > 2^24 (4096^2 maps)."""
    import pathplanning_amd as pa
    w, ms, val, ctx = make_pair(256, 6, 3)
    rng = np.random.RandomState(10)
    poses = np.column_stack([rng.uniform(-13.5, 13.5, 50000), rng.uniform(-13.5, 13.5, 50000), rng.uniform(-3.3, 3.3, 50000)])
    want = w.is_state_valid(poses).astype(bool)
    assert np.array_equal(val.is_state_valid(poses), want)
    d2 = w.d2()
    res = np.float64(np.float32(0.1))
    dist = (np.sqrt(d2.astype(np.float64)) * res).astype(np.float32)  # gvd.h:38: float(std::sqrt(int) * resolution), the product in double
    ms2 = pa.OccupancyMapSet.from_bounds(ctx, w.lb, w.ub, 0.1)
    ms2.upload_distance(dist)
    ms2.upload_occupancy(w.occ())
    ms2.upload_path_cost(w.pathcost())
    val2 = pa.StateValidatorOccupancyMap(ms2)
    assert np.array_equal(val2.is_state_valid(poses), want)
    a, b = valid_random_poses(rng, w, 500), valid_random_poses(rng, w, 500)
    P = O.rs_connect(a, b, 2.0)
    gv, gl = val2.is_rs_path_valid(P)
    wv, wl = O.rs_paths_valid(w, P)
    assert np.array_equal(gv, wv) and np.array_equal(gl, wl)
    # obstacle-free map: every d2 is INT_MAX -> distance 4634.1 m; everything inside the bounds is valid
    free = O.World(12.8, 12.8, 0.1)
    free.update()
    assert (free.d2() == 2**31 - 1).all()
    ms3 = pa.OccupancyMapSet.from_bounds(ctx, free.lb, free.ub, 0.1)
    ms3.upload_distance(np.full((free.rows, free.cols), np.float32(np.sqrt(np.float64(2**31 - 1)) * res), np.float32))
    val3 = pa.StateValidatorOccupancyMap(ms3)
    assert np.array_equal(val3.is_state_valid(poses), free.is_state_valid(poses).astype(bool))
    assert val3.is_state_valid(poses).sum() > 40000
