"""ctypes binding of the CPU oracle (oracle/libppo.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import math
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_u8p = C.POINTER(C.c_uint8)
_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)


def _build():
    so = os.path.join(ORACLE_DIR, "libppo.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("ppo_capi.cpp", "ppo_geometry.hpp", "ppo_world.hpp", "ppo_search.hpp", "ppo_post.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, os.path.join(ORACLE_DIR, "libppo.so")])
    return so


def dptr(a):
    return a.ctypes.data_as(_dp)


def fptr(a):
    return a.ctypes.data_as(_fp)


def iptr(a):
    return a.ctypes.data_as(_ip)


def u8ptr(a):
    return a.ctypes.data_as(_u8p)


def arr3(p):
    return np.ascontiguousarray(np.asarray(p, dtype=np.float64).reshape(-1))


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_build())
        L.ppo_world_create.restype = C.c_void_p
        L.ppo_world_create.argtypes = [_dp, _dp, C.c_float]
        L.ppo_hybrid_create.restype = C.c_void_p
        L.ppo_grid_astar.restype = C.c_void_p
        L.ppo_rrt.restype = C.c_void_p
        L.ppo_rs_shortest.restype = C.c_double
        L.ppo_hybrid_batch.restype = C.c_double
        L.ppo_world_distance_value.restype = C.c_float
        L.ppo_steering_from_radius.restype = C.c_double
        L.ppo_curvature_from_steering.restype = C.c_double
        _lib = L
    return _lib


DEFAULT_PARAMS = dict(wheelbase=2.6, min_turning_radius=2.0, direction_switching_cost=0.0, reverse_cost_multiplier=1.0,
                      forward_cost_multiplier=1.0, voronoi_cost_multiplier=1.0, num_generated_motion=5,
                      spatial_resolution=1.0, angular_resolution=0.0872)


def params_array(**kw):
    p = dict(DEFAULT_PARAMS)
    p.update(kw)
    return np.array([p["wheelbase"], p["min_turning_radius"], p["direction_switching_cost"], p["reverse_cost_multiplier"],
                     p["forward_cost_multiplier"], p["voronoi_cost_multiplier"], p["num_generated_motion"],
                     p["spatial_resolution"], p["angular_resolution"]], dtype=np.float64)


class World:
    """The reference's StateSpaceSE2 + ObstacleListOccupancyMap + StateValidatorOccupancyMap + GVD."""

    def __init__(self, half_x, half_y=None, resolution=0.1):
        half_y = half_x if half_y is None else half_y
        self.lb = np.array([-half_x, -half_y, -math.pi])
        self.ub = np.array([half_x, half_y, math.pi])
        self.resolution = np.float32(resolution)
        self.h = C.c_void_p(lib().ppo_world_create(dptr(self.lb), dptr(self.ub), C.c_float(resolution)))
        assert self.h
        r, c = C.c_int(), C.c_int()
        lib().ppo_world_dims(self.h, C.byref(r), C.byref(c))
        self.rows, self.cols = r.value, c.value
        o = np.zeros(2)
        lib().ppo_world_origin(self.h, dptr(o))
        self.origin = o
        self.min_safe_radius = 1.0
        self.min_interp = 0.1

    def __del__(self):
        try:
            lib().ppo_world_destroy(self.h)
        except Exception:
            pass

    def set_validator(self, min_safe_radius=1.0, min_interp=0.1):
        self.min_safe_radius, self.min_interp = min_safe_radius, min_interp
        lib().ppo_world_set_validator(self.h, C.c_float(min_safe_radius), C.c_float(min_interp))

    def add_rectangle(self, dx, dy, pose):
        return lib().ppo_world_add_rectangle(self.h, C.c_double(dx), C.c_double(dy), dptr(arr3(pose)))

    def remove_rectangle(self, ident, dx, dy, pose):
        """ObstacleListOccupancyMap::RemoveObstacle (obstacle_list_occupancy_map.cpp:46-61) of a rectangle added with add_rectangle"""
        return lib().ppo_world_remove_rectangle(self.h, C.c_int(int(ident)), C.c_double(dx), C.c_double(dy), dptr(arr3(pose)))

    def add_circle(self, radius, count, pose):
        return lib().ppo_world_add_circle(self.h, C.c_double(radius), C.c_int(count), dptr(arr3(pose)))

    def update(self):
        lib().ppo_world_update(self.h)

    def occ(self):
        a = np.empty((self.rows, self.cols), dtype=np.int32)
        lib().ppo_world_get_occ(self.h, iptr(a))
        return a

    def d2(self):
        a = np.empty((self.rows, self.cols), dtype=np.int32)
        lib().ppo_world_get_d2(self.h, iptr(a))
        return a

    def voro_d2(self):
        a = np.empty((self.rows, self.cols), dtype=np.int32)
        lib().ppo_world_get_voro_d2(self.h, iptr(a))
        return a

    def pathcost(self):
        a = np.empty((self.rows, self.cols), dtype=np.float32)
        lib().ppo_world_get_pathcost(self.h, fptr(a))
        return a

    def set_occ(self, a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        assert a.shape == (self.rows, self.cols)
        lib().ppo_world_set_occ(self.h, iptr(a))

    def set_d2(self, a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        assert a.shape == (self.rows, self.cols)
        lib().ppo_world_set_d2(self.h, iptr(a))

    def set_pathcost(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.shape == (self.rows, self.cols)
        lib().ppo_world_set_pathcost(self.h, fptr(a))

    def is_state_valid(self, poses):
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 3)
        out = np.empty(len(poses), dtype=np.uint8)
        lib().ppo_is_state_valid(self.h, C.c_int64(len(poses)), dptr(poses), u8ptr(out))
        return out

    def to_cell(self, xy, bounded=True):
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        out = np.empty((len(xy), 2), dtype=np.int32)
        lib().ppo_world_to_cell(self.h, C.c_int64(len(xy)), dptr(xy), C.c_int(1 if bounded else 0), iptr(out))
        return out

    def is_path_valid_csteer(self, frm, steering, length, direction, wheelbase=2.6):
        frm = np.ascontiguousarray(frm, dtype=np.float64).reshape(-1, 3)
        n = len(frm)
        steering = np.ascontiguousarray(np.broadcast_to(steering, n), dtype=np.float64)
        length = np.ascontiguousarray(np.broadcast_to(length, n), dtype=np.float64)
        direction = np.ascontiguousarray(np.broadcast_to(direction, n), dtype=np.int32)
        valid = np.empty(n, dtype=np.uint8)
        last = np.empty(n, dtype=np.float32)
        lib().ppo_is_path_valid_csteer(self.h, C.c_int64(n), dptr(frm), dptr(steering), dptr(length), iptr(direction),
                                       C.c_double(wheelbase), u8ptr(valid), fptr(last))
        return valid, last

    def is_path_valid_r2(self, frm, to):
        frm = np.ascontiguousarray(frm, dtype=np.float64).reshape(-1, 2)
        to = np.ascontiguousarray(to, dtype=np.float64).reshape(-1, 2)
        valid = np.empty(len(frm), dtype=np.uint8)
        lib().ppo_is_path_valid_r2(self.h, C.c_int64(len(frm)), dptr(frm), dptr(to), u8ptr(valid))
        return valid

    def obstacle_heuristic(self, goal_xy, rev=1.0, fwd=1.0, literal=False):
        cost = np.empty((self.rows, self.cols), dtype=np.float32)
        explored = np.empty((self.rows, self.cols), dtype=np.uint8)
        g = np.ascontiguousarray(goal_xy, dtype=np.float64)
        lib().ppo_obstacle_heuristic(self.h, dptr(g), C.c_double(rev), C.c_double(fwd), C.c_int(1 if literal else 0), fptr(cost), u8ptr(explored))
        return cost, explored

    def counters(self):
        a, b = C.c_uint64(), C.c_uint64()
        lib().ppo_world_counters(self.h, C.byref(a), C.byref(b))
        return a.value, b.value


def constant_steer(frm, steering, dist, direction, wheelbase=2.6):
    frm = np.ascontiguousarray(frm, dtype=np.float64).reshape(-1, 3)
    n = len(frm)
    steering = np.ascontiguousarray(np.broadcast_to(steering, n), dtype=np.float64)
    dist = np.ascontiguousarray(np.broadcast_to(dist, n), dtype=np.float64)
    direction = np.ascontiguousarray(np.broadcast_to(direction, n), dtype=np.int32)
    out = np.empty((n, 3))
    lib().ppo_constant_steer(C.c_int64(n), dptr(frm), dptr(steering), dptr(dist), iptr(direction), C.c_double(wheelbase), dptr(out))
    return out


def rs_shortest(start, goal, rmin=1.0):
    w = C.c_int(-1)
    tuv = np.zeros(3)
    length = lib().ppo_rs_shortest(dptr(arr3(start)), dptr(arr3(goal)), C.c_double(rmin), C.byref(w), dptr(tuv))
    return w.value, tuv, length


def rs_shortest_path(start, goal, rmin=1.0):
    w, nm, sl = C.c_int(-1), C.c_int(0), C.c_double(0)
    steer = np.zeros(5, dtype=np.int32)
    direc = np.zeros(5, dtype=np.int32)
    length = np.zeros(5)
    final = np.zeros(3)
    lib().ppo_rs_shortest_path(dptr(arr3(start)), dptr(arr3(goal)), C.c_double(rmin), C.byref(w), C.byref(nm), iptr(steer), iptr(direc), dptr(length),
                               C.byref(sl), dptr(final))
    return dict(word=w.value, nmotions=nm.value, steer=steer, direction=direc, length=length, seg_length=sl.value, final=final)


def rs_optimal_batch(starts, goals, rmin, rev=1.0, fwd=1.0, sw=0.0):
    starts = np.ascontiguousarray(starts, dtype=np.float64).reshape(-1, 3)
    goals = np.ascontiguousarray(goals, dtype=np.float64).reshape(-1, 3)
    n = len(starts)
    word = np.empty(n, dtype=np.int32)
    tuv = np.empty((n, 3))
    cost = np.empty(n, dtype=np.float32)
    seglen = np.empty(n)
    lib().ppo_rs_optimal_batch(C.c_int64(n), dptr(starts), dptr(goals), C.c_double(rmin), C.c_float(rev), C.c_float(fwd), C.c_float(sw), iptr(word),
                               dptr(tuv), fptr(cost), dptr(seglen))
    return word, tuv, cost, seglen


def rs_interpolate(start, word, tuv, rmin, ratios):
    ratios = np.ascontiguousarray(ratios, dtype=np.float64)
    out = np.empty((len(ratios), 3))
    lib().ppo_rs_interpolate(dptr(arr3(start)), C.c_int(word), dptr(np.ascontiguousarray(tuv, dtype=np.float64)), C.c_double(rmin), C.c_int64(len(ratios)),
                             dptr(ratios), dptr(out))
    return out


RS_PATH_DTYPE = np.dtype([("start", "<f8", 3), ("final_pose", "<f8", 3), ("motion_length", "<f8", 5), ("steer", "i1", 5), ("direction", "i1", 5),
                          ("reserved", "i1", 6), ("min_turning_radius", "<f8"), ("length", "<f8"), ("cost", "<f4"), ("word", "<i4")])


def _vp(a):
    return C.c_void_p(a.ctypes.data)


def rs_connect(starts, goals, rmin, rev=1.0, fwd=1.0, sw=0.0):
    """PathConnectionReedsShepp::Connect -> records (PathReedsShepp as data)."""
    a = np.ascontiguousarray(starts, dtype=np.float64).reshape(-1, 3)
    b = np.ascontiguousarray(goals, dtype=np.float64).reshape(-1, 3)
    out = np.zeros(len(a), dtype=RS_PATH_DTYPE)
    lib().ppo_rs_connect(C.c_int64(len(a)), dptr(a), dptr(b), C.c_double(rmin), C.c_float(rev), C.c_float(fwd), C.c_float(sw), _vp(out))
    return out


def rs_path_interpolate(paths, ratios):
    p = np.ascontiguousarray(paths, dtype=RS_PATH_DTYPE).reshape(-1)
    r = np.ascontiguousarray(np.broadcast_to(np.asarray(ratios, dtype=np.float64), len(p)))
    pose = np.empty((len(p), 3))
    direction = np.empty(len(p), dtype=np.int32)
    lib().ppo_rs_path_interpolate(C.c_int64(len(p)), _vp(p), dptr(r), dptr(pose), iptr(direction))
    return pose, direction


def rs_path_truncate(paths, ratios):
    p = np.ascontiguousarray(paths, dtype=RS_PATH_DTYPE).reshape(-1).copy()
    r = np.ascontiguousarray(np.broadcast_to(np.asarray(ratios, dtype=np.float64), len(p)))
    lib().ppo_rs_path_truncate(C.c_int64(len(p)), _vp(p), dptr(r))
    return p


def rs_path_cusps(paths):
    p = np.ascontiguousarray(paths, dtype=RS_PATH_DTYPE).reshape(-1)
    ratios = np.empty((len(p), 4))
    count = np.empty(len(p), dtype=np.int32)
    lib().ppo_rs_path_cusps(C.c_int64(len(p)), _vp(p), dptr(ratios), iptr(count))
    return [ratios[i, :count[i]].copy() for i in range(len(p))]


def rs_paths_valid(world, paths):
    p = np.ascontiguousarray(paths, dtype=RS_PATH_DTYPE).reshape(-1)
    valid = np.empty(len(p), dtype=np.uint8)
    last = np.empty(len(p), dtype=np.float32)
    lib().ppo_rs_paths_valid(world.h, C.c_int64(len(p)), _vp(p), u8ptr(valid), fptr(last))
    return valid.astype(bool), last


def se2_paths_valid(world, starts, goals):
    a = np.ascontiguousarray(starts, dtype=np.float64).reshape(-1, 3)
    b = np.ascontiguousarray(goals, dtype=np.float64).reshape(-1, 3)
    valid = np.empty(len(a), dtype=np.uint8)
    last = np.empty(len(a), dtype=np.float32)
    lib().ppo_se2_paths_valid(world.h, C.c_int64(len(a)), dptr(a), dptr(b), u8ptr(valid), fptr(last))
    return valid.astype(bool), last


def nonholo_build(lb, ub, params=None, threads=8):
    params = params_array() if params is None else params
    dims = np.zeros(3, dtype=np.int32)
    offs = np.zeros(2)
    lib().ppo_nonholo_build(dptr(arr3(lb)), dptr(arr3(ub)), dptr(params), iptr(dims), dptr(offs), None, C.c_int(1))
    table = np.empty(tuple(int(d) for d in dims))
    lib().ppo_nonholo_build(dptr(arr3(lb)), dptr(arr3(ub)), dptr(params), iptr(dims), dptr(offs), dptr(table), C.c_int(threads))
    return table, offs


class Hybrid:
    def __init__(self, world, params=None, heading_alias=True, negative_k_read=True, table=None):
        self.world = world
        self.params = params_array() if params is None else params
        self.h = C.c_void_p(lib().ppo_hybrid_create(world.h, dptr(self.params), C.c_int(int(heading_alias)), C.c_int(int(negative_k_read))))
        if table is None:
            table, _ = nonholo_build(world.lb, world.ub, self.params)
        self.table = np.ascontiguousarray(table, dtype=np.float64)
        lib().ppo_hybrid_initialize(self.h, dptr(self.table))
        self.P = lib().ppo_hybrid_num_primitives(self.h)

    def __del__(self):
        try:
            lib().ppo_hybrid_destroy(self.h)
        except Exception:
            pass

    def deltas(self):
        d = np.zeros(self.P // 2)
        lib().ppo_hybrid_deltas(self.h, dptr(d))
        return d

    def set_goal(self, goal):
        lib().ppo_hybrid_set_goal(self.h, dptr(arr3(goal)))

    def heuristic(self, poses):
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 3)
        n = len(poses)
        a, b, c = np.empty(n), np.empty(n), np.empty(n)
        lib().ppo_hybrid_heuristic(self.h, C.c_int64(n), dptr(poses), dptr(a), dptr(b), dptr(c))
        return a, b, c

    def discretize(self, poses):
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 3)
        out = np.empty((len(poses), 3), dtype=np.int32)
        lib().ppo_hybrid_discretize(self.h, C.c_int64(len(poses)), dptr(poses), iptr(out))
        return out

    def children(self, parents):
        parents = np.ascontiguousarray(parents, dtype=np.float64).reshape(-1, 3)
        n, P = len(parents), self.P
        valid = np.empty((n, P), dtype=np.uint8)
        poses = np.empty((n, P, 3))
        keys = np.empty((n, P, 3), dtype=np.int32)
        cost = np.empty((n, P))
        length = np.empty((n, P))
        lib().ppo_hybrid_children(self.h, C.c_int64(n), dptr(parents), u8ptr(valid), dptr(poses), iptr(keys), dptr(cost), dptr(length))
        return dict(valid=valid, poses=poses, keys=keys, cost=cost, length=length)

    def set_deltas(self, deltas):
        """explicit steering-angle list instead of the generated one (P = 2 * len(deltas))"""
        d = np.ascontiguousarray(deltas, dtype=np.float64)
        lib().ppo_hybrid_set_deltas(self.h, C.c_int(len(d)), dptr(d))
        self.P = 2 * len(d)

    def set_max_expansions(self, n):
        lib().ppo_hybrid_set_max_expansions(self.h, C.c_int64(n))

    def search(self, start, goal, seed=0):
        lib().ppo_hybrid_search.argtypes = [C.c_void_p, _dp, _dp, C.c_uint64]
        status = lib().ppo_hybrid_search(self.h, dptr(arr3(start)), dptr(arr3(goal)), C.c_uint64(seed))
        info = np.zeros(10, dtype=np.int64)
        cost = C.c_double()
        lib().ppo_hybrid_result_info(self.h, info.ctypes.data_as(_i64p), C.byref(cost))
        ne, npth = int(info[1]), int(info[2])
        expanded = np.empty((ne, 3), dtype=np.int32)
        if ne:
            lib().ppo_hybrid_result_expanded(self.h, iptr(expanded))
        poses = np.empty((npth, 3))
        kind = np.empty(npth, dtype=np.int32)
        steering = np.empty(npth)
        length = np.empty(npth)
        direction = np.empty(npth, dtype=np.int32)
        rsword = np.empty(npth, dtype=np.int32)
        pcost = np.empty(npth)
        if npth:
            lib().ppo_hybrid_result_path(self.h, dptr(poses), iptr(kind), dptr(steering), dptr(length), iptr(direction), iptr(rsword), dptr(pcost))
        return dict(status=status, cost=cost.value, expanded=expanded, path_poses=poses, path_kind=kind, path_steering=steering,
                    path_length=length, path_direction=direction, path_rsword=rsword, path_cost=pcost, n_nodes=int(info[3]),
                    n_state_checks=int(info[4]), n_path_checks=int(info[5]), n_rng_draws=int(info[6]), n_rs_attempts=int(info[7]),
                    n_children=int(info[8]), n_lattice_boundary_hits=int(info[9]))


SMOOTHER_DEFAULTS = dict(step_tolerance=1e-3, max_iterations=2000, learning_rate=0.01, path_weight=0.0, smooth_weight=0.4, voronoi_weight=0.02,
                         collision_weight=0.2, curvature_weight=0.4, collision_ratio=0.2, max_curvature=0.5)


def smoother_array(**kw):
    p = dict(SMOOTHER_DEFAULTS)
    p.update(kw)
    return np.array([p[k] for k in ("step_tolerance", "max_iterations", "learning_rate", "path_weight", "smooth_weight", "voronoi_weight", "collision_weight",
                                    "curvature_weight", "collision_ratio", "max_curvature")], dtype=np.float32)


def postprocess(world, result, goal, params=None, path_interpolation=0.1, smoother=None, nearest=None):
    """hybrid_a_star.cpp:260-304 on a search result of Hybrid.search: resampled path, cusp flags, smoothing status, smoothed path.
    nearest = (nearest_obstacle[rows, cols, 2], nearest_edge[rows, cols, 2]) to use label grids other than the world's brushfire."""
    p = params_array() if params is None else params
    hp = np.array([p[0], p[1], p[3], p[4], p[2]], dtype=np.float64)  # wheelbase, rmin, reverse, forward, switching
    sp = smoother_array(max_curvature=1.0 / p[1]) if smoother is None else smoother
    n = len(result["path_poses"])
    poses = np.ascontiguousarray(result["path_poses"], dtype=np.float64)
    kind = np.ascontiguousarray(result["path_kind"], dtype=np.int32)
    steering = np.ascontiguousarray(result["path_steering"], dtype=np.float64)
    length = np.ascontiguousarray(result["path_length"], dtype=np.float64)
    direction = np.ascontiguousarray(result["path_direction"], dtype=np.int32)
    no = ne = None
    if nearest is not None:
        no = np.ascontiguousarray(nearest[0], dtype=np.int32)
        ne = np.ascontiguousarray(nearest[1], dtype=np.int32)
    L = lib()
    L.ppo_postprocess.restype = C.c_void_p
    h = C.c_void_p(L.ppo_postprocess(world.h, dptr(hp), C.c_int(n), dptr(poses), iptr(kind), dptr(steering), dptr(length), iptr(direction), dptr(arr3(goal)),
                                     C.c_float(path_interpolation), fptr(sp), iptr(no) if no is not None else None, iptr(ne) if ne is not None else None))
    npts, status, iters, plen = C.c_int(), C.c_int(), C.c_int(), C.c_double()
    L.ppo_post_info(h, C.byref(npts), C.byref(status), C.byref(iters), C.byref(plen))
    k = npts.value
    resampled, smoothed, ratios = np.empty((k, 3)), np.empty((k, 3)), np.empty(k)
    cusp = np.empty(k, dtype=np.uint8)
    L.ppo_post_get(h, dptr(resampled), u8ptr(cusp), dptr(smoothed), dptr(ratios))
    L.ppo_post_destroy(h)
    return dict(n_points=k, status=status.value, iterations=iters.value, length=plen.value, resampled=resampled, cusp=cusp.astype(bool), smoothed=smoothed, ratios=ratios)


def smoother_libm_last_bit(shift):
    """Sensitivity probe (ppo_post.hpp: Smoother::LibmLastBit): +1 / -1 moves every cosine of the curvature term one ulp, 0 restores."""
    lib().ppo_smoother_libm_last_bit(C.c_int(int(shift)))


def world_nearest(world):
    no = np.empty((world.rows, world.cols, 2), dtype=np.int32)
    ne = np.empty((world.rows, world.cols, 2), dtype=np.int32)
    lib().ppo_world_get_nearest(world.h, iptr(no), iptr(ne))
    return no, ne


def hybrid_batch(world, table, starts, goals, seeds, threads=1, params=None, heading_alias=True, negative_k_read=True):
    params = params_array() if params is None else params
    starts = np.ascontiguousarray(starts, dtype=np.float64).reshape(-1, 3)
    goals = np.ascontiguousarray(goals, dtype=np.float64).reshape(-1, 3)
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    n = len(starts)
    status = np.empty(n, dtype=np.int32)
    cost = np.empty(n)
    nexp = np.empty(n, dtype=np.int64)
    table = np.ascontiguousarray(table, dtype=np.float64)
    secs = lib().ppo_hybrid_batch(world.h, dptr(params), C.c_int(int(heading_alias)), C.c_int(int(negative_k_read)), dptr(table), C.c_int64(n),
                                  dptr(starts), dptr(goals), seeds.ctypes.data_as(_u64p), C.c_int(threads), iptr(status), dptr(cost),
                                  nexp.ctypes.data_as(_i64p))
    return secs, status, cost, nexp


def hybrid_batch_paths(world, table, starts, goals, seeds, threads=1, max_poses=256, params=None, heading_alias=True, negative_k_read=True):
    """hybrid_batch plus the solution paths: returns (seconds, status, cost, n_expanded, poses [n, max_poses, 3], n_poses [n])"""
    params = params_array() if params is None else params
    starts = np.ascontiguousarray(starts, dtype=np.float64).reshape(-1, 3)
    goals = np.ascontiguousarray(goals, dtype=np.float64).reshape(-1, 3)
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    n = len(starts)
    status = np.empty(n, dtype=np.int32)
    cost = np.empty(n)
    nexp = np.empty(n, dtype=np.int64)
    poses = np.zeros((n, max_poses, 3))
    n_poses = np.zeros(n, dtype=np.int32)
    table = np.ascontiguousarray(table, dtype=np.float64)
    lib().ppo_hybrid_batch_paths.restype = C.c_double
    secs = lib().ppo_hybrid_batch_paths(world.h, dptr(params), C.c_int(int(heading_alias)), C.c_int(int(negative_k_read)), dptr(table), C.c_int64(n),
                                        dptr(starts), dptr(goals), seeds.ctypes.data_as(_u64p), C.c_int(threads), iptr(status), dptr(cost),
                                        nexp.ctypes.data_as(_i64p), C.c_int(max_poses), dptr(poses), iptr(n_poses))
    return secs, status, cost, nexp, poses, n_poses


def grid_astar(world, init, goal, bidirectional=False, inner_goal_f=None, inner_goal_r=None):
    init = np.array(init, dtype=np.int32)
    goal = np.array(goal, dtype=np.int32)
    igf = np.array(inner_goal_f if inner_goal_f is not None else goal, dtype=np.int32)
    igr = np.array(inner_goal_r if inner_goal_r is not None else goal, dtype=np.int32)
    h = C.c_void_p(lib().ppo_grid_astar(world.h, iptr(init), iptr(goal), C.c_int(int(bidirectional)), iptr(igf), iptr(igr), None, None))
    st, cost, npth, ne, ner = C.c_int(), C.c_double(), C.c_int(), C.c_int(), C.c_int()
    lib().ppo_grid_result_info(h, C.byref(st), C.byref(cost), C.byref(npth), C.byref(ne), C.byref(ner))
    path = np.empty((npth.value, 2), dtype=np.int32)
    ex = np.empty((ne.value, 2), dtype=np.int32)
    exr = np.empty((ner.value, 2), dtype=np.int32)
    lib().ppo_grid_result_get(h, iptr(path), iptr(ex), iptr(exr))
    lib().ppo_grid_result_destroy(h)
    return dict(status=st.value, cost=cost.value, path=path, explored=ex, explored_reverse=exr)


def rrt(world, lb, ub, init, goal, seed, star=False, max_iteration=100, max_nodes=10000, max_connection=0.1, goal_bias=0.05, gamma=0.0):
    """star: False / 0 RRT, True / 1 RRT* as the reference, 2 = + rewire, 3 = + rewire with the radius near-set (gamma)"""
    lb = np.array(lb, dtype=np.float64)
    ub = np.array(ub, dtype=np.float64)
    params = np.array([max_iteration, max_nodes, max_connection, goal_bias, gamma], dtype=np.float64)
    init = np.array(init, dtype=np.float64)
    goal = np.array(goal, dtype=np.float64)
    lib().ppo_rrt.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, C.c_uint64, C.c_int]
    h = C.c_void_p(lib().ppo_rrt(world.h if world is not None else None, dptr(lb), dptr(ub), dptr(params), dptr(init), dptr(goal), C.c_uint64(seed),
                                 C.c_int(int(star))))
    info = np.zeros(6, dtype=np.int64)
    lib().ppo_rrt_result_info(h, info.ctypes.data_as(_i64p))
    nn, npth = int(info[1]), int(info[2])
    nodes = np.empty((nn, 2))
    parents = np.empty(nn, dtype=np.int32)
    costs = np.empty(nn)
    path = np.empty((npth, 2))
    lib().ppo_rrt_result_get(h, dptr(nodes), iptr(parents), dptr(costs), dptr(path))
    lib().ppo_rrt_result_destroy(h)
    return dict(status=int(info[0]), nodes=nodes, parents=parents, costs=costs, path=path, iterations=int(info[3]), n_knn=int(info[4]),
                n_edge_checks=int(info[5]))


def tree_knn(points, query, k):
    """k nearest tree nodes of `query` through the oracle's PointTree (utils/tree.h:73-116 restated)."""
    points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 2)
    query = np.ascontiguousarray(query, dtype=np.float64)
    out = np.empty(max(k, 1), dtype=np.int32)
    n = lib().ppo_tree_knn(C.c_int(len(points)), dptr(points), dptr(query), C.c_int(k), iptr(out))
    return out[:n]


def frontier_replay(ops, costs, mode=0):
    ops = np.ascontiguousarray(ops, dtype=np.int32)
    costs = np.ascontiguousarray(costs, dtype=np.float64)
    popped = np.empty(len(ops), dtype=np.int32)
    n = lib().ppo_frontier_replay(C.c_int(mode), C.c_int(len(ops)), iptr(ops), dptr(costs), iptr(popped))
    return popped[:n]


def rng_uniform(seed, n, lb=0.0, ub=1.0):
    out = np.empty(n)
    lib().ppo_rng_uniform.argtypes = [C.c_uint64, C.c_int64, C.c_double, C.c_double, _dp]
    lib().ppo_rng_uniform(C.c_uint64(seed), C.c_int64(n), C.c_double(lb), C.c_double(ub), dptr(out))
    return out


def neighbors(row, col, rows, cols):
    n = C.c_int()
    rc = np.empty((8, 2), dtype=np.int32)
    lib().ppo_neighbors(C.c_int(row), C.c_int(col), C.c_int(rows), C.c_int(cols), C.byref(n), iptr(rc))
    return rc[:n.value]


def synthetic_world(n_cells, n_obstacles, seed, resolution=0.1):
    """SURVEY 8(d) map generator: K rectangle outlines (0.3*half x 0.04*half) at seeded poses within +-0.7*half."""
    half = n_cells * resolution / 2.0
    w = World(half, half, resolution)
    rng = np.random.RandomState(seed)
    for _ in range(n_obstacles):
        x, y = rng.uniform(-0.7 * half, 0.7 * half, 2)
        th = rng.uniform(-math.pi, math.pi)
        w.add_rectangle(0.3 * half, 0.04 * half, [x, y, th])
    w.update()
    return w
