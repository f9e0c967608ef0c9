"""Host-side mirror of the reference's operator interface for the hot path, on top of the C ABI.

Class / method names follow the reference's pybind11 module (interfaces/python/src/pyplanning.cpp)
where a counterpart exists: StateValidatorOccupancyMap.is_state_valid / is_path_valid,
HybridAStar.set_init_state / set_goal_state / search_path / get_path, Status.  Batched entry
points are the MI355X-native addition.  Everything computes on the GPU through libpphip.so;
nothing here falls back to the CPU.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import RS_PATH_DTYPE, HybridParams, MapDesc, PostResult, QueryResult, SmootherParams, check, ptr


class Status:
    """algo/path_planner.h:9-12"""
    SUCCESS = 0
    FAILURE = -1


def _f64(a, cols):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return a.reshape(-1, cols)


def _is_tensor(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


def _dev_ptr(t):
    assert t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


class Context:
    """One device + one HIP stream.  `stream` may be a raw hipStream_t (int), e.g.
    torch.cuda.current_stream().cuda_stream, so that torch events see the work."""

    def __init__(self, device=0, stream=None):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.pp_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h
        self.device = device

    def synchronize(self):
        check(self.lib.pp_ctx_synchronize(self.h))

    def is_idle(self):
        """True when everything enqueued on this context's stream has completed (non-blocking)."""
        idle = C.c_int32(0)
        check(self.lib.pp_ctx_is_idle(self.h, C.byref(idle)))
        return bool(idle.value)

    def timer_start(self):
        check(self.lib.pp_ctx_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        check(self.lib.pp_ctx_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def close(self):
        if self.h:
            self.lib.pp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OccupancyMapSet:
    """Device-resident data of one OccupancyMap + GVD: squared-distance grid, occupancy, Voronoi
    potential, plus the StateSpaceSE2 bounds (state_validator/occupancy_map.h:122-133)."""

    def __init__(self, ctx, lower, upper, resolution, rows, cols, grid_origin, local_origin=(0.0, 0.0)):
        self.ctx = ctx
        self.lib = ctx.lib
        self.lower = np.asarray(lower, dtype=np.float64)
        self.upper = np.asarray(upper, dtype=np.float64)
        self.resolution = np.float32(resolution)
        self.rows, self.cols = int(rows), int(cols)
        self.grid_origin = np.asarray(grid_origin, dtype=np.float64)
        d = MapDesc()
        d.rows, d.cols, d.resolution = self.rows, self.cols, float(self.resolution)
        d.grid_origin[:] = list(self.grid_origin)
        d.local_origin[:] = list(local_origin)
        d.lower[:] = list(self.lower)
        d.upper[:] = list(self.upper)
        self.desc = d
        h = C.c_void_p()
        check(self.lib.pp_map_create(ctx.h, C.byref(d), C.byref(h)))
        self.h = h

    @classmethod
    def from_bounds(cls, ctx, lower, upper, resolution):
        """Sizes the grid like StateValidatorOccupancyMap's constructor + OccupancyMap::InitializeSize
        (state_validator_occupancy_map.cpp:6-13, occupancy_map.cpp:6-14): float width/height."""
        lower = np.asarray(lower, dtype=np.float64)
        upper = np.asarray(upper, dtype=np.float64)
        res = np.float32(resolution)
        width = np.float32(upper[0] - lower[0])
        height = np.float32(upper[1] - lower[1])
        origin = (-float(width) / 2.0, -float(height) / 2.0)
        rows = int(math.ceil(float(np.float32(width / res))))
        cols = int(math.ceil(float(np.float32(height / res))))
        return cls(ctx, lower, upper, res, rows, cols, origin)

    def upload_dist2(self, d2):
        d2 = np.ascontiguousarray(d2, dtype=np.int32)
        assert d2.shape == (self.rows, self.cols)
        check(self.lib.pp_map_upload_dist2(self.h, ptr(d2)))

    def upload_distance(self, dist):
        """float distance grid in metres, as ObstacleDistanceMap::GetDistanceToNearestObstacle returns it (gvd.h:38)"""
        dist = np.ascontiguousarray(dist, dtype=np.float32)
        assert dist.shape == (self.rows, self.cols)
        check(self.lib.pp_map_upload_distance(self.h, ptr(dist)))

    # ---- map authoring / field construction on the device ----
    def rasterize_segments(self, p0, p1, value):
        """Shape::RasterizeLine for each world segment p0[i] -> p1[i]; writes `value` (obstacle id, or -1 to remove).  Returns cells written."""
        a, b = _f64(p0, 2), _f64(p1, 2)
        n = C.c_int32(0)
        check(self.lib.pp_map_rasterize_segments(self.h, len(a), ptr(a), ptr(b), int(value), C.byref(n)))
        return n.value

    def add_polygon(self, vertices, pose, value):
        """PolygonShape::GetGridCellsPosition (obstacle.cpp:77-93): vertices rotated / translated by `pose` on the host, edges
        rasterised on the device."""
        v = _f64(vertices, 2)
        c, s_ = math.cos(pose[2]), math.sin(pose[2])
        w = np.column_stack([c * v[:, 0] - s_ * v[:, 1] + pose[0], s_ * v[:, 0] + c * v[:, 1] + pose[1]])
        return self.rasterize_segments(w, np.roll(w, -1, axis=0), value)

    def download_occupancy(self):
        occ = np.empty((self.rows, self.cols), dtype=np.int32)
        check(self.lib.pp_map_download_occupancy(self.h, ptr(occ)))
        return occ

    GVD_EXACT_EDT, GVD_REFERENCE_ORDER = 0, 1

    def update_gvd(self, alpha=20.0, d_max=30.0, mode=0):
        """GVD::Update from the device occupancy grid.  mode GVD_REFERENCE_ORDER: the reference's brushfire replayed over the ordered cell
        edits (grids bit-identical to the reference's, incremental after the first call; returns heap pops so far); GVD_EXACT_EDT
        (default): exact Euclidean transform on the device (returns the number of device passes)."""
        it = C.c_int32(0)
        check(self.lib.pp_map_update_gvd_ex(self.h, C.c_float(alpha), C.c_float(d_max), int(mode), C.byref(it)))
        return it.value

    def download_gvd(self):
        shape = (self.rows, self.cols)
        out = dict(d2=np.empty(shape, np.int32), nearest_obstacle=np.empty(shape + (2,), np.int32), voronoi_edge=np.empty(shape, np.uint8),
                   voronoi_d2=np.empty(shape, np.int32), nearest_edge=np.empty(shape + (2,), np.int32), path_cost=np.empty(shape, np.float32))
        check(self.lib.pp_map_download_gvd(self.h, ptr(out["d2"]), ptr(out["nearest_obstacle"]), ptr(out["voronoi_edge"]), ptr(out["voronoi_d2"]),
                                           ptr(out["nearest_edge"]), ptr(out["path_cost"])))
        return out

    def path_cost_update(self, obstacle_d2, voronoi_d2, alpha=20.0, d_max=30.0):
        """PathCostMap::Update (gvd.cpp:266-283) over two squared-distance grids; becomes the map's path cost."""
        a = np.ascontiguousarray(obstacle_d2, dtype=np.int32)
        b = np.ascontiguousarray(voronoi_d2, dtype=np.int32)
        out = np.empty((self.rows, self.cols), dtype=np.float32)
        check(self.lib.pp_path_cost_update(self.h, ptr(a), ptr(b), C.c_float(alpha), C.c_float(d_max), ptr(out)))
        return out

    def upload_nearest_cells(self, nearest_obstacle, nearest_edge):
        """(rows, cols, 2) int grids: GVD::GetNearestObstacleCell / GetNearestVoronoiEdgeCell per cell (the smoother reads them)"""
        a = np.ascontiguousarray(nearest_obstacle, dtype=np.int32)
        b = np.ascontiguousarray(nearest_edge, dtype=np.int32)
        assert a.shape == b.shape == (self.rows, self.cols, 2)
        check(self.lib.pp_map_upload_nearest_cells(self.h, ptr(a), ptr(b)))

    def upload_occupancy(self, occ):
        occ = np.ascontiguousarray(occ, dtype=np.int32)
        assert occ.shape == (self.rows, self.cols)
        check(self.lib.pp_map_upload_occupancy(self.h, ptr(occ)))

    def upload_path_cost(self, cost):
        cost = np.ascontiguousarray(cost, dtype=np.float32)
        assert cost.shape == (self.rows, self.cols)
        check(self.lib.pp_map_upload_path_cost(self.h, ptr(cost)))

    def download_distance(self):
        out = np.empty((self.rows, self.cols), dtype=np.float32)
        check(self.lib.pp_map_download_distance(self.h, ptr(out)))
        return out

    def close(self):
        if self.h:
            self.lib.pp_map_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StateValidatorOccupancyMap:
    """state_validator/state_validator_occupancy_map.{h,cpp}: is_state_valid / is_path_valid, batched."""

    def __init__(self, map_set):
        self.map = map_set
        self.lib = map_set.lib
        self._min_safe_radius = 1.0
        self._min_interp = 0.1
        self._push()

    def _push(self):
        check(self.lib.pp_map_set_validator(self.map.h, C.c_float(self._min_safe_radius), C.c_float(self._min_interp)))

    @property
    def min_safe_radius(self):
        return self._min_safe_radius

    @min_safe_radius.setter
    def min_safe_radius(self, v):
        self._min_safe_radius = float(v)
        self._push()

    @property
    def min_path_interpolation_distance(self):
        return self._min_interp

    @min_path_interpolation_distance.setter
    def min_path_interpolation_distance(self, v):
        self._min_interp = float(v)
        self._push()

    def get_occupancy_map(self):
        return self.map

    def is_state_valid(self, poses):
        """poses: (n, 3) array-like -> bool array; or a CUDA float64 tensor -> CUDA uint8 tensor."""
        if _is_tensor(poses):
            import torch
            n = poses.numel() // 3
            out = torch.empty(n, dtype=torch.uint8, device=poses.device)
            check(self.lib.pp_check_states_dev(self.map.h, n, _dev_ptr(poses), _dev_ptr(out)))
            return out
        p = _f64(poses, 3)
        out = np.empty(len(p), dtype=np.uint8)
        check(self.lib.pp_check_states(self.map.h, len(p), ptr(p), ptr(out)))
        return out.astype(bool)

    def is_path_valid(self, start, curvature, length, direction):
        """IsPathValid over constant-steer arcs.  Returns (valid, last_valid_ratio)."""
        s = _f64(start, 3)
        n = len(s)
        k = np.ascontiguousarray(np.broadcast_to(np.asarray(curvature, dtype=np.float64), n))
        ln = np.ascontiguousarray(np.broadcast_to(np.asarray(length, dtype=np.float64), n))
        d = np.ascontiguousarray(np.broadcast_to(np.asarray(direction, dtype=np.int32), n))
        valid = np.empty(n, dtype=np.uint8)
        last = np.empty(n, dtype=np.float32)
        check(self.lib.pp_check_arcs(self.map.h, n, ptr(s), ptr(k), ptr(ln), ptr(d), ptr(valid), ptr(last)))
        return valid.astype(bool), last

    def is_segment_valid(self, start_xy, end_xy):
        a = _f64(start_xy, 2)
        b = _f64(end_xy, 2)
        valid = np.empty(len(a), dtype=np.uint8)
        check(self.lib.pp_check_segments(self.map.h, len(a), ptr(a), ptr(b), ptr(valid)))
        return valid.astype(bool)

    def is_rs_path_valid(self, paths):
        """IsPathValid over Reeds-Shepp paths (records of RS_PATH_DTYPE, e.g. from ReedsSheppPaths.connect).  Returns (valid, last)."""
        p = np.ascontiguousarray(paths, dtype=RS_PATH_DTYPE).reshape(-1)
        valid = np.empty(len(p), dtype=np.uint8)
        last = np.empty(len(p), dtype=np.float32)
        check(self.lib.pp_check_rs_paths(self.map.h, len(p), ptr(p), ptr(valid), ptr(last)))
        return valid.astype(bool), last

    def is_se2_path_valid(self, start, end):
        """IsPathValid over PathSE2 (paths/path_se2.cpp: linear in position and heading).  Returns (valid, last)."""
        a, b = _f64(start, 3), _f64(end, 3)
        valid = np.empty(len(a), dtype=np.uint8)
        last = np.empty(len(a), dtype=np.float32)
        check(self.lib.pp_check_se2_paths(self.map.h, len(a), ptr(a), ptr(b), ptr(valid), ptr(last)))
        return valid.astype(bool), last

    def count_valid_fused(self, n, seed, count_tensor):
        check(self.lib.pp_check_states_fused_dev(self.map.h, int(n), C.c_uint64(seed), _dev_ptr(count_tensor)))


class HybridAStarSearchParameters:
    """HybridAStar::SearchParameters, algo/hybrid_a_star.h:29-50 (same 8-argument constructor order as
    pyplanning.cpp:73-75) plus the two reference-behaviour switches."""

    def __init__(self, min_turning_radius=2.0, direction_switching_cost=0.0, reverse_cost_multiplier=1.0, forward_cost_multiplier=1.0,
                 voronoi_cost_multiplier=1.0, num_generated_motion=5, spatial_resolution=1.0, angular_resolution=0.0872, wheelbase=2.6,
                 heading_alias=True, negative_k_read=True):
        self.wheelbase = wheelbase
        self.min_turning_radius = min_turning_radius
        self.direction_switching_cost = direction_switching_cost
        self.reverse_cost_multiplier = reverse_cost_multiplier
        self.forward_cost_multiplier = forward_cost_multiplier
        self.voronoi_cost_multiplier = voronoi_cost_multiplier
        self.num_generated_motion = num_generated_motion
        self.spatial_resolution = spatial_resolution
        self.angular_resolution = angular_resolution
        self.heading_alias = heading_alias
        self.negative_k_read = negative_k_read

    def to_c(self):
        return HybridParams(self.wheelbase, self.min_turning_radius, self.direction_switching_cost, self.reverse_cost_multiplier,
                            self.forward_cost_multiplier, self.voronoi_cost_multiplier, int(self.num_generated_motion), self.spatial_resolution,
                            self.angular_resolution, int(bool(self.heading_alias)), int(bool(self.negative_k_read)))

    def primitives(self):
        """Steering angles / curvatures / directions in the reference's child order
        (hybrid_a_star.cpp:21-28,65-77; kinematic_bicycle_model.cpp:13-17,34-41)."""
        delta_max = math.atan(self.wheelbase / math.sqrt(math.pow(self.min_turning_radius, 2) - math.pow(0.0, 2)))
        deltas = [0.0]
        for i in range(int(self.num_generated_motion) // 2):
            d = (i + 1) / 2.0 * delta_max
            deltas += [d, -d]
        curv, direc, steer = [], [], []
        for d in deltas:
            tan_s = math.tan(d)
            beta = math.atan(0.0 * tan_s / self.wheelbase)
            k = math.cos(beta) * tan_s / self.wheelbase
            for dr in (0, 1):
                curv.append(k)
                direc.append(dr)
                steer.append(d)
        return np.array(steer), np.array(curv), np.array(direc, dtype=np.int32)


class ReedsSheppSolver:
    """ReedsShepp::Solver::GetOptimalPath, batched (geometry/reeds_shepp.cpp:654-683)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.lib = ctx.lib

    def get_optimal_path(self, start, goal, min_turning_radius, reverse_cost=1.0, forward_cost=1.0, switch_cost=0.0):
        a, b = _f64(start, 3), _f64(goal, 3)
        n = len(a)
        word = np.empty(n, dtype=np.int32)
        tuv = np.empty((n, 3))
        cost = np.empty(n, dtype=np.float32)
        seg = np.empty(n)
        check(self.lib.pp_rs_solve(self.ctx.h, n, ptr(a), ptr(b), C.c_double(min_turning_radius), C.c_float(reverse_cost), C.c_float(forward_cost),
                                   C.c_float(switch_cost), ptr(word), ptr(tuv), ptr(cost), ptr(seg)))
        return word, tuv, cost, seg


class ReedsSheppPaths:
    """PathReedsShepp / PathConnectionReedsShepp as batches of 128-byte records (paths/path_reeds_shepp.{h,cpp}): connect,
    interpolate (+ get_direction), truncate (with the reference's Q11 slot reset), get_cusp_point_ratios -- all on the device."""

    def __init__(self, ctx, min_turning_radius=1.0, direction_switching_cost=0.0, reverse_cost_multiplier=1.0, forward_cost_multiplier=1.0):
        self.ctx, self.lib = ctx, ctx.lib
        self.rmin, self.sw, self.rev, self.fwd = min_turning_radius, direction_switching_cost, reverse_cost_multiplier, forward_cost_multiplier

    def connect(self, start, goal):
        a, b = _f64(start, 3), _f64(goal, 3)
        out = np.zeros(len(a), dtype=RS_PATH_DTYPE)
        check(self.lib.pp_rs_connect(self.ctx.h, len(a), ptr(a), ptr(b), C.c_double(self.rmin), C.c_float(self.rev), C.c_float(self.fwd), C.c_float(self.sw), ptr(out)))
        return out

    def interpolate(self, paths, ratios):
        p = np.ascontiguousarray(paths, dtype=RS_PATH_DTYPE).reshape(-1)
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(ratios, dtype=np.float64), len(p)))
        pose = np.empty((len(p), 3))
        direction = np.empty(len(p), dtype=np.int32)
        check(self.lib.pp_rs_path_interpolate(self.ctx.h, len(p), ptr(p), ptr(r), ptr(pose), ptr(direction)))
        return pose, direction

    def truncate(self, paths, ratios, q11=True):
        p = np.ascontiguousarray(paths, dtype=RS_PATH_DTYPE).reshape(-1).copy()
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(ratios, dtype=np.float64), len(p)))
        check(self.lib.pp_rs_path_truncate(self.ctx.h, len(p), ptr(p), ptr(r), int(bool(q11))))
        return p

    def get_cusp_point_ratios(self, paths):
        p = np.ascontiguousarray(paths, dtype=RS_PATH_DTYPE).reshape(-1)
        ratios = np.empty((len(p), 4))
        count = np.empty(len(p), dtype=np.int32)
        check(self.lib.pp_rs_path_cusps(self.ctx.h, len(p), ptr(p), ptr(ratios), ptr(count)))
        return [ratios[i, :count[i]].copy() for i in range(len(p))]


class NonHolonomicHeuristic:
    """NonHolonomicHeuristic::Build (algo/heuristics.cpp:36-76) on the device."""

    @staticmethod
    def build(ctx, lower, upper, params):
        lo = np.ascontiguousarray(lower, dtype=np.float64)
        up = np.ascontiguousarray(upper, dtype=np.float64)
        cp = params.to_c()
        dims = np.zeros(3, dtype=np.int32)
        offs = np.zeros(2)
        check(ctx.lib.pp_nonholo_dims(ptr(lo), ptr(up), C.byref(cp), ptr(dims), ptr(offs)))
        table = np.empty(tuple(int(x) for x in dims))
        check(ctx.lib.pp_nonholo_build(ctx.h, ptr(lo), ptr(up), C.byref(cp), ptr(table)))
        return table, offs


class ObstaclesHeuristic:
    """ObstaclesHeuristic::Update (algo/heuristics.cpp:106-153): one exact-order wavefront per goal."""

    def __init__(self, map_set):
        self.map = map_set
        self.lib = map_set.lib

    def update(self, goals_xy):
        g = _f64(goals_xy, 2)
        out = np.empty((len(g), self.map.rows, self.map.cols), dtype=np.float32)
        check(self.lib.pp_obstacle_heuristic(self.map.h, len(g), ptr(g), ptr(out)))
        return out

    def update_dev(self, goals_xy, cost_tensor):
        g = _f64(goals_xy, 2)
        check(self.lib.pp_obstacle_heuristic_dev(self.map.h, len(g), ptr(g), _dev_ptr(cost_tensor)))

    TILE_STATS = ("goals", "tile_visits", "rounds", "candidate_passes", "cells", "handed_over", "wave_cycles", "tiles_per_goal",
                  "cycles_load", "cycles_masks", "cycles_passes", "cycles_requeue", "cycles_store")

    def update_dev_tile_stats(self, goals_xy, cost_tensor):
        """The fields into cost_tensor like update_dev, through the tile form of the wavefront with its work counters:
        returns (dict of TILE_STATS, launch milliseconds)."""
        g = _f64(goals_xy, 2)
        st = np.zeros(16, dtype=np.uint64)
        ms = C.c_float(0.0)
        handed = np.full(len(g), -1, dtype=np.int32)
        check(self.lib.pp_obstacle_heuristic_tiles_stats(self.map.h, len(g), ptr(g), _dev_ptr(cost_tensor), ptr(st), C.byref(ms), ptr(handed)))
        d = dict(zip(self.TILE_STATS, (int(x) for x in st)))
        d["handed_over_goals"] = sorted(int(x) for x in handed[:d["handed_over"]])
        return d, float(ms.value)


class Tree:
    """Tree::GetNearestNodes (utils/tree.h:73-116): exact kNN, squared L2, ascending."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.lib = ctx.lib

    def get_nearest_nodes(self, points_xy, queries_xy, k=1):
        p, q = _f64(points_xy, 2), _f64(queries_xy, 2)
        idx = np.empty((len(q), k), dtype=np.int32)
        d2 = np.empty((len(q), k))
        check(self.lib.pp_knn(self.ctx.h, len(p), ptr(p), len(q), ptr(q), int(k), ptr(idx), ptr(d2)))
        return idx, d2


class HybridAStarBatch:
    """HybridAStar graph search for batches of independent (start, goal, seed) queries.

    Single-query use mirrors PathPlannerSE2Base (pyplanning.cpp:66-71):
        planner.set_init_state(pose); planner.set_goal_state(pose); planner.search_path(); planner.get_path()
    """

    def __init__(self, validator, params=None, max_batch=1, max_nodes=16384, search_rows=0):
        """search_rows: rows (= node/heap/key-map buffer sets) of the four-queries-per-wave search kernel that
        throughput-sized planners (max_batch > 64) use; 0 = as many as can be resident.  Pass resident rows / k when k
        planners share one GPU."""
        self.validator = validator
        self.map = validator.map
        self.lib = self.map.lib
        self.params = params if params is not None else HybridAStarSearchParameters()
        self.cparams = self.params.to_c()
        self.max_batch, self.max_nodes = int(max_batch), int(max_nodes)
        h = C.c_void_p()
        check(self.lib.pp_planner_create_ex(self.map.h, C.byref(self.cparams), self.max_batch, self.max_nodes, int(search_rows), C.byref(h)))
        self.h = h
        self.num_primitives = self.lib.pp_planner_num_primitives(self.h)
        self.search_rows = self.lib.pp_planner_search_rows(self.h)  # 0: one-query-per-wave kernel
        self._init = np.zeros(3)
        self._goal = np.zeros(3)
        self._seed = 0
        self._results = None
        self.is_initialized = False

    def set_primitives(self, steering_angles):
        """Explicit steering-angle list instead of the one hybrid_a_star.cpp:21-28 generates (P = 2 * len: forward + backward each)."""
        d = np.ascontiguousarray(steering_angles, dtype=np.float64)
        check(self.lib.pp_planner_set_primitives(self.h, len(d), ptr(d)))
        self.num_primitives = self.lib.pp_planner_num_primitives(self.h)

    def initialize(self, nonholo_table=None):
        """HybridAStar::Initialize (hybrid_a_star.cpp:206-235): builds the non-holonomic table on the
        device unless one is supplied."""
        if nonholo_table is not None:
            t = np.ascontiguousarray(nonholo_table, dtype=np.float64)
            check(self.lib.pp_planner_set_nonholo_table(self.h, ptr(t)))
        else:
            check(self.lib.pp_planner_set_nonholo_table(self.h, None))
        self.is_initialized = True
        return True

    def nonholo_table(self):
        cp = self.cparams
        dims = np.zeros(3, dtype=np.int32)
        offs = np.zeros(2)
        lo = np.ascontiguousarray(self.map.lower)
        up = np.ascontiguousarray(self.map.upper)
        check(self.lib.pp_nonholo_dims(ptr(lo), ptr(up), C.byref(cp), ptr(dims), ptr(offs)))
        t = np.empty(tuple(int(x) for x in dims))
        check(self.lib.pp_planner_get_nonholo_table(self.h, ptr(t)))
        return t

    # -- batch API --------------------------------------------------------
    def search_batch(self, starts, goals, seeds):
        if not self.is_initialized:
            # hybrid_a_star.cpp:243-246: "The algorithm has not been initialized successfully."
            return None
        s, g = _f64(starts, 3), _f64(goals, 3)
        sd = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64))
        n = len(s)
        res = (QueryResult * n)()
        check(self.lib.pp_planner_search_batch(self.h, n, ptr(s), ptr(g), ptr(sd), C.cast(res, C.c_void_p)))
        self._results = res
        self._n = n
        return res

    def start_after_fields_of(self, predecessor):
        """the next batch of this planner starts on the device once `predecessor`'s current batch has its heuristic fields (one-shot):
        phases the batches in flight one wavefront apart instead of letting them run in step (scheduling only)"""
        check(self.lib.pp_planner_start_after_fields_of(self.h, predecessor.h))

    def search_batch_dev(self, starts_t, goals_t, seeds_t):
        n = starts_t.numel() // 3
        check(self.lib.pp_planner_search_batch_dev(self.h, n, _dev_ptr(starts_t), _dev_ptr(goals_t), _dev_ptr(seeds_t)))
        self._n = n

    def fetch_results(self, n=None):
        n = self._n if n is None else n
        res = (QueryResult * n)()
        check(self.lib.pp_planner_fetch_results(self.h, n, C.cast(res, C.c_void_p)))
        self._results = res
        return res

    def last_timings(self):
        a, b = C.c_float(), C.c_float()
        check(self.lib.pp_planner_last_timings(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def get_path_of(self, q):
        r = self._results[q]
        n = r.n_path
        poses = np.empty((n, 3))
        kind = np.empty(n, dtype=np.int32)
        prim = np.empty(n, dtype=np.int32)
        length = np.empty(n)
        tuv = np.empty((n, 3))
        if n:
            check(self.lib.pp_planner_get_path(self.h, q, ptr(poses), ptr(kind), ptr(prim), ptr(length), ptr(tuv)))
        return dict(poses=poses, kind=kind, prim=prim, length=length, tuv=tuv)

    def postprocess(self, n_queries=None, path_interpolation=0.1, smoother=None, max_points=2048):
        """HybridAStar::SearchPath's post-processing (hybrid_a_star.cpp:260-304) for the first n queries of the last batch: composite
        path, sampling every `path_interpolation` metres with cusp snapping, Smoother::Smooth.  `smoother`: dict of
        Smoother::Parameters fields to override (defaults: smoother.h:28-60, max_curvature = 1 / min_turning_radius).
        Returns one PostResult per query (n_points, smoothing_status, iterations, length)."""
        n = len(self._results) if n_queries is None else int(n_queries)
        sp = None
        if smoother is not None:
            d = dict(step_tolerance=1e-3, max_iterations=2000, learning_rate=0.01, path_weight=0.0, smooth_weight=0.4, voronoi_weight=0.02, collision_weight=0.2,
                     curvature_weight=0.4, collision_ratio=0.2, max_curvature=1.0 / self.params.min_turning_radius)
            d.update(smoother)
            sp = SmootherParams(**d)
        out = (PostResult * max(n, 1))()
        check(self.lib.pp_planner_postprocess(self.h, n, C.c_float(path_interpolation), C.byref(sp) if sp is not None else None, int(max_points), out))
        self._post = list(out)[:n]
        return self._post

    def get_processed_path(self, q):
        """sampled path, cusp flags and smoothed path of query q; `path` = what HybridAStar::GetPath() returns (the smoothed path
        when smoothing succeeded, else the sampled one, hybrid_a_star.cpp:293-303)"""
        r = self._post[q]
        n = r.n_points
        sampled, smoothed = np.empty((n, 3)), np.empty((n, 3))
        cusp = np.empty(n, dtype=np.uint8)
        if n:
            check(self.lib.pp_planner_get_processed_path(self.h, q, ptr(sampled), ptr(cusp), ptr(smoothed)))
        return dict(sampled=sampled, cusp=cusp.astype(bool), smoothed=smoothed, status=r.smoothing_status, iterations=r.iterations, length=r.length,
                    path=smoothed if r.smoothing_status >= 0 else sampled)

    def certify_lattice(self, q):
        """SURVEY 7.3 H2 as a contract (pp_planner_certify_lattice): created nodes and logged lattice-line children of query q recomputed on
        the host with glibc; returns (checked, cell mismatches, unverified events, largest pose difference).  mismatches == unverified == 0
        certifies the query's discrete outputs.  One-query-per-wave planners only."""
        a, b, c, d = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_double(0)
        check(self.lib.pp_planner_certify_lattice(self.h, int(q), C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    def get_expanded_of(self, q):
        r = self._results[q]
        cells = np.empty((r.n_expanded, 3), dtype=np.int32)
        if r.n_expanded:
            check(self.lib.pp_planner_get_expanded(self.h, q, ptr(cells)))
        return cells

    # -- PathPlannerSE2Base-shaped single query ---------------------------
    def set_init_state(self, pose):
        self._init = np.asarray(pose, dtype=np.float64)

    def set_goal_state(self, pose):
        self._goal = np.asarray(pose, dtype=np.float64)

    def set_seed(self, seed):
        self._seed = int(seed)

    def search_path(self):
        res = self.search_batch([self._init], [self._goal], [self._seed])
        if res is None:
            return Status.FAILURE
        return Status.SUCCESS if res[0].status == 0 else Status.FAILURE

    def get_path(self):
        """Graph-search path nodes (GetPath of the A* layer); post-processing/smoothing is out of scope."""
        if self._results is None:
            return np.empty((0, 3))
        return self.get_path_of(0)["poses"]

    def get_graph_search_optimal_cost(self):
        return self._results[0].cost if self._results is not None else float("inf")

    def close(self):
        if self.h:
            self.lib.pp_planner_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HybridAStarPipeline:
    """Streaming form of the batched planner (include/pp_hip.h: pp_pipeline_*): queries are submitted as they come, every goal's
    obstacle-heuristic field is handed by the wavefront kernel to ONE persistent search grid through a device-side queue, results are
    polled in completion order and their field slots recycled.  Per-query results are those of HybridAStarBatch.search_batch."""

    def __init__(self, validator, params=None, capacity=4096, max_nodes=81920, search_rows=0, log_expansions=False):
        self.validator = validator
        self.map = validator.map
        self.lib = self.map.lib
        self.params = params if params is not None else HybridAStarSearchParameters()
        self.cparams = self.params.to_c()
        h = C.c_void_p()
        check(self.lib.pp_pipeline_create(self.map.h, C.byref(self.cparams), int(capacity), int(max_nodes), int(search_rows), int(bool(log_expansions)), C.byref(h)))
        self.h = h
        self.planner_h = C.c_void_p(self.lib.pp_pipeline_planner(self.h))
        self.capacity = self.lib.pp_pipeline_capacity(self.h)
        self.search_rows = self.lib.pp_pipeline_search_rows(self.h)
        self.num_primitives = self.lib.pp_planner_num_primitives(self.planner_h)
        self._held = {}  # ticket -> QueryResult of completed queries polled with release=False

    def initialize(self, nonholo_table=None):
        t = None if nonholo_table is None else np.ascontiguousarray(nonholo_table, dtype=np.float64)
        check(self.lib.pp_planner_set_nonholo_table(self.planner_h, ptr(t) if t is not None else None))
        return True

    def nonholo_table(self):
        cp = self.cparams
        dims = np.zeros(3, dtype=np.int32)
        offs = np.zeros(2)
        lo, up = np.ascontiguousarray(self.map.lower), np.ascontiguousarray(self.map.upper)
        check(self.lib.pp_nonholo_dims(ptr(lo), ptr(up), C.byref(cp), ptr(dims), ptr(offs)))
        t = np.empty(tuple(int(x) for x in dims))
        check(self.lib.pp_planner_get_nonholo_table(self.planner_h, ptr(t)))
        return t

    def submit(self, starts, goals, seeds):
        """Returns the tickets of the queries taken (a prefix of the input: as many as there were free slots)."""
        s, g = _f64(starts, 3), _f64(goals, 3)
        sd = np.ascontiguousarray(seeds, dtype=np.uint64)
        n = len(s)
        tickets = np.empty(n, dtype=np.uint64)
        k = C.c_int32(0)
        check(self.lib.pp_pipeline_submit(self.h, n, ptr(s), ptr(g), ptr(sd), ptr(tickets), C.byref(k)))
        return tickets[:k.value]

    def submit_dev(self, starts_t, goals_t, seeds_t, n=None, offset=0):
        """device tensors ([n, 3] f64, [n, 3] f64, [n] int64 / uint64); returns (first ticket, number taken)"""
        n = starts_t.numel() // 3 - offset if n is None else n
        tickets = np.empty(max(n, 1), dtype=np.uint64)
        k = C.c_int32(0)
        check(self.lib.pp_pipeline_submit_dev(self.h, n, C.c_void_p(starts_t.data_ptr() + 24 * offset), C.c_void_p(goals_t.data_ptr() + 24 * offset),
                                              C.c_void_p(seeds_t.data_ptr() + 8 * offset), ptr(tickets), C.byref(k)))
        return (int(tickets[0]) if k.value else -1), k.value

    def poll(self, max_results=4096, release=True):
        """Completed queries so far: (tickets [k] uint64, results [k] QueryResult); never blocks."""
        tickets = np.empty(max_results, dtype=np.uint64)
        res = (QueryResult * max_results)()
        k = C.c_int32(0)
        check(self.lib.pp_pipeline_poll(self.h, int(max_results), ptr(tickets), C.cast(res, C.c_void_p), int(bool(release)), C.byref(k)))
        if not release:
            for i in range(k.value):
                self._held[int(tickets[i])] = res[i]
        return tickets[:k.value], res

    def poll_array(self, max_results=4096):
        """poll(release=True) as numpy arrays: (tickets [k] uint64, results [k] of _lib.QUERY_RESULT_DTYPE)"""
        from ._lib import QUERY_RESULT_DTYPE
        tickets = np.empty(max_results, dtype=np.uint64)
        res = np.empty(max_results, dtype=QUERY_RESULT_DTYPE)
        k = C.c_int32(0)
        check(self.lib.pp_pipeline_poll(self.h, int(max_results), ptr(tickets), ptr(res), 1, C.byref(k)))
        return tickets[:k.value], res[:k.value]

    def poll_array_held(self, max_results=4096):
        """poll(release=False) as numpy arrays: the slots stay held until get_paths(..., release=True) / release(); for callers that fetch
        every plan (bench.py)"""
        from ._lib import QUERY_RESULT_DTYPE
        tickets = np.empty(max_results, dtype=np.uint64)
        res = np.empty(max_results, dtype=QUERY_RESULT_DTYPE)
        k = C.c_int32(0)
        check(self.lib.pp_pipeline_poll(self.h, int(max_results), ptr(tickets), ptr(res), 0, C.byref(k)))
        return tickets[:k.value], res[:k.value]

    def get_paths(self, tickets, max_poses=256, release=True, out=None, n_out=None):
        """GetGraphSearchPath of completed, held queries, start pose first (pp_pipeline_get_paths): returns (poses [k, max_poses, 3] f64,
        n_poses [k] int32); rows beyond n_poses[i] are unspecified.  The poses arrive in pinned host memory with the completion records:
        this is a host copy."""
        t = np.ascontiguousarray(tickets, dtype=np.uint64)
        k = len(t)
        poses = out if out is not None else np.empty((max(k, 1), max_poses, 3))
        n_poses = n_out if n_out is not None else np.zeros(max(k, 1), dtype=np.int32)
        assert poses.flags.c_contiguous and poses.shape[0] >= k and poses.shape[1] == max_poses
        check(self.lib.pp_pipeline_get_paths(self.h, k, ptr(t), int(max_poses), ptr(poses), ptr(n_poses), int(bool(release))))
        if release:
            for x in t:
                self._held.pop(int(x), None)
        return poses[:k], n_poses[:k]

    def backlog(self):
        """(ready, searching): queries whose field is built and waiting for a row / claimed by a row, as of the last poll"""
        a, b = C.c_int64(0), C.c_int64(0)
        check(self.lib.pp_pipeline_backlog(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def timings(self):
        """kernel launch durations since the last call (pp_pipeline_timings), as a dict"""
        a, d, f = C.c_double(), C.c_double(), C.c_double()
        b, c, e = C.c_int64(), C.c_int64(), C.c_int64()
        check(self.lib.pp_pipeline_timings(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e), C.byref(f)))
        return dict(wavefront_ms_total=a.value, wavefront_launches=b.value, wavefront_goals=c.value, search_ms_total=d.value, search_launches=e.value, search_max_ms=f.value)

    def in_flight(self):
        return self.lib.pp_pipeline_in_flight(self.h)

    def alive_waves(self):
        """diagnostics: waves of the search grid that own their index right now (blocking copy)"""
        return self.lib.pp_pipeline_alive_waves(self.h)

    def free_slots(self):
        return self.lib.pp_pipeline_free_slots(self.h)

    def release(self, tickets):
        t = np.ascontiguousarray(tickets, dtype=np.uint64)
        check(self.lib.pp_pipeline_release(self.h, len(t), ptr(t)))
        for x in t:
            self._held.pop(int(x), None)

    def get_path_of(self, ticket):
        """solution path of a completed query polled with release=False"""
        slot = self.lib.pp_pipeline_slot_of(self.h, C.c_uint64(int(ticket)))
        if slot < 0:
            raise ValueError("ticket is not a completed, held query")
        n = self._held[int(ticket)].n_path
        poses, kind, prim, length, tuv = np.empty((n, 3)), np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32), np.empty(n), np.empty((n, 3))
        if n:
            check(self.lib.pp_planner_get_path(self.planner_h, slot, ptr(poses), ptr(kind), ptr(prim), ptr(length), ptr(tuv)))
        return dict(poses=poses, kind=kind, prim=prim, length=length, tuv=tuv)

    def postprocess_held(self, n_slots=None, path_interpolation=0.1, smoother=None, max_points=2048):
        """HybridAStar::SearchPath's post-processing (hybrid_a_star.cpp:260-304) over the field slots 0 .. n_slots-1 of the pipeline's buffer
        set (pp_planner_postprocess on pp_pipeline_planner()): meaningful for slots whose queries are completed and HELD (polled with
        release=False); read a query's processed path with get_processed_path_of(ticket)."""
        n = self.capacity if n_slots is None else int(n_slots)
        sp = None
        if smoother is not None:
            d = dict(step_tolerance=1e-3, max_iterations=2000, learning_rate=0.01, path_weight=0.0, smooth_weight=0.4, voronoi_weight=0.02, collision_weight=0.2,
                     curvature_weight=0.4, collision_ratio=0.2, max_curvature=1.0 / self.params.min_turning_radius)
            d.update(smoother)
            sp = SmootherParams(**d)
        out = (PostResult * max(n, 1))()
        check(self.lib.pp_planner_postprocess(self.planner_h, n, C.c_float(path_interpolation), C.byref(sp) if sp is not None else None, int(max_points), out))
        self._post = list(out)[:n]
        return self._post

    def get_processed_path_of(self, ticket):
        slot = self.lib.pp_pipeline_slot_of(self.h, C.c_uint64(int(ticket)))
        if slot < 0:
            raise ValueError("ticket is not a completed, held query")
        r = self._post[slot]
        n = r.n_points
        sampled, smoothed = np.empty((n, 3)), np.empty((n, 3))
        cusp = np.empty(n, dtype=np.uint8)
        if n:
            check(self.lib.pp_planner_get_processed_path(self.planner_h, slot, ptr(sampled), ptr(cusp), ptr(smoothed)))
        return dict(sampled=sampled, cusp=cusp.astype(bool), smoothed=smoothed, status=r.smoothing_status, iterations=r.iterations, length=r.length,
                    path=smoothed if r.smoothing_status >= 0 else sampled)

    def get_expanded_of(self, ticket):
        """expansion sequence (log_expansions=True) of a completed query polled with release=False"""
        slot = self.lib.pp_pipeline_slot_of(self.h, C.c_uint64(int(ticket)))
        if slot < 0:
            raise ValueError("ticket is not a completed, held query")
        n = self._held[int(ticket)].n_expanded
        cells = np.empty((n, 3), dtype=np.int32)
        if n:
            check(self.lib.pp_planner_get_expanded(self.planner_h, slot, ptr(cells)))
        return cells

    def close(self):
        if self.h:
            self.lib.pp_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _RRTBase:
    """RRT / RRTStar over R2 (algo/rrt.h, algo/rrt_star.h): the whole sequential loop runs on the device.
    validator=None reproduces StateValidatorFree (planner/tests/test_rrt*.cpp)."""
    _star = 0

    def __init__(self, ctx, lower, upper, validator=None, max_iteration=None, max_number_tree_node=10000, max_connection_distance=0.1,
                 goal_bias=0.05, rewire=False, radius_gamma=None):
        """rewire / radius_gamma (RRTStar only) go beyond the reference, whose RRT* has no rewire step (rrt_star.h:83): rewire=True
        re-parents near nodes through the new node and carries the saving down their subtrees; radius_gamma=g also replaces the
        k-nearest near-set by the <= 16 nearest nodes within g * sqrt(ln(n + 1) / (n + 1))."""
        self.ctx = ctx
        self.lib = ctx.lib
        if self._star and (rewire or radius_gamma is not None):
            self._star = 3 if radius_gamma is not None else 2
        self.gamma = float(radius_gamma) if radius_gamma is not None else 0.0
        self.lower = np.ascontiguousarray(lower, dtype=np.float64)[:2].copy()
        self.upper = np.ascontiguousarray(upper, dtype=np.float64)[:2].copy()
        self.validator = validator
        if max_iteration is None:
            max_iteration = 10000 if self._star else 100  # RRTStarParameters / RRTParameters defaults
        self.max_iteration = max_iteration
        self.max_number_tree_node = max_number_tree_node
        self.max_connection_distance = max_connection_distance
        self.goal_bias = goal_bias
        self._init = np.zeros(2)
        self._goal = np.zeros(2)
        self._seed = 0
        self.result = None

    def set_init_state(self, p):
        self._init = np.ascontiguousarray(p, dtype=np.float64)[:2].copy()

    def set_goal_state(self, p):
        self._goal = np.ascontiguousarray(p, dtype=np.float64)[:2].copy()

    def set_seed(self, seed):
        self._seed = int(seed)

    def search_path(self):
        params = np.array([self.max_iteration, self.max_number_tree_node, self.max_connection_distance, self.goal_bias, self.gamma], dtype=np.float64)
        h = C.c_void_p()
        res = _lib.RrtResult()
        mh = self.validator.map.h if self.validator is not None else None
        check(self.lib.pp_rrt_run(self.ctx.h, mh, ptr(self.lower), ptr(self.upper), ptr(params), ptr(self._init), ptr(self._goal),
                                  C.c_uint64(self._seed), self._star, C.byref(h), C.byref(res)))
        nodes = np.empty((res.n_nodes, 2))
        parents = np.empty(res.n_nodes, dtype=np.int32)
        costs = np.empty(res.n_nodes)
        path = np.empty((res.n_path, 2))
        check(self.lib.pp_rrt_get(h, ptr(nodes), ptr(parents), ptr(costs), ptr(path)))
        self.lib.pp_rrt_destroy(h)
        self.result = dict(status=res.status, nodes=nodes, parents=parents, costs=costs, path=path, iterations=res.iterations,
                           n_knn=res.n_knn_queries, n_edge_checks=res.n_edge_checks)
        return Status.SUCCESS if res.status == 0 else Status.FAILURE

    def get_path(self):
        return self.result["path"] if self.result is not None else np.empty((0, 2))

    def search_batch(self, inits, goals, seeds):
        """n independent problems (own start, goal, seed), one workgroup each, run together on the GPU.
        Returns one result dict per problem (same fields as `self.result`)."""
        inits = np.ascontiguousarray(inits, dtype=np.float64).reshape(-1, 2)
        goals = np.ascontiguousarray(goals, dtype=np.float64).reshape(-1, 2)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        n = len(inits)
        assert goals.shape == inits.shape and seeds.shape == (n,)
        params = np.array([self.max_iteration, self.max_number_tree_node, self.max_connection_distance, self.goal_bias, self.gamma], dtype=np.float64)
        hs = (C.c_void_p * n)()
        res = (_lib.RrtResult * n)()
        mh = self.validator.map.h if self.validator is not None else None
        check(self.lib.pp_rrt_run_batch(self.ctx.h, mh, ptr(self.lower), ptr(self.upper), ptr(params), n, ptr(inits), ptr(goals), ptr(seeds), self._star,
                                        C.cast(hs, C.POINTER(C.c_void_p)), res))
        out = []
        for i in range(n):
            nodes = np.empty((res[i].n_nodes, 2))
            parents = np.empty(res[i].n_nodes, dtype=np.int32)
            costs = np.empty(res[i].n_nodes)
            path = np.empty((res[i].n_path, 2))
            h = C.c_void_p(hs[i])
            check(self.lib.pp_rrt_get(h, ptr(nodes), ptr(parents), ptr(costs), ptr(path)))
            self.lib.pp_rrt_destroy(h)
            out.append(dict(status=res[i].status, nodes=nodes, parents=parents, costs=costs, path=path, iterations=res[i].iterations,
                            n_knn=res[i].n_knn_queries, n_edge_checks=res[i].n_edge_checks))
        return out


class RRT(_RRTBase):
    _star = 0


class RRTStar(_RRTBase):
    _star = 1


class GridAStarBatch:
    """AStarN2 / BidirectionalAStarN2 (algo/a_star_n2.h, algo/bidirectional_a_star.h) for many (init, goal) pairs at once, one wave per
    query on the device, with the transition cost and heuristic of the reference's script (example_a_star_grid.py:46-52: Euclidean
    distance between cells).  Other cost / heuristic functions run on the host engine (pyplanning.AStarN2, per-edge callbacks as in
    the reference)."""

    def __init__(self, map_set):
        self.map = map_set
        self.lib = map_set.lib

    def search_batch(self, inits, goals, bidirectional=False, inner_goals=None, max_path=None, max_expanded=None, want_expanded=None):
        """inits / goals: [n][2] (row, col).  inner_goals [n][4] (bidirectional only): the goals held by the two heuristics the
        AverageHeuristic pair wraps (default: forward -> goal, reverse -> init).  Returns one dict per query: status, cost, path
        [(row, col)], n_expanded and -- want_expanded (default: while n * cells <= 2^26) -- expanded (expansion order; + expanded_reverse
        when bidirectional)."""
        inits = np.ascontiguousarray(inits, dtype=np.int32).reshape(-1, 2)
        goals = np.ascontiguousarray(goals, dtype=np.int32).reshape(-1, 2)
        n = len(inits)
        assert goals.shape == inits.shape
        cells = self.map.rows * self.map.cols
        if want_expanded is None:  # the expansion orders are [n][cells][2] int32 per direction: only while that stays small
            want_expanded = max_expanded is not None or n * cells <= (1 << 26)
        max_path = int(max_path) if max_path is not None else min(cells + 1, 4 * (self.map.rows + self.map.cols))
        max_expanded = int(max_expanded) if max_expanded is not None else (cells if want_expanded else 0)
        ig = None
        if inner_goals is not None:
            ig = np.ascontiguousarray(inner_goals, dtype=np.int32).reshape(n, 4)
        res = (_lib.GridResult * max(n, 1))()
        paths = np.zeros((n, max_path, 2), dtype=np.int32)
        exp = np.zeros((n, max_expanded, 2), dtype=np.int32) if want_expanded else None
        expr = np.zeros((n, max_expanded, 2), dtype=np.int32) if want_expanded and bidirectional else None
        check(self.lib.pp_grid_astar_batch(self.map.h, n, ptr(inits), ptr(goals), 1 if bidirectional else 0, ptr(ig) if ig is not None else None,
                                           max_path, max_expanded if want_expanded else 0, res, ptr(paths), ptr(exp) if exp is not None else None,
                                           ptr(expr) if expr is not None else None))
        out = []
        for i in range(n):
            r = res[i]
            if r.n_path > max_path or (want_expanded and max(r.n_expanded, r.n_expanded_reverse) > max_expanded):
                raise _lib.PPError("query %d: path (%d) or expansion list (%d) longer than the buffers" % (i, r.n_path, r.n_expanded))
            d = dict(status=r.status, cost=r.cost, path=paths[i, :r.n_path].copy(), n_expanded=r.n_expanded, n_expanded_reverse=r.n_expanded_reverse)
            if want_expanded:
                d["expanded"] = exp[i, :r.n_expanded].copy()
                if bidirectional:
                    d["expanded_reverse"] = expr[i, :r.n_expanded_reverse].copy()
            out.append(d)
        return out
