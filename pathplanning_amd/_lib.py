"""ctypes binding of the C ABI (include/pp_hip.h).  No CPU fallback: if libpphip.so is
missing or no GPU is visible, calls fail loudly."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# The streaming pipeline (pp_pipeline_*) and bench.py keep several kernels running side by side on their own streams; the runtime maps
# streams onto 4 hardware queues by default, and two streams that share a queue serialise.  Read when the HIP runtime starts, so
# it only takes effect if nothing has touched the GPU yet (set it in the environment otherwise).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# PP_HIP_LIB: another build of the same library (tuning experiments: tools/build_variant.py)
LIB_PATH = os.environ.get("PP_HIP_LIB") or os.path.join(HERE, "lib", "libpphip.so")

c_dp = C.POINTER(C.c_double)
c_fp = C.POINTER(C.c_float)
c_ip = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)
c_u64p = C.POINTER(C.c_uint64)


class PPError(RuntimeError):
    """A libpphip call returned a negative code (include/pp_hip.h: PP_ERR_INVALID -1, PP_ERR_HIP -2, PP_ERR_NO_DEVICE -3,
    PP_ERR_CAPACITY -4); `code` holds it, the text is pp_last_error()."""
    code = 0


class MapDesc(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("resolution", C.c_float), ("grid_origin", C.c_double * 2),
                ("local_origin", C.c_double * 2), ("lower", C.c_double * 3), ("upper", C.c_double * 3)]


class HybridParams(C.Structure):
    _fields_ = [("wheelbase", C.c_double), ("min_turning_radius", C.c_double), ("direction_switching_cost", C.c_double),
                ("reverse_cost_multiplier", C.c_double), ("forward_cost_multiplier", C.c_double), ("voronoi_cost_multiplier", C.c_double),
                ("num_generated_motion", C.c_uint32), ("spatial_resolution", C.c_double), ("angular_resolution", C.c_double),
                ("heading_alias", C.c_int32), ("negative_k_read", C.c_int32)]

    @classmethod
    def default(cls, **kw):
        # HybridAStar::SearchParameters defaults, algo/hybrid_a_star.h:29-38
        d = dict(wheelbase=2.6, min_turning_radius=2.0, direction_switching_cost=0.0, reverse_cost_multiplier=1.0,
                 forward_cost_multiplier=1.0, voronoi_cost_multiplier=1.0, num_generated_motion=5, spatial_resolution=1.0,
                 angular_resolution=0.0872, heading_alias=1, negative_k_read=1)
        d.update(kw)
        return cls(**d)


class QueryResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_expanded", C.c_int32), ("n_nodes", C.c_int32), ("n_path", C.c_int32), ("cost", C.c_double),
                ("n_rng_draws", C.c_int32), ("n_rs_attempts", C.c_int32), ("n_state_checks", C.c_int64), ("n_path_checks", C.c_int64),
                ("n_lattice_boundary_hits", C.c_int32), ("reserved", C.c_int32)]


# the same record as a numpy dtype (arrays of results without a Python loop)
QUERY_RESULT_DTYPE = np.dtype([("status", "<i4"), ("n_expanded", "<i4"), ("n_nodes", "<i4"), ("n_path", "<i4"), ("cost", "<f8"), ("n_rng_draws", "<i4"),
                               ("n_rs_attempts", "<i4"), ("n_state_checks", "<i8"), ("n_path_checks", "<i8"), ("n_lattice_boundary_hits", "<i4"), ("reserved", "<i4")])
assert QUERY_RESULT_DTYPE.itemsize == C.sizeof(QueryResult)

# pp_rs_path (include/pp_hip.h): PathReedsShepp as a 128-byte record
RS_PATH_DTYPE = np.dtype([("start", "<f8", 3), ("final_pose", "<f8", 3), ("motion_length", "<f8", 5), ("steer", "i1", 5), ("direction", "i1", 5),
                          ("reserved", "i1", 6), ("min_turning_radius", "<f8"), ("length", "<f8"), ("cost", "<f4"), ("word", "<i4")])
assert RS_PATH_DTYPE.itemsize == 128


class SmootherParams(C.Structure):
    """Smoother::Parameters, algo/smoother.h:28-60"""
    _fields_ = [("step_tolerance", C.c_float), ("max_iterations", C.c_int32), ("learning_rate", C.c_float), ("path_weight", C.c_float), ("smooth_weight", C.c_float),
                ("voronoi_weight", C.c_float), ("collision_weight", C.c_float), ("curvature_weight", C.c_float), ("collision_ratio", C.c_float), ("max_curvature", C.c_float)]


class PostResult(C.Structure):
    _fields_ = [("n_points", C.c_int32), ("smoothing_status", C.c_int32), ("iterations", C.c_int32), ("reserved", C.c_int32), ("length", C.c_double)]


class GridResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_path", C.c_int32), ("n_expanded", C.c_int32), ("n_expanded_reverse", C.c_int32), ("cost", C.c_double)]


class RrtResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_nodes", C.c_int32), ("n_path", C.c_int32), ("iterations", C.c_int64),
                ("n_knn_queries", C.c_int64), ("n_edge_checks", C.c_int64)]


_lib = None


def load():
    """Loads libpphip.so; raises if it has not been built (run `python -m pathplanning_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PPError("libpphip.so not found at %s -- build it with `python -m pathplanning_amd.build` (hipcc, gfx950). "
                      "There is no CPU fallback." % LIB_PATH)
    try:
        # PyTorch ships its own copy of the HIP / HSA runtime.  Whichever copy initialises second in a process sees no
        # device, so when torch is installed it goes first and libpphip binds to the copy torch loaded.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.pp_last_error.restype = C.c_char_p
    L.pp_obstacle_heuristic_workspace_bytes.restype = C.c_int64
    vp = C.c_void_p
    L.pp_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.pp_ctx_destroy.argtypes = [vp]
    L.pp_ctx_synchronize.argtypes = [vp]
    L.pp_ctx_is_idle.argtypes = [vp, C.POINTER(C.c_int32)]
    L.pp_ctx_timer_start.argtypes = [vp]
    L.pp_ctx_timer_stop.argtypes = [vp, c_fp]
    L.pp_map_create.argtypes = [vp, C.POINTER(MapDesc), C.POINTER(vp)]
    L.pp_map_destroy.argtypes = [vp]
    L.pp_map_upload_dist2.argtypes = [vp, vp]
    L.pp_map_upload_occupancy.argtypes = [vp, vp]
    L.pp_map_upload_path_cost.argtypes = [vp, vp]
    L.pp_map_set_validator.argtypes = [vp, C.c_float, C.c_float]
    L.pp_map_download_distance.argtypes = [vp, vp]
    L.pp_check_states.argtypes = [vp, C.c_int64, vp, vp]
    L.pp_check_states_dev.argtypes = [vp, C.c_int64, vp, vp]
    L.pp_check_states_fused_dev.argtypes = [vp, C.c_int64, C.c_uint64, vp]
    L.pp_check_arcs.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, vp]
    L.pp_check_arcs_dev.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, vp]
    L.pp_check_segments.argtypes = [vp, C.c_int64, vp, vp, vp]
    L.pp_check_segments_dev.argtypes = [vp, C.c_int64, vp, vp, vp]
    L.pp_rollout_children.argtypes = [vp, C.POINTER(HybridParams), C.c_int32, vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp]
    L.pp_rollout_children_dev.argtypes = [vp, C.POINTER(HybridParams), C.c_int32, vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp]
    L.pp_map_upload_distance.argtypes = [vp, vp]
    L.pp_map_rasterize_segments.argtypes = [vp, C.c_int32, vp, vp, C.c_int32, vp]
    L.pp_map_download_occupancy.argtypes = [vp, vp]
    L.pp_map_update_gvd.argtypes = [vp, C.c_float, C.c_float, vp]
    L.pp_map_update_gvd_ex.argtypes = [vp, C.c_float, C.c_float, C.c_int32, vp]
    L.pp_map_download_gvd.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.pp_path_cost_update.argtypes = [vp, vp, vp, C.c_float, C.c_float, vp]
    L.pp_rs_connect.argtypes = [vp, C.c_int64, vp, vp, C.c_double, C.c_float, C.c_float, C.c_float, vp]
    L.pp_rs_path_interpolate.argtypes = [vp, C.c_int64, vp, vp, vp, vp]
    L.pp_rs_path_truncate.argtypes = [vp, C.c_int64, vp, vp, C.c_int32]
    L.pp_rs_path_cusps.argtypes = [vp, C.c_int64, vp, vp, vp]
    L.pp_check_rs_paths.argtypes = [vp, C.c_int64, vp, vp, vp]
    L.pp_check_rs_paths_dev.argtypes = [vp, C.c_int64, vp, vp, vp]
    L.pp_check_se2_paths.argtypes = [vp, C.c_int64, vp, vp, vp, vp]
    L.pp_rs_solve.argtypes = [vp, C.c_int64, vp, vp, C.c_double, C.c_float, C.c_float, C.c_float, vp, vp, vp, vp]
    L.pp_rs_solve_dev.argtypes = [vp, C.c_int64, vp, vp, C.c_double, C.c_float, C.c_float, C.c_float, vp, vp, vp, vp]
    L.pp_nonholo_dims.argtypes = [vp, vp, C.POINTER(HybridParams), vp, vp]
    L.pp_nonholo_build.argtypes = [vp, vp, vp, C.POINTER(HybridParams), vp]
    L.pp_nonholo_build_dev.argtypes = [vp, vp, vp, C.POINTER(HybridParams), vp]
    L.pp_obstacle_heuristic.argtypes = [vp, C.c_int32, vp, vp]
    L.pp_obstacle_heuristic_dev.argtypes = [vp, C.c_int32, vp, vp]
    L.pp_obstacle_heuristic_workspace_bytes.argtypes = [vp]
    L.pp_obstacle_heuristic_profile.argtypes = [vp, C.c_int32, vp, vp]
    L.pp_obstacle_heuristic_tiles_stats.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp]
    L.pp_planner_create.argtypes = [vp, C.POINTER(HybridParams), C.c_int32, C.c_int32, C.POINTER(vp)]
    L.pp_planner_search_rows.argtypes = [vp]
    L.pp_planner_debug_nodes.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, vp]
    L.pp_planner_debug_node_actions.argtypes = [vp, C.c_int32, C.c_int32, vp, vp]
    L.pp_planner_certify_lattice.argtypes = [vp, C.c_int32, vp, vp, vp, vp]
    L.pp_planner_create_ex.argtypes = [vp, C.POINTER(HybridParams), C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.pp_planner_destroy.argtypes = [vp]
    L.pp_planner_set_nonholo_table.argtypes = [vp, vp]
    L.pp_planner_get_nonholo_table.argtypes = [vp, vp]
    L.pp_planner_num_primitives.argtypes = [vp]
    L.pp_planner_set_primitives.argtypes = [vp, C.c_int32, vp]
    L.pp_planner_search_batch.argtypes = [vp, C.c_int32, vp, vp, vp, vp]
    L.pp_planner_search_batch_dev.argtypes = [vp, C.c_int32, vp, vp, vp]
    L.pp_planner_fetch_results.argtypes = [vp, C.c_int32, vp]
    L.pp_planner_get_path.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp]
    L.pp_planner_get_expanded.argtypes = [vp, C.c_int32, vp]
    L.pp_pipeline_create.argtypes = [vp, C.POINTER(HybridParams), C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]
    L.pp_pipeline_destroy.argtypes = [vp]
    L.pp_pipeline_submit_dev.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp]
    L.pp_pipeline_submit.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp]
    L.pp_pipeline_poll.argtypes = [vp, C.c_int32, vp, vp, C.c_int32, vp]
    L.pp_pipeline_release.argtypes = [vp, C.c_int32, vp]
    L.pp_pipeline_get_paths.argtypes = [vp, C.c_int32, vp, C.c_int32, vp, vp, C.c_int32]
    L.pp_pipeline_slot_of.argtypes = [vp, C.c_uint64]
    L.pp_pipeline_timings.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.pp_pipeline_backlog.argtypes = [vp, vp, vp]
    L.pp_pipeline_planner.argtypes = [vp]
    L.pp_pipeline_planner.restype = vp
    for f in ("pp_pipeline_capacity", "pp_pipeline_search_rows", "pp_pipeline_in_flight", "pp_pipeline_free_slots", "pp_pipeline_alive_waves"):
        getattr(L, f).argtypes = [vp]
    L.pp_planner_postprocess.argtypes = [vp, C.c_int32, C.c_float, vp, C.c_int32, vp]
    L.pp_planner_get_processed_path.argtypes = [vp, C.c_int32, vp, vp, vp]
    L.pp_map_upload_nearest_cells.argtypes = [vp, vp, vp]
    L.pp_planner_last_timings.argtypes = [vp, c_fp, c_fp]
    L.pp_planner_set_profiling.argtypes = [vp, C.c_int32]
    L.pp_planner_phase_cycles.argtypes = [vp, C.c_int32, vp]
    L.pp_knn.argtypes = [vp, C.c_int64, vp, C.c_int64, vp, C.c_int32, vp, vp]
    L.pp_knn_dev.argtypes = [vp, C.c_int64, vp, C.c_int64, vp, C.c_int32, vp, vp]
    L.pp_rrt_run.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_uint64, C.c_int32, C.POINTER(vp), C.POINTER(RrtResult)]
    L.pp_rrt_run_batch.argtypes = [vp, vp, vp, vp, vp, C.c_int32, vp, vp, vp, C.c_int32, C.POINTER(vp), C.POINTER(RrtResult)]
    L.pp_rrt_get.argtypes = [vp, vp, vp, vp, vp]
    L.pp_rrt_destroy.argtypes = [vp]
    L.pp_planner_start_after_fields_of.argtypes = [vp, vp]
    L.pp_grid_astar_batch.argtypes = [vp, C.c_int32, vp, vp, C.c_int32, vp, C.c_int32, C.c_int32, C.POINTER(GridResult), vp, vp, vp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        e = PPError("libpphip error %d: %s" % (rc, load().pp_last_error().decode()))
        e.code = rc
        raise e


def ptr(a):
    """Host pointer of a contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)
