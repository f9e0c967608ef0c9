/*
 * pp_hip.h -- C ABI of the MI355X-native planner core (libpphip.so).
 *
 * Drop-in boundary for the hot path of lfilipozzi/PathPlanning.
 * The reference has no FFI of its own: its plugin surface is the C++ abstract
 * classes StateValidator / PathPlanner plus the pybind11 module.  These entry
 * points are what those classes call into (see INTEGRATION.md for the binding a
 * reference maintainer adds).  Each function cites the reference interface it
 * replaces, paths relative to the reference's planner/src.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (never throws);
 *     pp_last_error() gives the message of the calling thread's last failure;
 *   - *_dev functions take DEVICE pointers and only enqueue work on the context's
 *     stream (no host synchronisation); the others take HOST pointers, copy,
 *     run and synchronise;
 *   - poses are 3 contiguous doubles {x, y, theta} exactly like Pose2d
 *     (geometry/2dplane.h:17-34); grids are row-major, index = row*cols + col,
 *     row <- x, col <- y (utils/grid.h:87, state_validator/occupancy_map.h:106-117);
 *   - no torch types, no C++ types, no global state besides the error string.
 */
#ifndef PP_HIP_H
#define PP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_OK 0
#define PP_ERR_INVALID (-1)
#define PP_ERR_HIP (-2)
#define PP_ERR_NO_DEVICE (-3)
#define PP_ERR_CAPACITY (-4)

typedef struct pp_ctx pp_ctx;         /* one device + one stream */
typedef struct pp_map pp_map;         /* device-resident map set of one OccupancyMap */
typedef struct pp_planner pp_planner; /* Hybrid-A* tables + per-query workspaces */

const char* pp_last_error(void);
int pp_version(void);

/* ---- context ----------------------------------------------------------- */
/* stream == NULL: the context creates and owns a stream; otherwise it enqueues
 * on the caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). */
int pp_ctx_create(int device, void* stream, pp_ctx** out);
int pp_ctx_destroy(pp_ctx* ctx);
int pp_ctx_synchronize(pp_ctx* ctx);
/* Non-blocking: *idle = 1 when everything enqueued on the context's stream has completed (hipStreamQuery), else 0.
 * For callers that keep several contexts busy and refill whichever finishes first. */
int pp_ctx_is_idle(pp_ctx* ctx, int32_t* idle);
/* HIP-event timer on the context's stream: start, stop -> milliseconds. */
int pp_ctx_timer_start(pp_ctx* ctx);
int pp_ctx_timer_stop(pp_ctx* ctx, float* ms);
int pp_device_count(void);

/* ---- map set ------------------------------------------------------------
 * Replaces the data members of OccupancyMap (state_validator/occupancy_map.h:122-133),
 * GVD::ObstacleDistanceMap::m_distance (state_validator/gvd.h:58), GVD::PathCostMap
 * (gvd.h:111-121) and the StateSpaceSE2 bounds (state_space/state_space.h:60-61). */
typedef struct pp_map_desc {
	int32_t rows, cols;
	float resolution;         /* OccupancyMap::resolution (float) */
	double grid_origin[2];    /* m_worldGridOrigin */
	double local_origin[2];   /* m_localOrigin */
	double lower[3], upper[3]; /* StateSpaceSE2::bounds */
} pp_map_desc;

int pp_map_create(pp_ctx* ctx, const pp_map_desc* desc, pp_map** out);
int pp_map_destroy(pp_map* map);
/* int32 squared distances to the nearest obstacle, INT_MAX = none (gvd.cpp:21). */
int pp_map_upload_dist2(pp_map* map, const int32_t* d2_host);
/* The same grid as the reference's accessor returns it: float distance in metres,
 * ObstacleDistanceMap::GetDistanceToNearestObstacle(row, col) = sqrt(d2) * resolution (gvd.h:38).  Either upload
 * serves; this one needs no access to the private int grid and loses nothing for INT_MAX / d2 > 2^24 cells. */
int pp_map_upload_distance(pp_map* map, const float* distance_host);
/* int32 occupancy, >= 0 occupied, < 0 free (obstacle_list_occupancy_map.cpp:63-69). */
int pp_map_upload_occupancy(pp_map* map, const int32_t* occ_host);
/* float Voronoi-field potential, GVD::GetPathCost (gvd.h:160-162). */
int pp_map_upload_path_cost(pp_map* map, const float* cost_host);
/* ---- map authoring and field construction on the device ----
 * Shape::RasterizeLine (state_validator/obstacle.cpp:7-61) for n segments given by their world end points (the caller rotates and
 * translates the shape's vertices, PolygonShape::GetVerticesPosition, obstacle.cpp:86-93), writing `value` into every boundary
 * cell: an obstacle id for ObstacleListOccupancyMap::AddObstacle, -1 for RemoveObstacle
 * (obstacle_list_occupancy_map.cpp:29-61).  n_cells_out (may be NULL): cells written, with repeats. */
int pp_map_rasterize_segments(pp_map* map, int32_t n_segments, const double* p0_xy_host, const double* p1_xy_host, int32_t value, int32_t* n_cells_out);
/* The same walk with the cells listed instead of written (Shape::GetGridCellsPosition, Obstacle::GetBoundaryGridCellPosition):
 * segment i fills cells_host[i * cap_per_segment * 2 ...] with (row, col) pairs in Bresenham order, count_host[i] of them.
 * pp_map_set_cells writes `value` into listed cells (AddObstacle / RemoveObstacle for any Shape, also caller-defined ones). */
int pp_rasterize_cells(pp_map* map, int32_t n_segments, const double* p0_xy_host, const double* p1_xy_host, int32_t cap_per_segment, int32_t* cells_host,
	int32_t* count_host);
int pp_map_set_cells(pp_map* map, int64_t n_cells, const int32_t* cells_host, int32_t value);
int pp_map_download_occupancy(pp_map* map, int32_t* occ_host);
/* GVD::Update (state_validator/gvd.cpp:294-301) from the device occupancy grid: squared obstacle distance + nearest obstacle cell
 * (ObstacleDistanceMap::Update, gvd.cpp:30-72), Voronoi edges (CheckVoro, :105-131), squared distance to the nearest edge
 * (VoronoiDistanceMap::Update, :200-237), PathCostMap (:266-283; alpha, d_max: GVD::alpha / dMax, gvd.h:181); also refreshes what
 * the validator reads.  Two modes for the two distance maps:
 *   PP_GVD_REFERENCE_ORDER  the reference's dynamic brushfire itself, replayed (on the host: it is a sequential priority-queue sweep
 *                           whose order among equal keys is part of the result) over the ORDERED cell edits this map has received
 *                           -- pp_map_set_cells / pp_map_rasterize_segments in call order, cells in list order, i.e.
 *                           SetObstacle / UnsetObstacle (gvd.cpp:74-89) as AddObstacle / RemoveObstacle issue them; a
 *                           pp_map_upload_occupancy counts as an empty map followed by its occupied cells in row-major order.
 *                           Grids equal the reference's bit for bit; edits after the first update are incremental.
 *   PP_GVD_EXACT_EDT        exact Euclidean transform on the device (milliseconds, no host round trip): true minima, which the
 *                           brushfire's values are not always, and its own tie rule -- NOT the reference's bits on a few cells in
 *                           ten thousand.  pp_map_update_gvd = this mode.
 * iterations_out (may be NULL): heap pops of the brushfire so far / device passes. */
enum { PP_GVD_EXACT_EDT = 0, PP_GVD_REFERENCE_ORDER = 1 };
int pp_map_update_gvd_ex(pp_map* map, float alpha, float d_max, int32_t mode, int32_t* iterations_out);
int pp_map_update_gvd(pp_map* map, float alpha, float d_max, int32_t* iterations_out);
/* any pointer may be NULL; nearest_*: (row, col) per cell, (-1, -1) = none */
int pp_map_download_gvd(pp_map* map, int32_t* d2_host, int32_t* nearest_obstacle_host, uint8_t* voronoi_edge_host, int32_t* voronoi_d2_host, int32_t* nearest_edge_host,
	float* path_cost_host);
/* PathCostMap::Update alone (gvd.cpp:266-283) over two squared-distance grids of the caller; keeps the result as the map's path
 * cost and copies it out when path_cost_host != NULL. */
int pp_path_cost_update(pp_map* map, const int32_t* obstacle_d2_host, const int32_t* voronoi_d2_host, float alpha, float d_max, float* path_cost_host);

/* StateValidatorOccupancyMap::minSafeRadius / minPathInterpolationDistance
 * (state_validator/state_validator_occupancy_map.h:27-28). */
int pp_map_set_validator(pp_map* map, float min_safe_radius, float min_path_interpolation_distance);
/* Copies the derived float distance grid back (tests): (float)(sqrt((double)d2) * resolution), gvd.h:38. */
int pp_map_download_distance(pp_map* map, float* dist_host);

/* ---- a1: StateValidatorOccupancyMap::IsStateValid -------------------------
 * (state_validator/state_validator_occupancy_map.cpp:15-26), batched. */
int pp_check_states(pp_map* map, int64_t n, const double* poses_host, uint8_t* valid_host);
int pp_check_states_dev(pp_map* map, int64_t n, const double* poses_dev, uint8_t* valid_dev);
/* Fused microbench form ("M1 fused"): poses are generated in-kernel from a
 * counter-based hash, only a per-block count of valid poses leaves the kernel. */
int pp_check_states_fused_dev(pp_map* map, int64_t n, uint64_t seed, uint64_t* valid_count_dev);

/* ---- a2+a3: IsPathValid over constant-steer arcs --------------------------
 * (state_validator_occupancy_map.cpp:28-71 with paths/path_constant_steer.cpp:11-20 and
 * models/kinematic_bicycle_model.cpp:5-32).  curvature = cos(beta)*tan(steer)/wheelbase
 * (kinematic_bicycle_model.cpp:17), computed by the caller with libm. direction: 0 fwd, 1 bwd. */
int pp_check_arcs_dev(pp_map* map, int64_t n, const double* from_dev, const double* curvature_dev, const double* length_dev,
	const int32_t* direction_dev, uint8_t* valid_dev, float* last_ratio_dev);
int pp_check_arcs(pp_map* map, int64_t n, const double* from_host, const double* curvature_host, const double* length_host,
	const int32_t* direction_host, uint8_t* valid_host, float* last_ratio_host);
/* R2 segments (paths/path_r2.cpp) against the same validator with theta = 0: the
 * RRT / RRT* edge check on an occupancy map. */
int pp_check_segments_dev(pp_map* map, int64_t n, const double* from_xy_dev, const double* to_xy_dev, uint8_t* valid_dev);
int pp_check_segments(pp_map* map, int64_t n, const double* from_xy_host, const double* to_xy_host, uint8_t* valid_host);

/* ---- a4: HybridAStar::StatePropagator::GetConstantSteerChild ---------------
 * (algo/hybrid_a_star.cpp:111-147, 41-48, 93-109; hybrid_a_star.h:104-111).
 * One child per (parent, primitive), reference order: primitive p = 2*deltaIndex + dir. */
typedef struct pp_hybrid_params {
	double wheelbase;                /* SearchParameters::wheelbase */
	double min_turning_radius;
	double direction_switching_cost;
	double reverse_cost_multiplier;
	double forward_cost_multiplier;
	double voronoi_cost_multiplier;
	uint32_t num_generated_motion;
	double spatial_resolution;
	double angular_resolution;
	/* reference-behaviour switches; 1 = as the Release build behaves */
	int32_t heading_alias;    /* Q6 */
	int32_t negative_k_read;  /* Q7 */
} pp_hybrid_params;

int pp_rollout_children_dev(pp_map* map, const pp_hybrid_params* params, int32_t n_primitives, const double* curvature_host,
	const int32_t* direction_host, int64_t n_parents, const double* parents_dev, uint8_t* valid_dev, double* pose_dev, int32_t* key_dev,
	double* cost_dev, double* length_dev);
int pp_rollout_children(pp_map* map, const pp_hybrid_params* params, int32_t n_primitives, const double* curvature_host,
	const int32_t* direction_host, int64_t n_parents, const double* parents_host, uint8_t* valid_host, double* pose_host, int32_t* key_host,
	double* cost_host, double* length_host);

/* ---- a6: ReedsShepp::Solver::GetOptimalPath -------------------------------
 * (geometry/reeds_shepp.cpp:654-683), one (from, to) pair per element.
 * word: PathWords index or -1; tuv: the three parameters; cost: float as the
 * reference compares it; seg_length: normalised length. */
int pp_rs_solve_dev(pp_ctx* ctx, int64_t n, const double* from_dev, const double* to_dev, double min_turning_radius, float reverse_cost,
	float forward_cost, float switch_cost, int32_t* word_dev, double* tuv_dev, float* cost_dev, double* seg_length_dev);
int pp_rs_solve(pp_ctx* ctx, int64_t n, const double* from_host, const double* to_host, double min_turning_radius, float reverse_cost,
	float forward_cost, float switch_cost, int32_t* word_host, double* tuv_host, float* cost_host, double* seg_length_host);

/* ---- a7: PathReedsShepp as a value (paths/path_reeds_shepp.{h,cpp}) ---------
 * What the reference's object holds: m_init, m_final, the five ReedsShepp::Motion slots of its PathSegment
 * (geometry/reeds_shepp.h:54-67: steer, direction, normalised length; +inf / NoMotion = unused slot),
 * m_minTurningRadius and m_length (metres; after Truncate it is scaled by the ratio and no longer the sum of the slots).
 * steer: 0 Left, 1 Straight, 2 Right (paths/path.h:10-14); direction: 0 Forward, 1 Backward, 2 NoMotion (:16-20). */
typedef struct pp_rs_path {
	double start[3];
	double final_pose[3];
	double motion_length[5];
	int8_t steer[5];
	int8_t direction[5];
	int8_t reserved[6];
	double min_turning_radius;
	double length;
	float cost;   /* pp_rs_connect: PathSegment::ComputeCost of the chosen word (float, as the reference compares it) */
	int32_t word; /* pp_rs_connect: PathWords index, -1 = NoPath; carried along otherwise */
} pp_rs_path;

/* PathConnectionReedsShepp::Connect (path_reeds_shepp.cpp:174-179): GetOptimalPath + the PathReedsShepp constructor
 * (m_final = Interpolate(1.0)), one path per (from, to) pair. */
int pp_rs_connect(pp_ctx* ctx, int64_t n, const double* from_host, const double* to_host, double min_turning_radius, float reverse_cost,
	float forward_cost, float switch_cost, pp_rs_path* paths_host);
/* PathReedsShepp::Interpolate (:12-47) and ::GetDirection (:155-167) of path i at ratio_host[i]; either output may be NULL. */
int pp_rs_path_interpolate(pp_ctx* ctx, int64_t n, const pp_rs_path* paths_host, const double* ratio_host, double* pose_host, int32_t* direction_host);
/* PathReedsShepp::Truncate (:49-93) in place, INCLUDING the reference's wrong-slot reset when q11 != 0 (SURVEY
 * Appendix A Q11: truncating inside motion i < 4 invalidates motion i itself); q11 == 0 drops the later motions instead. */
int pp_rs_path_truncate(pp_ctx* ctx, int64_t n, pp_rs_path* paths_host, const double* ratio_host, int32_t q11);
/* PathReedsShepp::GetCuspPointRatios (:95-121): ascending, no duplicates; ratios_host is [n][4], count_host[i] <= 4. */
int pp_rs_path_cusps(pp_ctx* ctx, int64_t n, const pp_rs_path* paths_host, double* ratios_host, int32_t* count_host);
/* StateValidatorOccupancyMap::IsPathValid (state_validator_occupancy_map.cpp:28-71) over Reeds-Shepp paths. */
int pp_check_rs_paths_dev(pp_map* map, int64_t n, const pp_rs_path* paths_dev, uint8_t* valid_dev, float* last_ratio_dev);
int pp_check_rs_paths(pp_map* map, int64_t n, const pp_rs_path* paths_host, uint8_t* valid_host, float* last_ratio_host);
/* ... and over PathSE2 (paths/path_se2.cpp:5-22: position and heading interpolated linearly, length = |to - from|). */
int pp_check_se2_paths(pp_map* map, int64_t n, const double* from_host, const double* to_host, uint8_t* valid_host, float* last_ratio_host);

/* ---- a10: NonHolonomicHeuristic::Build (algo/heuristics.cpp:36-76) ---------
 * dims = {nX, nY, nAngular}; table[(i*nY + j)*nAngular + k]. */
int pp_nonholo_dims(const double lower[3], const double upper[3], const pp_hybrid_params* params, int32_t dims[3], double offsets[2]);
int pp_nonholo_build_dev(pp_ctx* ctx, const double lower[3], const double upper[3], const pp_hybrid_params* params, double* table_dev);
int pp_nonholo_build(pp_ctx* ctx, const double lower[3], const double upper[3], const pp_hybrid_params* params, double* table_host);

/* ---- a8: ObstaclesHeuristic::Update (algo/heuristics.cpp:106-153) ----------
 * One wavefront per goal; cost_dev is [n_goals][rows*cols] float, +inf where the
 * reference leaves the cell unexplored.  The reference's values bit for bit (first discovery,
 * no relaxation, LIFO ties): built as the fixed point those rules define, one wave per goal over
 * 64 x 64-cell tiles (pp_wavefront_tiles.hip); a goal whose value could depend on the pop order among
 * equal costs is detected and rebuilt in the reference's own order (pp_wavefront.hip), as is every goal
 * when PP_WF_TILES=0 is in the environment.  goal_xy: world positions. */
int pp_obstacle_heuristic_dev(pp_map* map, int32_t n_goals, const double* goal_xy_host, float* cost_dev);
int pp_obstacle_heuristic(pp_map* map, int32_t n_goals, const double* goal_xy_host, float* cost_host);
/* Diagnostics of the tile form (a stamped instantiation of its kernel): the same fields into cost_dev ([n_goals][rows*cols]), plus 16 words
 * {goals built, tile visits, bucket rounds, candidate passes, cells settled, goals handed to the ordered kernel,
 * summed wave cycles, tiles per goal, then shader-clock sums of a tile visit's phases: loads + LDS set-up, the rounds' masks,
 * their candidate passes, re-queueing, store issue; 3 spare} and the launch's duration in ms (HIP events on the context's stream).
 * handed_over_host (optional, [n_goals]): the indices of the goals that were handed over, in no particular order.
 * PP_ERR_INVALID when the tile form is switched off or does not support the map. */
int pp_obstacle_heuristic_tiles_stats(pp_map* map, int32_t n_goals, const double* goal_xy_host, float* cost_dev, uint64_t stats_host[16], float* ms_out, int32_t* handed_over_host);
/* Diagnostics: stamped build of the wavefront kernel; per goal 20 words {init, min, partition, sort, offer, push,
 * tail cycles, rounds, sum of window sizes, rounds with the open list in HBM, fallback rounds, push cycles of fallback rounds,
 * offer sub-phases: store wait, neighbourhood loads, candidate count, whole offer up to the end of insertion,
 * push sub-phases (cumulative from the start of push): look-up, slot scan, stores; one pad word}. */
int pp_obstacle_heuristic_profile(pp_map* map, int32_t n_goals, const double* goal_xy_host, uint64_t* counters_host);
/* bytes of scratch pp_obstacle_heuristic_dev keeps per concurrently running goal */
int64_t pp_obstacle_heuristic_workspace_bytes(pp_map* map);

/* ---- a5/a9/a11: HybridAStar graph search, batched --------------------------
 * (algo/hybrid_a_star.cpp:59-91,149-173,237-257; algo/a_star.h:326-427;
 * algo/heuristics.cpp:78-95,155-165; utils/frontier.h).  One independent query per
 * (start, goal, seed); seed reseeds the query's own mt19937_64 (utils/random.h). */
int pp_planner_create(pp_map* map, const pp_hybrid_params* params, int32_t max_batch, int32_t max_nodes_per_query, pp_planner** out);
/* Same with an explicit number of search rows.  Planners sized for throughput (max_batch > 64) run the graph search
 * four queries per wave on a persistent grid whose rows take queries from a counter; node / heap / key-map buffers
 * exist per ROW, not per query.  search_rows = 0: as many rows as can be resident on the device (or max_batch if
 * smaller); callers that keep several planners in flight on one GPU pass resident rows / planners. */
int pp_planner_create_ex(pp_map* map, const pp_hybrid_params* params, int32_t max_batch, int32_t max_nodes_per_query, int32_t search_rows,
	pp_planner** out);
/* Diagnostics: the search tree of query q in creation order (parent index, pose, {pathCost, totalCost}, replaced flag);
 * available while node records are kept per query (one-query-per-wave kernel).  Any pointer may be NULL. */
int pp_planner_debug_nodes(pp_planner* planner, int32_t q, int32_t max_nodes, int32_t* parents_host, double* poses_host, double* costs_host,
	int32_t* dead_host);
/* ... and per node the action that created it (-1 root, 0..P-1 constant-steer primitive, 1000 + word Reeds-Shepp) and the
 * length of that edge: with the poses above, the reference's GetGraphSearchExploredPathSet (algo/hybrid_a_star.cpp:186-198). */
int pp_planner_debug_node_actions(pp_planner* planner, int32_t q, int32_t max_nodes, int32_t* action_host, double* length_host);
/* Rows the planner's search runs with (k_hybrid_search_rows); 0 = the one-query-per-wave kernel (k_hybrid_search). */
int pp_planner_search_rows(pp_planner* planner);
int pp_planner_destroy(pp_planner* planner);
/* Uses a table built elsewhere (host pointer), or builds it on the device when NULL. */
int pp_planner_set_nonholo_table(pp_planner* planner, const double* table_host);
int pp_planner_get_nonholo_table(pp_planner* planner, double* table_host);
int pp_planner_num_primitives(pp_planner* planner);
/* Replaces StatePropagator::m_deltas (algo/hybrid_a_star.cpp:21-28 generates {0, +-0.5 dMax, +-1.0 dMax, ...} from num_generated_motion,
 * which can only give 2 * odd primitives): any list of steering angles [rad]; every angle gives a forward and a backward primitive,
 * children in list order, forward first (hybrid_a_star.cpp:65-77).  36 angles = the "72 motion primitives" of BASELINE config 2.
 * Waits for the planner's stream; applies from the next batch on. */
int pp_planner_set_primitives(pp_planner* planner, int32_t n_steering_angles, const double* steering_angles);

typedef struct pp_query_result {
	int32_t status;        /* 0 = Success, -1 = Failure (algo/path_planner.h:9-12), -4 = node capacity exceeded */
	int32_t n_expanded;    /* expansions (frontier pops that were expanded) */
	int32_t n_nodes;       /* nodes created */
	int32_t n_path;        /* poses on the solution path (root..goal) */
	double cost;           /* GetGraphSearchOptimalCost */
	int32_t n_rng_draws;
	int32_t n_rs_attempts;
	int64_t n_state_checks;
	int64_t n_path_checks;
	/* Guard band of the exactness contract: poses of this query (start, children, Reeds-Shepp child) whose
	 * DiscretizePose (hybrid_a_star.h:104-111) truncated a coordinate within 1e-9 cells of a lattice boundary -- the only places
	 * where a last-bit libm difference could select another cell.  0: the query's discrete outputs equal the reference's by
	 * construction; > 0 (e.g. a start or goal placed exactly on a multiple of the resolution): equal as far as observed. */
	int32_t n_lattice_boundary_hits;
	int32_t reserved;
} pp_query_result;

/* Runs the batch (obstacle heuristic + graph search per query).  All pointers host. */
int pp_planner_search_batch(pp_planner* planner, int32_t n_queries, const double* starts_host, const double* goals_host,
	const uint64_t* seeds_host, pp_query_result* results_host);
/* Same, inputs already resident (starts/goals/seeds device pointers); results stay on the device until fetched. */
int pp_planner_search_batch_dev(pp_planner* planner, int32_t n_queries, const double* starts_dev, const double* goals_dev,
	const uint64_t* seeds_dev);
int pp_planner_fetch_results(pp_planner* planner, int32_t n_queries, pp_query_result* results_host);
/* Throughput use, several planners with a batch in flight each (own stream each): the NEXT pp_planner_search_batch*_dev of `planner`
 * starts on the device only when the obstacle-heuristic fields of `predecessor`'s most recently launched batch are built (one-shot,
 * cleared by that call).  Batches launched together otherwise run in phase -- all wavefronts, then all searches, then all tails of long
 * queries with the GPU nearly empty; started one wavefront apart they interleave.  Scheduling only: results do not depend on it. */
int pp_planner_start_after_fields_of(pp_planner* planner, pp_planner* predecessor);
/* Solution path of query q: poses (3*n_path doubles), per-node action kind (0 root, 1 arc, 2 Reeds-Shepp),
 * primitive index (arc) or RS word, arc length; expanded: cells in expansion order (3*n_expanded ints). Any pointer may be NULL. */
int pp_planner_get_path(pp_planner* planner, int32_t q, double* poses_host, int32_t* kind_host, int32_t* prim_host, double* length_host,
	double* tuv_host);
int pp_planner_get_expanded(pp_planner* planner, int32_t q, int32_t* cells_host);
/* ---- after the graph search: HybridAStar::SearchPath's post-processing, batched -----------------
 * (algo/hybrid_a_star.cpp:260-304: composite path of the solution's edges, sampling every path_interpolation metres with
 * cusp snapping, then Smoother::Smooth, algo/smoother.cpp:33-226) for the first n_queries queries of the last batch, one
 * workgroup per query.  Needs the nearest-obstacle / nearest-Voronoi-edge cell grids on the map (pp_map_update_gvd builds
 * them, pp_map_upload_nearest_cells takes the reference's GVD::GetNearestObstacleCell / GetNearestVoronoiEdgeCell).
 * smoother == NULL: Smoother::Parameters defaults with maxCurvature = 1 / minTurningRadius (hybrid_a_star.cpp:214). */
typedef struct pp_smoother_params { /* algo/smoother.h:28-60 */
	float step_tolerance;
	int32_t max_iterations;
	float learning_rate, path_weight, smooth_weight, voronoi_weight, collision_weight, curvature_weight, collision_ratio, max_curvature;
} pp_smoother_params;
typedef struct pp_post_result {
	int32_t n_points;         /* poses of the sampled path (0: the search failed) */
	int32_t smoothing_status; /* Smoother::Status: 0 MaxIteration, 1 StepTolerance, 2 PathSize, -1 Failure; -4: more than max_points samples */
	int32_t iterations;
	int32_t reserved;
	double length;            /* length of the composite path */
} pp_post_result;
int pp_planner_postprocess(pp_planner* planner, int32_t n_queries, float path_interpolation, const pp_smoother_params* smoother, int32_t max_points,
	pp_post_result* results_host);
/* sampled path (3 doubles per pose), cusp flags, smoothed path of query q; any pointer may be NULL.  HybridAStar::GetPath() is the
 * smoothed path when smoothing_status >= 0, else the sampled one (hybrid_a_star.cpp:293-303). */
int pp_planner_get_processed_path(pp_planner* planner, int32_t q, double* sampled_host, uint8_t* cusp_host, double* smoothed_host);
/* (row, col) per cell, row-major, (-1, -1) = none: the two label grids of the reference's GVD for maps whose fields were built elsewhere */
int pp_map_upload_nearest_cells(pp_map* map, const int32_t* nearest_obstacle_host, const int32_t* nearest_edge_host);

/* The guard band of the lattice as a contract (pp_query_result::n_lattice_boundary_hits > 0 marks the queries it concerns: some child's
 * DiscretizePose quotient lay within 1e-9 cells of a lattice line, where a last-bit difference between this libm and glibc could
 * choose the other cell).  The host recomputes every created constant-steer node and every LOGGED lattice-line child of query q with
 * glibc along its ancestors (ConstantSteer, models/kinematic_bicycle_model.cpp:5-32) and discretises it (DiscretizePose,
 * algo/hybrid_a_star.h:104-111).  n_cell_mismatches: cells that differ from the device's; n_unverified: flagged events that cannot
 * be recomputed (a Reeds-Shepp child on a lattice line, log entries beyond 64).  Both 0: the query's discrete outputs are certified
 * against the reference's arithmetic; else hand it to the CPU reference.  One-query-per-wave planners only (max_batch <= 64): they
 * keep the tree and the log; flagged queries of a throughput planner / pipeline are re-planned there first (identical results). */
int pp_planner_certify_lattice(pp_planner* planner, int32_t q, int32_t* n_checked, int32_t* n_cell_mismatches, int32_t* n_unverified, double* max_pose_difference);

/* ---- streaming form of HybridAStar::SearchPath's search stage (algo/hybrid_a_star.cpp:237-257) ------------------------------
 * One pipeline per GPU: `capacity` queries in flight (a field slot each: the obstacle-heuristic field of its goal, start / goal /
 * seed, path and Reeds-Shepp log), ObstaclesHeuristic::Update by the wavefront kernel, which hands every finished field to ONE
 * persistent search grid of `search_rows` rows (0 = 4096) through a device-side queue; a row takes the next ready query as soon as
 * its own ends, slots are recycled as results are polled.  No batch boundary: a query that exhausts the lattice (~1 s) holds one
 * row, not a batch's 17 GB of fields.  Results per query are exactly those of pp_planner_search_batch (same kernels' device code).
 * log_expansions != 0 keeps the expansion log per slot (parity tests; 4 B x max_nodes_per_query per slot).
 * 4 <= capacity < 2^20.  The ORDER in which fields are built is the pipeline's own (never a result): queries with a start or goal pose
 * closer than twice the validator's minimum safe radius to an obstacle or to the edge of the state space -- the ones that tend to
 * exhaust the lattice and end a run -- are built ahead of the submissions queued before them (PP_PIPE_URGENT_CLEARANCE=<m>, 0 = in
 * order of submission).  Results arrive in completion order either way.
 * Not thread-safe per pipeline.  Set GPU_MAX_HW_QUEUES >= 16 before the HIP runtime starts: the pipeline's long launches run on
 * streams of their own, and two streams that share a hardware queue serialise.  pp_pipeline_create CHECKS this (an oversubscribed
 * probe launch on every long-launch stream, a tiny kernel on every other one) and returns PP_ERR_INVALID with a message that names
 * the variable when a stream had to wait; PP_PIPE_ALLOW_SHARED_QUEUES=1 runs regardless (idle waves then leave after PP_PIPE_IDLE_MS). */
typedef struct pp_pipeline pp_pipeline;
int pp_pipeline_create(pp_map* map, const pp_hybrid_params* params, int32_t capacity, int32_t max_nodes_per_query, int32_t search_rows, int32_t log_expansions,
	pp_pipeline** out);
int pp_pipeline_destroy(pp_pipeline* pipeline);
/* Takes up to n_queries queries (as many as there are free slots: *n_accepted; submit the rest after a poll has released slots).
 * tickets_out (may be NULL): one id per accepted query, in input order.  The input arrays are free when the call returns.
 * The map as the kernels see it (bounds, grid pointers, pp_map_set_validator's tunables) is fixed per launch of the persistent search grid: a
 * submission after it changed is refused (PP_ERR_INVALID) while queries are in flight and accepted once they have been polled (the old grid's
 * waves are waited for).  Changing the CONTENTS of the map's grids with queries in flight is the caller's to avoid. */
int pp_pipeline_submit_dev(pp_pipeline* pipeline, int32_t n_queries, const double* starts_dev, const double* goals_dev, const uint64_t* seeds_dev, uint64_t* tickets_out,
	int32_t* n_accepted);
int pp_pipeline_submit(pp_pipeline* pipeline, int32_t n_queries, const double* starts_host, const double* goals_host, const uint64_t* seeds_host, uint64_t* tickets_out,
	int32_t* n_accepted);
/* Completed queries, at most max_results, in completion order; never blocks.  release != 0: their slots are free again at once;
 * release == 0: a slot stays held (pp_pipeline_slot_of + the pp_planner_get_path / get_expanded accessors of pp_pipeline_planner
 * read its path) until pp_pipeline_release. */
int pp_pipeline_poll(pp_pipeline* pipeline, int32_t max_results, uint64_t* tickets_out, pp_query_result* results_out, int32_t release, int32_t* n_out);
int pp_pipeline_release(pp_pipeline* pipeline, int32_t n, const uint64_t* tickets);
/* The plans themselves: GetGraphSearchPath (hybrid_a_star.h:223, a_star.h:254-288) of n completed, held queries (polled with release == 0), start
 * pose first.  poses_host: [n][max_poses][3] doubles (x, y, theta), n_poses_host[i] = nodes on path i (0: no solution; it may exceed max_poses,
 * then only the first max_poses are written).  The row that finishes a query writes its path's poses (up to PP_PIPE_PATH_POSES = 192 per query)
 * into a ring in pinned host memory together with the completion record, so this call copies host memory: no kernel, no device copy (a longer
 * path's remaining records are fetched from the device).  release != 0: the slots are returned like pp_pipeline_release. */
int pp_pipeline_get_paths(pp_pipeline* pipeline, int32_t n, const uint64_t* tickets, int32_t max_poses, double* poses_host, int32_t* n_poses_host, int32_t release);
int pp_pipeline_slot_of(pp_pipeline* pipeline, uint64_t ticket); /* -1 unless completed and held */
/* Diagnostics: waves of the search grid that are alive right now (a blocking device-to-host copy; -1 on error). */
int pp_pipeline_alive_waves(pp_pipeline* pipeline);
pp_planner* pp_pipeline_planner(pp_pipeline* pipeline);           /* the buffer set: set_nonholo_table, set_primitives, get_path(slot), ... */
int pp_pipeline_capacity(pp_pipeline* pipeline);
int pp_pipeline_search_rows(pp_pipeline* pipeline);
int pp_pipeline_in_flight(pp_pipeline* pipeline);  /* submitted and not yet polled */
int pp_pipeline_free_slots(pp_pipeline* pipeline);
/* Where the queries in flight are, as the row that announced the latest polled result saw the queue's counters (they ride in every
 * completion record): ready = fields built, waiting for a search row; searching = claimed by a row, not yet polled.  A ready queue near 0 = the wavefront stage is the bottleneck, a long one = the search grid is. */
int pp_pipeline_backlog(pp_pipeline* pipeline, int64_t* ready, int64_t* searching);
/* Kernel launch durations since the last call (HIP events on the launching streams), then reset: total milliseconds / launches / goals of
 * the wavefront kernel; total milliseconds / launches of the search grid and its longest launch (a launch that tops up a full grid lasts
 * microseconds, the one that starts it lasts as long as there is work). */
int pp_pipeline_timings(pp_pipeline* pipeline, double* wavefront_ms_total, int64_t* wavefront_launches, int64_t* wavefront_goals, double* search_ms_total,
	int64_t* search_launches, double* search_max_ms);

/* last batch: milliseconds spent in the wavefront kernel and in the search kernel (HIP events) */
int pp_planner_last_timings(pp_planner* planner, float* wavefront_ms, float* search_ms);
/* Diagnostics (never on by default): launch the stamped build of the search kernel and read, per query,
 * 8 shader-clock sums {pop, node load, parent heuristic, children, duplicate scan, insertion, node write, RS}. */
int pp_planner_set_profiling(pp_planner* planner, int32_t enable);
int pp_planner_phase_cycles(pp_planner* planner, int32_t n_queries, uint64_t* cycles_host);

/* ---- a14: Tree::GetNearestNodes (utils/tree.h:73-116; flann exact kNN) ------
 * points/queries: {x, y} doubles; idx/d2: [n_queries][k], ascending squared L2,
 * ties -> lower point index; slots beyond n_points hold -1 / +inf. */
int pp_knn_dev(pp_ctx* ctx, int64_t n_points, const double* points_dev, int64_t n_queries, const double* queries_dev, int32_t k,
	int32_t* idx_dev, double* d2_dev);
int pp_knn(pp_ctx* ctx, int64_t n_points, const double* points_host, int64_t n_queries, const double* queries_host, int32_t k,
	int32_t* idx_host, double* d2_host);

/* ---- a13: RRT / RRT* (algo/rrt.h:55-95, algo/rrt_star.h:53-112) ------------
 * The whole loop runs on the device, one workgroup per tree.  map == NULL: free space
 * (StateValidatorFree).  params = {maxIteration, maxNumberTreeNode, maxConnectionDistance, goalBias[, gamma]}.
 * star: 0 = RRT, 1 = RRT* exactly as the reference (choose-parent among the k = ln N nearest, NO rewire: rrt_star.h:83-97);
 * beyond the reference: 2 = + rewire of the near nodes through the new node with the saving carried down
 * their subtrees (what utils/node.h:203-225 Reparent is for), 3 = the same with a radius near-set, the <= 16 nearest nodes
 * within params[4] * sqrt(ln(n + 1) / (n + 1)). */
typedef struct pp_rrt_result {
	int32_t status;
	int32_t n_nodes;
	int32_t n_path;
	int64_t iterations, n_knn_queries, n_edge_checks;
} pp_rrt_result;
typedef struct pp_rrt pp_rrt;
int pp_rrt_run(pp_ctx* ctx, pp_map* map, const double lower[2], const double upper[2], const double params[4], const double init[2],
	const double goal[2], uint64_t seed, int32_t star, pp_rrt** out, pp_rrt_result* result);
/* n independent problems (own start, goal, seed) on the same map and parameters, one workgroup each, run together. */
int pp_rrt_run_batch(pp_ctx* ctx, pp_map* map, const double lower[2], const double upper[2], const double params[4], int32_t n_problems,
	const double* inits_xy, const double* goals_xy, const uint64_t* seeds, int32_t star, pp_rrt** outs, pp_rrt_result* results);
int pp_rrt_get(pp_rrt* r, double* nodes_xy, int32_t* parents, double* costs, double* path_xy);
int pp_rrt_destroy(pp_rrt* r);

/* ---- a12 / f4: AStarN2 and BidirectionalAStarN2 as a batch on the device ------
 * Engine: algo/a_star.h:326-427 (SearchPath, Expand, ProcessPossibleShortcut; open list order utils/frontier.h:39-48,83-91);
 * propagator: AStarStatePropagatorFcnN2::GetNeighborStates (algo/a_star_n2.cpp:12-28) over the map's occupancy grid; the user
 * functions the reference takes (transition cost, heuristic) are fixed to the ones of its own script
 * (interfaces/python/scripts/example_a_star_grid.py:46-52): Euclidean distance between the two cells, in double.  Other
 * functions stay on the host engine (pathplanning_amd/host/a_star.hpp), which calls back into the caller per edge as the
 * reference does.  bidirectional != 0: BidirectionalAStar::SearchPath (algo/bidirectional_a_star.h:130-196) with the
 * AverageHeuristic pair (:10-39,58-63); inner_goals[q] = {forward heuristic's goal (row, col), reverse heuristic's goal} are the
 * goals held by the two wrapped heuristic objects, which AverageHeuristic never updates (NULL: forward -> goal, reverse -> init).
 * cells are (row, col) pairs.  results[q].status: 0 success, -1 failure (open list empty), -2 open list beyond its workspace
 * (the call then returns PP_ERR_CAPACITY).  paths: [n][max_path][2], root .. goal; in bidirectional mode the meeting cell
 * appears twice as in the reference (GetPath, :66-72); n_path may exceed max_path (then only the first max_path cells are stored).
 * expanded / expanded_reverse (optional): [n][max_expanded][2], states in expansion order = GetExploredStates as a sequence
 * (the root is expanded first; counts may exceed max_expanded likewise). */
typedef struct pp_grid_result {
	int32_t status;
	int32_t n_path;
	int32_t n_expanded;
	int32_t n_expanded_reverse;
	double cost; /* GetOptimalCost; +inf without a path */
} pp_grid_result;
int pp_grid_astar_batch(pp_map* map, int32_t n_queries, const int32_t* init_cells, const int32_t* goal_cells, int32_t bidirectional,
	const int32_t* inner_goals, int32_t max_path, int32_t max_expanded, pp_grid_result* results, int32_t* paths, int32_t* expanded,
	int32_t* expanded_reverse);

#ifdef __cplusplus
}
#endif
#endif /* PP_HIP_H */
