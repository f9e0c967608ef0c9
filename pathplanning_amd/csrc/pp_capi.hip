// C ABI of libpphip.so: context, map set, and the batched per-pose / per-path entry points.
// (planner: pp_planner.hip; RRT: pp_rrt.hip)
#include "pp_internal.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>

namespace pph {

static thread_local std::string g_lastError;

void set_error(const std::string& msg) { g_lastError = msg; }
int hip_fail(hipError_t e, const char* what)
{
	g_lastError = std::string(what) + ": " + hipGetErrorString(e);
	return e == hipErrorNoDevice ? PP_ERR_NO_DEVICE : PP_ERR_HIP;
}

} // namespace pph

using namespace pph;

ppd::MapView pp_map::view() const
{
	ppd::MapView v;
	v.rows = desc.rows;
	v.cols = desc.cols;
	v.res = desc.resolution;
	v.invRes = 1.0 / (double)desc.resolution;
	v.gx = desc.grid_origin[0];
	v.gy = desc.grid_origin[1];
	v.lox = desc.local_origin[0];
	v.loy = desc.local_origin[1];
	v.lbx = desc.lower[0];
	v.lby = desc.lower[1];
	v.lbt = desc.lower[2];
	v.ubx = desc.upper[0];
	v.uby = desc.upper[1];
	v.ubt = desc.upper[2];
	v.minSafeRadius = minSafeRadius;
	v.minInterp = minInterp;
	v.dist = dist;
	v.pathcost = pathcost;
	v.occ8 = occ8;
	v.validBits = validBits;
	return v;
}

int pph::refresh_occupancy_views(pp_map* map, hipStream_t s)
{
	const size_t n = map->cells();
	if (!map->occ8)
		PP_HIP_TRY(hipMalloc((void**)&map->occ8, n));
	int wpr = 0, nWordRows = 0;
	pph::occ_bits_dims(map->desc.rows, map->desc.cols, wpr, nWordRows);
	if (!map->occBits)
		PP_HIP_TRY(hipMalloc((void**)&map->occBits, (size_t)wpr * nWordRows * 8));
	PP_HIP_TRY(pph::launch_occ_to_u8(s, map->occ32, map->occ8, (int64_t)n));
	PP_HIP_TRY(pph::launch_occ_bits(s, map->occ8, map->desc.rows, map->desc.cols, map->occBits));
	return PP_OK;
}

namespace {

/// RAII device scratch for the host-pointer convenience entry points.
struct DevBuf {
	void* p = nullptr;
	~DevBuf()
	{
		if (p)
			(void)hipFree(p);
	}
	hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
	template <typename T>
	T* as() { return (T*)p; }
};

int check_map(pp_map* map, bool needDist)
{
	if (!map) {
		set_error("null map");
		return PP_ERR_INVALID;
	}
	if (needDist && !map->dist) {
		set_error("distance grid not uploaded (pp_map_upload_dist2)");
		return PP_ERR_INVALID;
	}
	return PP_OK;
}

} // namespace

namespace pph {
void ctx_release(pp_ctx* ctx)
{
	if (!ctx || __atomic_sub_fetch(&ctx->refs, 1, __ATOMIC_ACQ_REL) > 0)
		return;
	(void)hipSetDevice(ctx->device);
	if (ctx->ev0)
		(void)hipEventDestroy(ctx->ev0);
	if (ctx->ev1)
		(void)hipEventDestroy(ctx->ev1);
	if (ctx->ownsStream && ctx->stream)
		(void)hipStreamDestroy(ctx->stream);
	delete ctx;
}
void map_release(pp_map* map)
{
	if (!map || __atomic_sub_fetch(&map->refs, 1, __ATOMIC_ACQ_REL) > 0)
		return;
	(void)hipSetDevice(map->ctx->device);
	(void)hipStreamSynchronize(map->ctx->stream);
	if (map->d2)
		(void)hipFree(map->d2);
	if (map->dist)
		(void)hipFree(map->dist);
	if (map->pathcost)
		(void)hipFree(map->pathcost);
	if (map->occ8)
		(void)hipFree(map->occ8);
	if (map->occBits)
		(void)hipFree(map->occBits);
	if (map->validBits)
		(void)hipFree(map->validBits);
	void* more[] = { map->occ32, map->obstLabel[0], map->obstLabel[1], map->voroLabel[0], map->voroLabel[1], map->voroD2, map->voroEdge, map->gvdFlag };
	for (void* q : more)
		if (q)
			(void)hipFree(q);
	gvd_reference_free(map);
	pp_ctx* ctx = map->ctx;
	delete map;
	ctx_release(ctx);
}
} // namespace pph

extern "C" {

const char* pp_last_error(void) { return g_lastError.c_str(); }
int pp_version(void) { return 100; }

int pp_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

int pp_ctx_create(int device, void* stream, pp_ctx** out)
{
	if (!out) {
		set_error("null out");
		return PP_ERR_INVALID;
	}
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n == 0) {
		set_error("no HIP device visible: libpphip has no CPU fallback");
		return PP_ERR_NO_DEVICE;
	}
	if (device < 0 || device >= n) {
		set_error("device index out of range");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(device));
	auto ctx = std::make_unique<pp_ctx>();
	ctx->device = device;
	if (stream) {
		ctx->stream = (hipStream_t)stream;
		ctx->ownsStream = false;
	} else {
		PP_HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
		ctx->ownsStream = true;
	}
	PP_HIP_TRY(hipEventCreate(&ctx->ev0));
	PP_HIP_TRY(hipEventCreate(&ctx->ev1));
	*out = ctx.release();
	return PP_OK;
}

int pp_ctx_destroy(pp_ctx* ctx)
{
	pph::ctx_release(ctx);
	return PP_OK;
}

int pp_ctx_synchronize(pp_ctx* ctx)
{
	if (!ctx) {
		set_error("null ctx");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipStreamSynchronize(ctx->stream));
	return PP_OK;
}

int pp_ctx_is_idle(pp_ctx* ctx, int32_t* idle)
{
	if (!ctx || !idle) {
		set_error("null ctx");
		return PP_ERR_INVALID;
	}
	const hipError_t e = hipStreamQuery(ctx->stream);
	if (e != hipSuccess && e != hipErrorNotReady)
		return pph::hip_fail(e, "hipStreamQuery");
	*idle = e == hipSuccess ? 1 : 0;
	return PP_OK;
}

int pp_ctx_timer_start(pp_ctx* ctx)
{
	if (!ctx) {
		set_error("null ctx");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
	return PP_OK;
}

int pp_ctx_timer_stop(pp_ctx* ctx, float* ms)
{
	if (!ctx || !ms) {
		set_error("null argument");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
	PP_HIP_TRY(hipEventSynchronize(ctx->ev1));
	PP_HIP_TRY(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
	return PP_OK;
}

// ------------------------------------------------------------------ map ----
int pp_map_create(pp_ctx* ctx, const pp_map_desc* desc, pp_map** out)
{
	if (!ctx || !desc || !out) {
		set_error("null argument");
		return PP_ERR_INVALID;
	}
	if (desc->rows <= 0 || desc->cols <= 0 || !(desc->resolution > 0.0f)) {
		set_error("invalid grid size: received " + std::to_string(desc->rows) + " x " + std::to_string(desc->cols)); // utils/grid.h:69-72
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(ctx->device));
	auto m = std::make_unique<pp_map>();
	m->ctx = ctx;
	__atomic_add_fetch(&ctx->refs, 1, __ATOMIC_RELAXED); // the map keeps its context alive
	m->desc = *desc;
	*out = m.release();
	return PP_OK;
}

int pp_map_destroy(pp_map* map)
{
	pph::map_release(map);
	return PP_OK;
}

int pp_map_upload_dist2(pp_map* map, const int32_t* d2_host)
{
	if (check_map(map, false) || !d2_host) {
		set_error("null argument");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	const size_t n = map->cells();
	if (!map->d2)
		PP_HIP_TRY(hipMalloc((void**)&map->d2, n * sizeof(int32_t)));
	if (!map->dist)
		PP_HIP_TRY(hipMalloc((void**)&map->dist, n * sizeof(float)));
	PP_HIP_TRY(hipMemcpyAsync(map->d2, d2_host, n * sizeof(int32_t), hipMemcpyHostToDevice, map->ctx->stream));
	PP_HIP_TRY(launch_d2_to_distance(map->ctx->stream, map->d2, map->dist, (int64_t)n, map->desc.resolution));
	if (!map->validBits)
		PP_HIP_TRY(hipMalloc((void**)&map->validBits, ((n + 63) / 64) * 8)); // whole 64-cell groups: one ballot each
	PP_HIP_TRY(launch_valid_bits(map->ctx->stream, map->dist, (int64_t)n, map->minSafeRadius, map->validBits));
	PP_HIP_TRY(hipStreamSynchronize(map->ctx->stream));
	return PP_OK;
}

int pp_map_upload_occupancy(pp_map* map, const int32_t* occ_host)
{
	if (check_map(map, false) || !occ_host) {
		set_error("null argument");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	const size_t n = map->cells();
	if (!map->occ32) // kept: pp_map_update_gvd builds the distance / Voronoi / path-cost fields from it
		PP_HIP_TRY(hipMalloc((void**)&map->occ32, n * sizeof(int32_t)));
	PP_HIP_TRY(hipMemcpyAsync(map->occ32, occ_host, n * sizeof(int32_t), hipMemcpyHostToDevice, map->ctx->stream));
	// for the reference-order field update (pp_map_update_gvd_ex) a whole-grid upload is "an empty map, then every occupied cell in
	// row-major order": the reference has no such entry point (its cells only arrive through AddObstacle), so this order is ours
	map->journal.clear();
	map->journalLost = false;
	map->journalReset = true;
	for (size_t i = 0; i < n; i++)
		if (occ_host[i] >= 0) {
			map->journal.push_back((int32_t)i);
			map->journal.push_back(occ_host[i]);
		}
	if (int rc = pph::refresh_occupancy_views(map, map->ctx->stream))
		return rc;
	PP_HIP_TRY(hipStreamSynchronize(map->ctx->stream));
	return PP_OK;
}

int pp_map_upload_path_cost(pp_map* map, const float* cost_host)
{
	if (check_map(map, false) || !cost_host) {
		set_error("null argument");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	const size_t n = map->cells();
	if (!map->pathcost)
		PP_HIP_TRY(hipMalloc((void**)&map->pathcost, n * sizeof(float)));
	PP_HIP_TRY(hipMemcpyAsync(map->pathcost, cost_host, n * sizeof(float), hipMemcpyHostToDevice, map->ctx->stream));
	PP_HIP_TRY(hipStreamSynchronize(map->ctx->stream));
	return PP_OK;
}

int pp_map_set_validator(pp_map* map, float min_safe_radius, float min_path_interpolation_distance)
{
	if (check_map(map, false))
		return PP_ERR_INVALID;
	const bool radiusChanged = !(map->minSafeRadius == min_safe_radius);
	map->minSafeRadius = min_safe_radius;
	map->minInterp = min_path_interpolation_distance;
	if (radiusChanged && map->dist && map->validBits) { // the validity bitmap is the comparison against this radius
		PP_HIP_TRY(hipSetDevice(map->ctx->device));
		PP_HIP_TRY(launch_valid_bits(map->ctx->stream, map->dist, (int64_t)map->cells(), map->minSafeRadius, map->validBits));
	}
	return PP_OK;
}

int pp_map_download_distance(pp_map* map, float* dist_host)
{
	if (check_map(map, true) || !dist_host)
		return PP_ERR_INVALID;
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	PP_HIP_TRY(hipMemcpyAsync(dist_host, map->dist, map->cells() * sizeof(float), hipMemcpyDeviceToHost, map->ctx->stream));
	PP_HIP_TRY(hipStreamSynchronize(map->ctx->stream));
	return PP_OK;
}

// --------------------------------------------------------- check_states ----
int pp_check_states_dev(pp_map* map, int64_t n, const double* poses_dev, uint8_t* valid_dev)
{
	if (int rc = check_map(map, true))
		return rc;
	if (n < 0 || (n > 0 && (!poses_dev || !valid_dev))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(launch_check_states(map->ctx->stream, map->view(), n, poses_dev, valid_dev));
	return PP_OK;
}

int pp_check_states(pp_map* map, int64_t n, const double* poses_host, uint8_t* valid_host)
{
	if (int rc = check_map(map, true))
		return rc;
	if (n < 0 || (n > 0 && (!poses_host || !valid_host))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	if (n == 0)
		return PP_OK;
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipStream_t s = map->ctx->stream;
	DevBuf dp, dv;
	PP_HIP_TRY(dp.alloc((size_t)n * 24));
	PP_HIP_TRY(dv.alloc((size_t)n));
	PP_HIP_TRY(hipMemcpyAsync(dp.p, poses_host, (size_t)n * 24, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(launch_check_states(s, map->view(), n, dp.as<double>(), dv.as<uint8_t>()));
	PP_HIP_TRY(hipMemcpyAsync(valid_host, dv.p, (size_t)n, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipStreamSynchronize(s));
	return PP_OK;
}

int pp_check_states_fused_dev(pp_map* map, int64_t n, uint64_t seed, uint64_t* valid_count_dev)
{
	if (int rc = check_map(map, true))
		return rc;
	if (n < 0 || !valid_count_dev) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(launch_check_states_fused(map->ctx->stream, map->view(), n, seed, valid_count_dev));
	return PP_OK;
}

// ----------------------------------------------------------- check_arcs ----
int pp_check_arcs_dev(pp_map* map, int64_t n, const double* from_dev, const double* curvature_dev, const double* length_dev, const int32_t* direction_dev,
	uint8_t* valid_dev, float* last_ratio_dev)
{
	if (int rc = check_map(map, true))
		return rc;
	if (n < 0 || (n > 0 && (!from_dev || !curvature_dev || !length_dev || !direction_dev || !valid_dev))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(launch_check_arcs(map->ctx->stream, map->view(), n, from_dev, curvature_dev, length_dev, direction_dev, valid_dev, last_ratio_dev));
	return PP_OK;
}

int pp_check_arcs(pp_map* map, int64_t n, const double* from_host, const double* curvature_host, const double* length_host, const int32_t* direction_host,
	uint8_t* valid_host, float* last_ratio_host)
{
	if (int rc = check_map(map, true))
		return rc;
	if (n < 0 || (n > 0 && (!from_host || !curvature_host || !length_host || !direction_host || !valid_host))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	if (n == 0)
		return PP_OK;
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipStream_t s = map->ctx->stream;
	DevBuf df, dk, dl, dd, dv, dr;
	PP_HIP_TRY(df.alloc((size_t)n * 24));
	PP_HIP_TRY(dk.alloc((size_t)n * 8));
	PP_HIP_TRY(dl.alloc((size_t)n * 8));
	PP_HIP_TRY(dd.alloc((size_t)n * 4));
	PP_HIP_TRY(dv.alloc((size_t)n));
	PP_HIP_TRY(dr.alloc((size_t)n * 4));
	PP_HIP_TRY(hipMemcpyAsync(df.p, from_host, (size_t)n * 24, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(hipMemcpyAsync(dk.p, curvature_host, (size_t)n * 8, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(hipMemcpyAsync(dl.p, length_host, (size_t)n * 8, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(hipMemcpyAsync(dd.p, direction_host, (size_t)n * 4, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(launch_check_arcs(s, map->view(), n, df.as<double>(), dk.as<double>(), dl.as<double>(), dd.as<int32_t>(), dv.as<uint8_t>(), dr.as<float>()));
	PP_HIP_TRY(hipMemcpyAsync(valid_host, dv.p, (size_t)n, hipMemcpyDeviceToHost, s));
	if (last_ratio_host)
		PP_HIP_TRY(hipMemcpyAsync(last_ratio_host, dr.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipStreamSynchronize(s));
	return PP_OK;
}

int pp_check_segments_dev(pp_map* map, int64_t n, const double* from_xy_dev, const double* to_xy_dev, uint8_t* valid_dev)
{
	if (int rc = check_map(map, true))
		return rc;
	if (n < 0 || (n > 0 && (!from_xy_dev || !to_xy_dev || !valid_dev))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(launch_check_segments(map->ctx->stream, map->view(), n, from_xy_dev, to_xy_dev, valid_dev));
	return PP_OK;
}

int pp_check_segments(pp_map* map, int64_t n, const double* from_xy_host, const double* to_xy_host, uint8_t* valid_host)
{
	if (int rc = check_map(map, true))
		return rc;
	if (n < 0 || (n > 0 && (!from_xy_host || !to_xy_host || !valid_host))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	if (n == 0)
		return PP_OK;
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipStream_t s = map->ctx->stream;
	DevBuf df, dt, dv;
	PP_HIP_TRY(df.alloc((size_t)n * 16));
	PP_HIP_TRY(dt.alloc((size_t)n * 16));
	PP_HIP_TRY(dv.alloc((size_t)n));
	PP_HIP_TRY(hipMemcpyAsync(df.p, from_xy_host, (size_t)n * 16, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(hipMemcpyAsync(dt.p, to_xy_host, (size_t)n * 16, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(launch_check_segments(s, map->view(), n, df.as<double>(), dt.as<double>(), dv.as<uint8_t>()));
	PP_HIP_TRY(hipMemcpyAsync(valid_host, dv.p, (size_t)n, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipStreamSynchronize(s));
	return PP_OK;
}

// -------------------------------------------------------------- rollout ----
static int make_rollout(pp_map* map, const pp_hybrid_params* params, int32_t n_primitives, const double* curvature_host, const int32_t* direction_host,
	RolloutParams& rp, PrimTable& pt)
{
	if (!params || !curvature_host || !direction_host || n_primitives < 1 || n_primitives > kMaxPrimitives) {
		set_error("invalid primitive table (1.." + std::to_string(kMaxPrimitives) + " primitives)");
		return PP_ERR_INVALID;
	}
	if (!map->pathcost) {
		set_error("path-cost grid not uploaded (pp_map_upload_path_cost)");
		return PP_ERR_INVALID;
	}
	rp.arcLength = params->spatial_resolution * 1.5; // hybrid_a_star.cpp:115
	rp.spatialRes = params->spatial_resolution;
	rp.angularRes = params->angular_resolution;
	rp.lat.set(params->spatial_resolution, params->angular_resolution);
	rp.forwardMult = params->forward_cost_multiplier;
	rp.reverseMult = params->reverse_cost_multiplier;
	rp.voronoiMult = params->voronoi_cost_multiplier;
	rp.voroDiagRes = (float)(map->desc.resolution * std::sqrt(2.0)); // hybrid_a_star.cpp:38
	rp.headingAlias = params->heading_alias;
	pt.n = n_primitives;
	for (int i = 0; i < n_primitives; i++) {
		pt.kappa[i] = curvature_host[i];
		pt.invKappa[i] = curvature_host[i] != 0.0 ? 1 / curvature_host[i] : 0.0;
		pt.backward[i] = direction_host[i] == 1;
	}
	return PP_OK;
}

int pp_rollout_children_dev(pp_map* map, const pp_hybrid_params* params, int32_t n_primitives, const double* curvature_host, const int32_t* direction_host,
	int64_t n_parents, const double* parents_dev, uint8_t* valid_dev, double* pose_dev, int32_t* key_dev, double* cost_dev, double* length_dev)
{
	if (int rc = check_map(map, true))
		return rc;
	RolloutParams rp;
	PrimTable pt;
	if (int rc = make_rollout(map, params, n_primitives, curvature_host, direction_host, rp, pt))
		return rc;
	if (n_parents < 0 || (n_parents > 0 && (!parents_dev || !valid_dev || !pose_dev || !key_dev || !cost_dev || !length_dev))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(launch_rollout(map->ctx->stream, map->view(), rp, pt, n_parents, parents_dev, valid_dev, pose_dev, key_dev, cost_dev, length_dev));
	return PP_OK;
}

int pp_rollout_children(pp_map* map, const pp_hybrid_params* params, int32_t n_primitives, const double* curvature_host, const int32_t* direction_host,
	int64_t n_parents, const double* parents_host, uint8_t* valid_host, double* pose_host, int32_t* key_host, double* cost_host, double* length_host)
{
	if (int rc = check_map(map, true))
		return rc;
	RolloutParams rp;
	PrimTable pt;
	if (int rc = make_rollout(map, params, n_primitives, curvature_host, direction_host, rp, pt))
		return rc;
	if (n_parents <= 0)
		return n_parents == 0 ? PP_OK : PP_ERR_INVALID;
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipStream_t s = map->ctx->stream;
	const size_t total = (size_t)n_parents * n_primitives;
	DevBuf dp, dv, dpose, dkey, dcost, dlen;
	PP_HIP_TRY(dp.alloc((size_t)n_parents * 24));
	PP_HIP_TRY(dv.alloc(total));
	PP_HIP_TRY(dpose.alloc(total * 24));
	PP_HIP_TRY(dkey.alloc(total * 12));
	PP_HIP_TRY(dcost.alloc(total * 8));
	PP_HIP_TRY(dlen.alloc(total * 8));
	PP_HIP_TRY(hipMemcpyAsync(dp.p, parents_host, (size_t)n_parents * 24, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(launch_rollout(s, map->view(), rp, pt, n_parents, dp.as<double>(), dv.as<uint8_t>(), dpose.as<double>(), dkey.as<int32_t>(), dcost.as<double>(),
		dlen.as<double>()));
	PP_HIP_TRY(hipMemcpyAsync(valid_host, dv.p, total, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipMemcpyAsync(pose_host, dpose.p, total * 24, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipMemcpyAsync(key_host, dkey.p, total * 12, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipMemcpyAsync(cost_host, dcost.p, total * 8, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipMemcpyAsync(length_host, dlen.p, total * 8, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipStreamSynchronize(s));
	return PP_OK;
}

// ------------------------------------------------------------- rs_solve ----
int pp_rs_solve_dev(pp_ctx* ctx, int64_t n, const double* from_dev, const double* to_dev, double min_turning_radius, float reverse_cost, float forward_cost,
	float switch_cost, int32_t* word_dev, double* tuv_dev, float* cost_dev, double* seg_length_dev)
{
	if (!ctx || n < 0 || (n > 0 && (!from_dev || !to_dev || !word_dev))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(launch_rs_solve(ctx->stream, n, from_dev, to_dev, min_turning_radius, reverse_cost, forward_cost, switch_cost, word_dev, tuv_dev, cost_dev,
		seg_length_dev));
	return PP_OK;
}

int pp_rs_solve(pp_ctx* ctx, int64_t n, const double* from_host, const double* to_host, double min_turning_radius, float reverse_cost, float forward_cost,
	float switch_cost, int32_t* word_host, double* tuv_host, float* cost_host, double* seg_length_host)
{
	if (!ctx || n < 0 || (n > 0 && (!from_host || !to_host || !word_host))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	if (n == 0)
		return PP_OK;
	PP_HIP_TRY(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	DevBuf df, dt, dw, dtuv, dc, dl;
	PP_HIP_TRY(df.alloc((size_t)n * 24));
	PP_HIP_TRY(dt.alloc((size_t)n * 24));
	PP_HIP_TRY(dw.alloc((size_t)n * 4));
	PP_HIP_TRY(dtuv.alloc((size_t)n * 24));
	PP_HIP_TRY(dc.alloc((size_t)n * 4));
	PP_HIP_TRY(dl.alloc((size_t)n * 8));
	PP_HIP_TRY(hipMemcpyAsync(df.p, from_host, (size_t)n * 24, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(hipMemcpyAsync(dt.p, to_host, (size_t)n * 24, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(launch_rs_solve(s, n, df.as<double>(), dt.as<double>(), min_turning_radius, reverse_cost, forward_cost, switch_cost, dw.as<int32_t>(),
		dtuv.as<double>(), dc.as<float>(), dl.as<double>()));
	PP_HIP_TRY(hipMemcpyAsync(word_host, dw.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
	if (tuv_host)
		PP_HIP_TRY(hipMemcpyAsync(tuv_host, dtuv.p, (size_t)n * 24, hipMemcpyDeviceToHost, s));
	if (cost_host)
		PP_HIP_TRY(hipMemcpyAsync(cost_host, dc.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
	if (seg_length_host)
		PP_HIP_TRY(hipMemcpyAsync(seg_length_host, dl.p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipStreamSynchronize(s));
	return PP_OK;
}

// -------------------------------------------------------------- nonholo ----
static int nonholo_desc(const double lower[3], const double upper[3], const pp_hybrid_params* p, NonHoloDesc& d)
{
	if (!lower || !upper || !p || !(p->spatial_resolution > 0) || !(p->angular_resolution > 0)) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	// heuristics.cpp:43-51 (sizes) and :13-14 (numAngular, offsets)
	const double spatialSizeX = upper[0] - lower[0];
	const double spatialSizeY = upper[1] - lower[1];
	unsigned int nx = std::ceil(spatialSizeX / p->spatial_resolution);
	if (nx % 2 == 0)
		nx++;
	unsigned int ny = std::ceil(spatialSizeY / p->spatial_resolution);
	if (ny % 2 == 0)
		ny++;
	d.nx = (int)nx;
	d.ny = (int)ny;
	d.na = (int)std::ceil(2 * M_PI / p->angular_resolution);
	d.spatialRes = p->spatial_resolution;
	d.angularRes = p->angular_resolution;
	d.offX = std::floor(nx / 2.0) * p->spatial_resolution;
	d.offY = std::floor(ny / 2.0) * p->spatial_resolution;
	d.rmin = p->min_turning_radius;
	d.reverseCost = (float)p->reverse_cost_multiplier; // double -> float at the GetOptimalPath call (heuristics.cpp:67)
	d.forwardCost = (float)p->forward_cost_multiplier;
	d.switchCost = (float)p->direction_switching_cost;
	d.minMult = std::min(p->reverse_cost_multiplier, p->forward_cost_multiplier);
	d.negativeKRead = p->negative_k_read;
	return PP_OK;
}

int pp_nonholo_dims(const double lower[3], const double upper[3], const pp_hybrid_params* params, int32_t dims[3], double offsets[2])
{
	NonHoloDesc d;
	if (int rc = nonholo_desc(lower, upper, params, d))
		return rc;
	if (dims) {
		dims[0] = d.nx;
		dims[1] = d.ny;
		dims[2] = d.na;
	}
	if (offsets) {
		offsets[0] = d.offX;
		offsets[1] = d.offY;
	}
	return PP_OK;
}

int pp_nonholo_build_dev(pp_ctx* ctx, const double lower[3], const double upper[3], const pp_hybrid_params* params, double* table_dev)
{
	NonHoloDesc d;
	if (!ctx || !table_dev) {
		set_error("null argument");
		return PP_ERR_INVALID;
	}
	if (int rc = nonholo_desc(lower, upper, params, d))
		return rc;
	PP_HIP_TRY(launch_nonholo_build(ctx->stream, d, table_dev));
	return PP_OK;
}

int pp_nonholo_build(pp_ctx* ctx, const double lower[3], const double upper[3], const pp_hybrid_params* params, double* table_host)
{
	NonHoloDesc d;
	if (!ctx || !table_host) {
		set_error("null argument");
		return PP_ERR_INVALID;
	}
	if (int rc = nonholo_desc(lower, upper, params, d))
		return rc;
	PP_HIP_TRY(hipSetDevice(ctx->device));
	const size_t total = (size_t)d.nx * d.ny * d.na;
	DevBuf dt;
	PP_HIP_TRY(dt.alloc(total * 8));
	PP_HIP_TRY(launch_nonholo_build(ctx->stream, d, dt.as<double>()));
	PP_HIP_TRY(hipMemcpyAsync(table_host, dt.p, total * 8, hipMemcpyDeviceToHost, ctx->stream));
	PP_HIP_TRY(hipStreamSynchronize(ctx->stream));
	return PP_OK;
}

// ------------------------------------------------------------ wavefront ----
int64_t pp_obstacle_heuristic_workspace_bytes(pp_map* map)
{
	if (!map)
		return 0;
	return wavefront_workspace_bytes(map->desc.rows, map->desc.cols);
}

static int goal_cells(pp_map* map, int32_t n_goals, const double* goal_xy_host, std::vector<int32_t>& cells)
{
	// m_map->WorldPositionToGridCell(goal.position), bounded (heuristics.cpp:115-117)
	const pp_map_desc& d = map->desc;
	cells.resize(n_goals);
	for (int g = 0; g < n_goals; g++) {
		const double x = goal_xy_host[2 * g], y = goal_xy_host[2 * g + 1];
		const double fx = (x - d.grid_origin[0]) / d.resolution, fy = (y - d.grid_origin[1]) / d.resolution;
		int row = (fx > -2147483649.0 && fx < 2147483648.0) ? (int)fx : INT32_MIN;
		int col = (fy > -2147483649.0 && fy < 2147483648.0) ? (int)fy : INT32_MIN;
		bool inside = row >= 0 && row < d.rows && col >= 0 && col < d.cols;
		cells[g] = inside ? row * d.cols + col : -1;
	}
	return PP_OK;
}

int pp_obstacle_heuristic_dev(pp_map* map, int32_t n_goals, const double* goal_xy_host, float* cost_dev)
{
	if (check_map(map, false) || !map->occ8) {
		set_error("occupancy grid not uploaded (pp_map_upload_occupancy)");
		return PP_ERR_INVALID;
	}
	if (n_goals < 0 || (n_goals > 0 && (!goal_xy_host || !cost_dev))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	if (n_goals == 0)
		return PP_OK;
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipStream_t s = map->ctx->stream;
	std::vector<int32_t> cells;
	goal_cells(map, n_goals, goal_xy_host, cells);
	const int resident = wavefront_resident_blocks();
	int nSlots = n_goals < resident ? n_goals : resident;
	const int64_t wsb = wavefront_workspace_bytes(map->desc.rows, map->desc.cols);
	const bool tiles = map->occBits && wavefront_tiles_enabled() && wavefront_tiles_supported(map->desc.rows, map->desc.cols);
	if (tiles && nSlots > 16)
		nSlots = 16; // (the ordered kernel only takes the goals the tile form hands over)
	DevBuf ws, dc, derr, dtiles;
	PP_HIP_TRY(ws.alloc((size_t)wsb * nSlots));
	PP_HIP_TRY(dc.alloc((size_t)n_goals * 4));
	PP_HIP_TRY(derr.alloc(8));
	PP_HIP_TRY(dtiles.alloc(64 + (size_t)n_goals * 4));
	PP_HIP_TRY(hipMemsetAsync(derr.p, 0, 8, s));
	PP_HIP_TRY(hipMemsetAsync(dtiles.p, 0, 64, s));
	PP_HIP_TRY(hipMemcpyAsync(dc.p, cells.data(), (size_t)n_goals * 4, hipMemcpyHostToDevice, s));
	WavefrontPublish pub;
	if (tiles) {
		pub.tilesCtl = dtiles.as<int>();
		pub.tilesFallback = dtiles.as<int32_t>() + 16;
		pub.occBits = map->occBits;
	}
	PP_HIP_TRY(launch_wavefront(s, map->view(), n_goals, dc.as<int32_t>(), cost_dev, ws.p, wsb, nSlots, derr.as<int32_t>(), nullptr, false, nullptr, false, nullptr, nullptr, nullptr,
		nullptr, pub));
	int32_t err = 0;
	PP_HIP_TRY(hipMemcpyAsync(&err, derr.p, 4, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipStreamSynchronize(s));
	if (err) {
		set_error("obstacle-heuristic open list exceeded its workspace");
		return PP_ERR_CAPACITY;
	}
	return PP_OK;
}

int pp_obstacle_heuristic_tiles_stats(pp_map* map, int32_t n_goals, const double* goal_xy_host, float* cost_dev, uint64_t stats_host[16], float* ms_out, int32_t* handed_over_host)
{
	if (check_map(map, false) || !map->occ8 || n_goals <= 0 || !goal_xy_host || !cost_dev || !stats_host) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	if (!map->occBits || !wavefront_tiles_enabled() || !wavefront_tiles_supported(map->desc.rows, map->desc.cols)) {
		set_error("the tile form of the wavefront is switched off (PP_WF_TILES=0) or does not support this map size");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipStream_t s = map->ctx->stream;
	std::vector<int32_t> cells;
	goal_cells(map, n_goals, goal_xy_host, cells);
	const int nSlots = n_goals < 16 ? n_goals : 16;
	const int64_t wsb = wavefront_workspace_bytes(map->desc.rows, map->desc.cols);
	DevBuf ws, dc, derr, dtiles, dstats;
	PP_HIP_TRY(ws.alloc((size_t)wsb * nSlots));
	PP_HIP_TRY(dc.alloc((size_t)n_goals * 4));
	PP_HIP_TRY(derr.alloc(8));
	PP_HIP_TRY(dtiles.alloc(64 + (size_t)n_goals * 4));
	PP_HIP_TRY(dstats.alloc(128));
	PP_HIP_TRY(hipMemsetAsync(derr.p, 0, 8, s));
	PP_HIP_TRY(hipMemsetAsync(dtiles.p, 0, 64, s));
	PP_HIP_TRY(hipMemsetAsync(dstats.p, 0, 128, s));
	PP_HIP_TRY(hipMemcpyAsync(dc.p, cells.data(), (size_t)n_goals * 4, hipMemcpyHostToDevice, s));
	WavefrontPublish pub;
	pub.tilesCtl = dtiles.as<int>();
	pub.tilesFallback = dtiles.as<int32_t>() + 16;
	pub.tilesStats = dstats.as<unsigned long long>();
	pub.occBits = map->occBits;
	PP_HIP_TRY(hipEventRecord(map->ctx->ev0, s));
	PP_HIP_TRY(launch_wavefront(s, map->view(), n_goals, dc.as<int32_t>(), cost_dev, ws.p, wsb, nSlots, derr.as<int32_t>(), nullptr, false, nullptr, false, nullptr, nullptr, nullptr,
		nullptr, pub));
	PP_HIP_TRY(hipEventRecord(map->ctx->ev1, s));
	int32_t err = 0;
	PP_HIP_TRY(hipMemcpyAsync(&err, derr.p, 4, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipMemcpyAsync(stats_host, dstats.p, 128, hipMemcpyDeviceToHost, s));
	if (handed_over_host) // (the list outlives the launches: only the count is set back)
		PP_HIP_TRY(hipMemcpyAsync(handed_over_host, dtiles.as<int32_t>() + 16, (size_t)n_goals * 4, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipStreamSynchronize(s));
	stats_host[7] = (uint64_t)((map->desc.rows + 63) / 64) * (uint64_t)((map->desc.cols + 63) / 64);
	if (ms_out)
		PP_HIP_TRY(hipEventElapsedTime(ms_out, map->ctx->ev0, map->ctx->ev1));
	if (err) {
		set_error("obstacle-heuristic open list exceeded its workspace");
		return PP_ERR_CAPACITY;
	}
	return PP_OK;
}

int pp_obstacle_heuristic_profile(pp_map* map, int32_t n_goals, const double* goal_xy_host, uint64_t* counters_host)
{
	if (check_map(map, false) || !map->occ8 || n_goals <= 0 || !goal_xy_host || !counters_host) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipStream_t s = map->ctx->stream;
	std::vector<int32_t> cells;
	goal_cells(map, n_goals, goal_xy_host, cells);
	const int resident = wavefront_resident_blocks();
	int nSlots = n_goals < resident ? n_goals : resident;
	const int64_t wsb = wavefront_workspace_bytes(map->desc.rows, map->desc.cols);
	DevBuf ws, dc, derr, dcost, dprof;
	PP_HIP_TRY(ws.alloc((size_t)wsb * nSlots));
	PP_HIP_TRY(dc.alloc((size_t)n_goals * 4));
	PP_HIP_TRY(derr.alloc(8));
	PP_HIP_TRY(dcost.alloc((size_t)n_goals * map->cells() * 4));
	PP_HIP_TRY(dprof.alloc((size_t)n_goals * 20 * 8));
	PP_HIP_TRY(hipMemsetAsync(derr.p, 0, 8, s));
	PP_HIP_TRY(hipMemcpyAsync(dc.p, cells.data(), (size_t)n_goals * 4, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(launch_wavefront(s, map->view(), n_goals, dc.as<int32_t>(), dcost.as<float>(), ws.p, wsb, nSlots, derr.as<int32_t>(), dprof.as<unsigned long long>()));
	PP_HIP_TRY(hipMemcpyAsync(counters_host, dprof.p, (size_t)n_goals * 20 * 8, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipStreamSynchronize(s));
	return PP_OK;
}

int pp_obstacle_heuristic(pp_map* map, int32_t n_goals, const double* goal_xy_host, float* cost_host)
{
	if (check_map(map, false))
		return PP_ERR_INVALID;
	if (n_goals < 0 || (n_goals > 0 && (!goal_xy_host || !cost_host))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	if (n_goals == 0)
		return PP_OK;
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	DevBuf dcost;
	const size_t bytes = (size_t)n_goals * map->cells() * sizeof(float);
	PP_HIP_TRY(dcost.alloc(bytes));
	if (int rc = pp_obstacle_heuristic_dev(map, n_goals, goal_xy_host, dcost.as<float>()))
		return rc;
	PP_HIP_TRY(hipMemcpy(cost_host, dcost.p, bytes, hipMemcpyDeviceToHost));
	return PP_OK;
}

// ------------------------------------------------------------------ knn ----
int pp_knn_dev(pp_ctx* ctx, int64_t n_points, const double* points_dev, int64_t n_queries, const double* queries_dev, int32_t k, int32_t* idx_dev,
	double* d2_dev)
{
	if (!ctx || n_points < 0 || n_queries < 0 || k < 1 || k > 16 || (n_queries > 0 && (!queries_dev || !idx_dev || !d2_dev)) || (n_points > 0 && !points_dev)) {
		set_error("invalid arguments (k must be 1..16)");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(launch_knn(ctx->stream, n_points, points_dev, n_queries, queries_dev, k, idx_dev, d2_dev));
	return PP_OK;
}

int pp_knn(pp_ctx* ctx, int64_t n_points, const double* points_host, int64_t n_queries, const double* queries_host, int32_t k, int32_t* idx_host,
	double* d2_host)
{
	if (!ctx || n_points < 0 || n_queries < 0 || k < 1 || k > 16 || (n_queries > 0 && (!queries_host || !idx_host || !d2_host)) || (n_points > 0 && !points_host)) {
		set_error("invalid arguments (k must be 1..16)");
		return PP_ERR_INVALID;
	}
	if (n_queries == 0)
		return PP_OK;
	PP_HIP_TRY(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	DevBuf dp, dq, di, dd;
	PP_HIP_TRY(dp.alloc((size_t)n_points * 16));
	PP_HIP_TRY(dq.alloc((size_t)n_queries * 16));
	PP_HIP_TRY(di.alloc((size_t)n_queries * k * 4));
	PP_HIP_TRY(dd.alloc((size_t)n_queries * k * 8));
	if (n_points)
		PP_HIP_TRY(hipMemcpyAsync(dp.p, points_host, (size_t)n_points * 16, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(hipMemcpyAsync(dq.p, queries_host, (size_t)n_queries * 16, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(launch_knn(s, n_points, dp.as<double>(), n_queries, dq.as<double>(), k, di.as<int32_t>(), dd.as<double>()));
	PP_HIP_TRY(hipMemcpyAsync(idx_host, di.p, (size_t)n_queries * k * 4, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipMemcpyAsync(d2_host, dd.p, (size_t)n_queries * k * 8, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipStreamSynchronize(s));
	return PP_OK;
}

} // extern "C"
