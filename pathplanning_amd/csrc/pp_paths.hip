// Path value types of the plugin surface on the device (SURVEY 8a row a7 and the IsPathValid overloads of row a2):
//   paths/path_reeds_shepp.{h,cpp}   PathReedsShepp: constructor (m_final = Interpolate(1.0)), Interpolate, Truncate
//                                    (with the reference's wrong-slot reset, Appendix A Q11), GetDirection,
//                                    GetCuspPointRatios; PathConnectionReedsShepp::Connect
//   paths/path_se2.cpp               PathSE2 (linear in position and heading)
//   state_validator_occupancy_map.cpp:28-71   IsPathValid over either
// One thread per path; the arithmetic is pp_rs_device.hpp's, the same functions the search kernels inline.
#include "pp_internal.hpp"
#include "pp_rs_device.hpp"

using namespace ppd;

namespace {

constexpr int kBlock = 256;
inline int grid_for(int64_t n, int block)
{
	int64_t g = (n + block - 1) / block;
	return (int)(g < 1 ? 1 : (g > 65535 * 4 ? 65535 * 4 : g));
}

static_assert(sizeof(pp_rs_path) == 128, "pp_rs_path is a 128-byte record");

__device__ __forceinline__ rs::Path load_path(const pp_rs_path& r)
{
	rs::Path p;
	p.init = { r.start[0], r.start[1], r.start[2] };
	p.rmin = r.min_turning_radius;
	p.length = r.length;
	p.seg.length = 0.0;
	p.seg.n = 0;
#pragma unroll
	for (int i = 0; i < rs::kNumMotion; i++) {
		p.seg.len[i] = r.motion_length[i];
		p.seg.steer[i] = r.steer[i];
		p.seg.dir[i] = r.direction[i];
	}
	return p;
}

__device__ __forceinline__ void store_motions(pp_rs_path& r, const rs::Path& p)
{
#pragma unroll
	for (int i = 0; i < rs::kNumMotion; i++) {
		r.motion_length[i] = p.seg.len[i];
		r.steer[i] = p.seg.steer[i];
		r.direction[i] = p.seg.dir[i];
	}
}

__global__ void __launch_bounds__(kBlock) k_rs_connect(int64_t n, const double* __restrict__ from, const double* __restrict__ to, double rmin, float rev, float fwd, float sw,
	pp_rs_path* __restrict__ out)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		// Pose2d arguments of the reference arrive through its wrapping constructor
		const Pose a = { from[3 * i], from[3 * i + 1], wrap_theta(from[3 * i + 2]) };
		const Pose b = { to[3 * i], to[3 * i + 1], wrap_theta(to[3 * i + 2]) };
		double t, u, v, segLength;
		float cost;
		const int word = rs::optimal_word(a, b, rmin, rev, fwd, sw, t, u, v, cost, segLength);
		rs::Path p;
		p.init = a;
		p.rmin = rmin;
		if (word >= 0) {
			rs::word_segment(word, t, u, v, p.seg);
		} else { // PathSegment(): every slot invalid, length 0 (reeds_shepp.cpp:430-437)
			p.seg.n = 0;
			p.seg.length = 0.0;
			for (int k = 0; k < rs::kNumMotion; k++) {
				p.seg.len[k] = rs::inf();
				p.seg.steer[k] = (int8_t)rs::kLeft;
				p.seg.dir[k] = (int8_t)rs::kNoMotion;
			}
		}
		p.length = p.seg.length * rmin; // PathSegment::GetLength
		const Pose fin = p.interpolate(1.0);
		pp_rs_path r;
		r.start[0] = a.x, r.start[1] = a.y, r.start[2] = a.t;
		r.final_pose[0] = fin.x, r.final_pose[1] = fin.y, r.final_pose[2] = fin.t;
		store_motions(r, p);
		for (int k = 0; k < 6; k++)
			r.reserved[k] = 0;
		r.min_turning_radius = rmin;
		r.length = p.length;
		r.cost = cost;
		r.word = word;
		out[i] = r;
	}
}

/// op 0: Interpolate + GetDirection at ratio[i]; op 1: Truncate(ratio[i]) in place; op 2: GetCuspPointRatios
__global__ void __launch_bounds__(kBlock) k_rs_path_ops(int op, int64_t n, pp_rs_path* __restrict__ paths, const double* __restrict__ ratio, double* __restrict__ pose,
	int32_t* __restrict__ direction, int q11, double* __restrict__ cuspRatios, int32_t* __restrict__ cuspCount)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		rs::Path p = load_path(paths[i]);
		if (op == 0) {
			if (pose) {
				const Pose s = p.interpolate(ratio[i]);
				pose[3 * i] = s.x, pose[3 * i + 1] = s.y, pose[3 * i + 2] = s.t;
			}
			if (direction)
				direction[i] = p.direction(ratio[i]);
		} else if (op == 1) {
			const Pose fin = p.truncate(ratio[i], q11 != 0);
			pp_rs_path r = paths[i];
			store_motions(r, p);
			r.final_pose[0] = fin.x, r.final_pose[1] = fin.y, r.final_pose[2] = fin.t;
			r.length = p.length;
			paths[i] = r;
		} else {
			double c[4] = { 0.0, 0.0, 0.0, 0.0 };
			const int cnt = p.cusps(c);
			for (int k = 0; k < 4; k++)
				cuspRatios[4 * i + k] = k < cnt ? c[k] : 0.0;
			cuspCount[i] = cnt;
		}
	}
}

__global__ void __launch_bounds__(kBlock) k_check_rs_paths(MapView m, int64_t n, const pp_rs_path* __restrict__ paths, uint8_t* __restrict__ valid, float* __restrict__ last)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const rs::Path p = load_path(paths[i]);
		float l = -1.0f;
		int checks = 0;
		const bool ok = is_path_valid(m, p, p.init, l, checks);
		valid[i] = ok ? 1 : 0;
		if (last)
			last[i] = l;
	}
}

/// PathSE2, paths/path_se2.cpp:5-22
struct Se2Line {
	Pose init, fin;
	double length;
	__device__ __forceinline__ Pose interpolate(double ratio) const
	{
		Pose s;
		s.x = (1 - ratio) * init.x + ratio * fin.x;
		s.y = (1 - ratio) * init.y + ratio * fin.y;
		s.t = (1 - ratio) * init.t + ratio * fin.t; // assigned to the member: not wrapped
		return s;
	}
};

__global__ void __launch_bounds__(kBlock) k_check_se2_paths(MapView m, int64_t n, const double* __restrict__ from, const double* __restrict__ to, uint8_t* __restrict__ valid,
	float* __restrict__ last)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		Se2Line p;
		p.init = { from[3 * i], from[3 * i + 1], wrap_theta(from[3 * i + 2]) };
		p.fin = { to[3 * i], to[3 * i + 1], wrap_theta(to[3 * i + 2]) };
		const double dx = p.fin.x - p.init.x, dy = p.fin.y - p.init.y;
		p.length = sqrt(dx * dx + dy * dy); // (to.position - from.position).norm()
		float l = -1.0f;
		int checks = 0;
		const bool ok = is_path_valid(m, p, p.init, l, checks);
		valid[i] = ok ? 1 : 0;
		if (last)
			last[i] = l;
	}
}

struct Scratch { // device buffers of the host-pointer entry points
	void* p = nullptr;
	~Scratch()
	{
		if (p)
			(void)hipFree(p);
	}
	hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
	template <typename T>
	T* as() { return (T*)p; }
};

using pph::set_error;

int need_ctx(pp_ctx* ctx, int64_t n, bool argsOk)
{
	if (!ctx || n < 0 || !argsOk) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	return PP_OK;
}
int need_map(pp_map* map, int64_t n, bool argsOk)
{
	if (!map || n < 0 || !argsOk) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	if (!map->dist || !map->validBits) {
		set_error("distance grid not uploaded (pp_map_upload_dist2 / pp_map_upload_distance)");
		return PP_ERR_INVALID;
	}
	return PP_OK;
}

} // namespace

extern "C" {

int pp_rs_connect(pp_ctx* ctx, int64_t n, const double* from_host, const double* to_host, double min_turning_radius, float reverse_cost, float forward_cost, float switch_cost,
	pp_rs_path* paths_host)
{
	if (int rc = need_ctx(ctx, n, n == 0 || (from_host && to_host && paths_host)))
		return rc;
	if (n == 0)
		return PP_OK;
	PP_HIP_TRY(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	Scratch df, dt, dp;
	PP_HIP_TRY(df.alloc((size_t)n * 24));
	PP_HIP_TRY(dt.alloc((size_t)n * 24));
	PP_HIP_TRY(dp.alloc((size_t)n * sizeof(pp_rs_path)));
	PP_HIP_TRY(hipMemcpyAsync(df.p, from_host, (size_t)n * 24, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(hipMemcpyAsync(dt.p, to_host, (size_t)n * 24, hipMemcpyHostToDevice, s));
	hipLaunchKernelGGL(k_rs_connect, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, df.as<double>(), dt.as<double>(), min_turning_radius, reverse_cost, forward_cost, switch_cost,
		dp.as<pp_rs_path>());
	PP_HIP_TRY(hipGetLastError());
	PP_HIP_TRY(hipMemcpyAsync(paths_host, dp.p, (size_t)n * sizeof(pp_rs_path), hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipStreamSynchronize(s));
	return PP_OK;
}

static int rs_path_op(pp_ctx* ctx, int op, int64_t n, pp_rs_path* paths_host, bool writeBack, const double* ratio_host, double* pose_host, int32_t* direction_host, int32_t q11,
	double* cusp_host, int32_t* count_host)
{
	PP_HIP_TRY(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	Scratch dp, dr, dpose, ddir, dcr, dcc;
	PP_HIP_TRY(dp.alloc((size_t)n * sizeof(pp_rs_path)));
	PP_HIP_TRY(hipMemcpyAsync(dp.p, paths_host, (size_t)n * sizeof(pp_rs_path), hipMemcpyHostToDevice, s));
	if (ratio_host) {
		PP_HIP_TRY(dr.alloc((size_t)n * 8));
		PP_HIP_TRY(hipMemcpyAsync(dr.p, ratio_host, (size_t)n * 8, hipMemcpyHostToDevice, s));
	}
	if (pose_host)
		PP_HIP_TRY(dpose.alloc((size_t)n * 24));
	if (direction_host)
		PP_HIP_TRY(ddir.alloc((size_t)n * 4));
	if (cusp_host) {
		PP_HIP_TRY(dcr.alloc((size_t)n * 32));
		PP_HIP_TRY(dcc.alloc((size_t)n * 4));
	}
	hipLaunchKernelGGL(k_rs_path_ops, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, op, n, dp.as<pp_rs_path>(), dr.as<double>(), dpose.as<double>(), ddir.as<int32_t>(), (int)q11,
		dcr.as<double>(), dcc.as<int32_t>());
	PP_HIP_TRY(hipGetLastError());
	if (writeBack)
		PP_HIP_TRY(hipMemcpyAsync(paths_host, dp.p, (size_t)n * sizeof(pp_rs_path), hipMemcpyDeviceToHost, s));
	if (pose_host)
		PP_HIP_TRY(hipMemcpyAsync(pose_host, dpose.p, (size_t)n * 24, hipMemcpyDeviceToHost, s));
	if (direction_host)
		PP_HIP_TRY(hipMemcpyAsync(direction_host, ddir.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
	if (cusp_host) {
		PP_HIP_TRY(hipMemcpyAsync(cusp_host, dcr.p, (size_t)n * 32, hipMemcpyDeviceToHost, s));
		PP_HIP_TRY(hipMemcpyAsync(count_host, dcc.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
	}
	PP_HIP_TRY(hipStreamSynchronize(s));
	return PP_OK;
}

int pp_rs_path_interpolate(pp_ctx* ctx, int64_t n, const pp_rs_path* paths_host, const double* ratio_host, double* pose_host, int32_t* direction_host)
{
	if (int rc = need_ctx(ctx, n, n == 0 || (paths_host && ratio_host)))
		return rc;
	if (n == 0 || (!pose_host && !direction_host))
		return PP_OK;
	return rs_path_op(ctx, 0, n, const_cast<pp_rs_path*>(paths_host), false, ratio_host, pose_host, direction_host, 1, nullptr, nullptr);
}

int pp_rs_path_truncate(pp_ctx* ctx, int64_t n, pp_rs_path* paths_host, const double* ratio_host, int32_t q11)
{
	if (int rc = need_ctx(ctx, n, n == 0 || (paths_host && ratio_host)))
		return rc;
	if (n == 0)
		return PP_OK;
	return rs_path_op(ctx, 1, n, paths_host, true, ratio_host, nullptr, nullptr, q11, nullptr, nullptr);
}

int pp_rs_path_cusps(pp_ctx* ctx, int64_t n, const pp_rs_path* paths_host, double* ratios_host, int32_t* count_host)
{
	if (int rc = need_ctx(ctx, n, n == 0 || (paths_host && ratios_host && count_host)))
		return rc;
	if (n == 0)
		return PP_OK;
	return rs_path_op(ctx, 2, n, const_cast<pp_rs_path*>(paths_host), false, nullptr, nullptr, nullptr, 1, ratios_host, count_host);
}

int pp_check_rs_paths_dev(pp_map* map, int64_t n, const pp_rs_path* paths_dev, uint8_t* valid_dev, float* last_ratio_dev)
{
	if (int rc = need_map(map, n, n == 0 || (paths_dev && valid_dev)))
		return rc;
	if (n == 0)
		return PP_OK;
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipLaunchKernelGGL(k_check_rs_paths, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, map->ctx->stream, map->view(), n, paths_dev, valid_dev, last_ratio_dev);
	PP_HIP_TRY(hipGetLastError());
	return PP_OK;
}

int pp_check_rs_paths(pp_map* map, int64_t n, const pp_rs_path* paths_host, uint8_t* valid_host, float* last_ratio_host)
{
	if (int rc = need_map(map, n, n == 0 || (paths_host && valid_host)))
		return rc;
	if (n == 0)
		return PP_OK;
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipStream_t s = map->ctx->stream;
	Scratch dp, dv, dl;
	PP_HIP_TRY(dp.alloc((size_t)n * sizeof(pp_rs_path)));
	PP_HIP_TRY(dv.alloc((size_t)n));
	PP_HIP_TRY(dl.alloc((size_t)n * 4));
	PP_HIP_TRY(hipMemcpyAsync(dp.p, paths_host, (size_t)n * sizeof(pp_rs_path), hipMemcpyHostToDevice, s));
	hipLaunchKernelGGL(k_check_rs_paths, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, map->view(), n, dp.as<pp_rs_path>(), dv.as<uint8_t>(), dl.as<float>());
	PP_HIP_TRY(hipGetLastError());
	PP_HIP_TRY(hipMemcpyAsync(valid_host, dv.p, (size_t)n, hipMemcpyDeviceToHost, s));
	if (last_ratio_host)
		PP_HIP_TRY(hipMemcpyAsync(last_ratio_host, dl.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipStreamSynchronize(s));
	return PP_OK;
}

int pp_check_se2_paths(pp_map* map, int64_t n, const double* from_host, const double* to_host, uint8_t* valid_host, float* last_ratio_host)
{
	if (int rc = need_map(map, n, n == 0 || (from_host && to_host && valid_host)))
		return rc;
	if (n == 0)
		return PP_OK;
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipStream_t s = map->ctx->stream;
	Scratch df, dt, dv, dl;
	PP_HIP_TRY(df.alloc((size_t)n * 24));
	PP_HIP_TRY(dt.alloc((size_t)n * 24));
	PP_HIP_TRY(dv.alloc((size_t)n));
	PP_HIP_TRY(dl.alloc((size_t)n * 4));
	PP_HIP_TRY(hipMemcpyAsync(df.p, from_host, (size_t)n * 24, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(hipMemcpyAsync(dt.p, to_host, (size_t)n * 24, hipMemcpyHostToDevice, s));
	hipLaunchKernelGGL(k_check_se2_paths, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, map->view(), n, df.as<double>(), dt.as<double>(), dv.as<uint8_t>(), dl.as<float>());
	PP_HIP_TRY(hipGetLastError());
	PP_HIP_TRY(hipMemcpyAsync(valid_host, dv.p, (size_t)n, hipMemcpyDeviceToHost, s));
	if (last_ratio_host)
		PP_HIP_TRY(hipMemcpyAsync(last_ratio_host, dl.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipStreamSynchronize(s));
	return PP_OK;
}

int pp_map_upload_distance(pp_map* map, const float* distance_host)
{
	if (!map || !distance_host) {
		set_error("null argument");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	const size_t n = map->cells();
	hipStream_t s = map->ctx->stream;
	if (!map->dist)
		PP_HIP_TRY(hipMalloc((void**)&map->dist, n * sizeof(float)));
	PP_HIP_TRY(hipMemcpyAsync(map->dist, distance_host, n * sizeof(float), hipMemcpyHostToDevice, s));
	if (!map->validBits)
		PP_HIP_TRY(hipMalloc((void**)&map->validBits, ((n + 63) / 64) * 8));
	PP_HIP_TRY(pph::launch_valid_bits(s, map->dist, (int64_t)n, map->minSafeRadius, map->validBits));
	PP_HIP_TRY(hipStreamSynchronize(s));
	return PP_OK;
}

} // extern "C"
