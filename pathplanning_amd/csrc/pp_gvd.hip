// Map authoring and field construction on the device: everything the hot path READS is built here
// from obstacle outlines, so no grid has to come from the host.
//   state_validator/obstacle.cpp:7-61              Shape::RasterizeLine (Bresenham over cells)          -> k_rasterize
//   obstacle_list_occupancy_map.cpp:29-61          AddObstacle / RemoveObstacle (id or -1 per boundary cell)
//   state_validator/gvd.cpp:30-72                  ObstacleDistanceMap::Update                          -> k_propagate
//   state_validator/gvd.cpp:100-131                CheckVoro / CheckVoroConditions                      -> k_voronoi_edges
//   state_validator/gvd.cpp:200-237                VoronoiDistanceMap::Update                           -> k_propagate
//   state_validator/gvd.cpp:266-283                PathCostMap::Update                                  -> k_path_cost
//
// What is exact and what is not.  Rasterisation is integer arithmetic on cells: exact.  PathCostMap::Update is elementwise: the
// same bits as the reference given the same two distance grids.  The two distance maps are NOT the reference's algorithm: the
// reference runs Lau's dynamic brushfire, a sequential std::priority_queue sweep in which the order of equal keys (the heap's
// internal order) decides which obstacle cell a tie keeps and therefore what its neighbours inherit.  That order cannot be
// reproduced without replaying the heap.  Here every cell repeatedly takes the best label among its own and its 8 neighbours'
// (the same 8-neighbour vector propagation, run to its fixed point, ties to the label already held); on the test maps the
// fixed point is the exact Euclidean distance transform, and the brushfire differs from it in 0.02-0.04 % of the cells by at
// most 2 (squared cells) -- tests/test_gpu_gvd.py measures and bounds this.  Voronoi edges follow gvd.cpp's CheckVoro rule
// evaluated on the FINAL labels of every neighbouring pair; the reference evaluates a pair when the later of the two cells is
// popped and can keep marks made with labels that changed afterwards.
#include "pp_internal.hpp"

#include <climits>
#include <vector>

using namespace ppd;
using pph::set_error;

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kNone = 0xFFFFFFFFu; // GridCellPosition(-1, -1)
inline int grid_for(int64_t n, int block)
{
	int64_t g = (n + block - 1) / block;
	return (int)(g < 1 ? 1 : (g > 65535 * 8 ? 65535 * 8 : g));
}

// ---------------------------------------------------------------------------------------------- rasterisation --
/// One thread per segment: Shape::RasterizeLine, obstacle.cpp:7-61 (endpoints through WorldPositionToGridCell(p, false))
__global__ void __launch_bounds__(64) k_rasterize(MapView m, int nSegments, const double* __restrict__ p0, const double* __restrict__ p1, int32_t value, int32_t* __restrict__ occ,
	int32_t* __restrict__ cellCount)
{
	const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (i >= nSegments)
		return;
	int x0, y0, x1, y1;
	world_to_cell(m, p0[2 * i], p0[2 * i + 1], x0, y0);
	world_to_cell(m, p1[2 * i], p1[2 * i + 1], x1, y1);
	const bool steep = abs(y1 - y0) > abs(x1 - x0);
	if (steep) {
		int t = x0;
		x0 = y0;
		y0 = t;
		t = x1;
		x1 = y1;
		y1 = t;
	}
	if (x0 > x1) {
		int t = x0;
		x0 = x1;
		x1 = t;
		t = y0;
		y0 = y1;
		y1 = t;
	}
	const int dx = x1 - x0, dy = abs(y1 - y0);
	int err = dx / 2;
	const int ystep = y0 < y1 ? 1 : -1;
	int y = y0, n = 0;
	for (int x = x0; x <= x1; x++) {
		const int row = steep ? y : x, col = steep ? x : y;
		if (row >= 0 && row < m.rows && col >= 0 && col < m.cols) {
			occ[(int64_t)row * m.cols + col] = value;
			n++;
		}
		err -= dy;
		if (err < 0) {
			y += ystep;
			err += dx;
		}
	}
	if (cellCount)
		atomicAdd(cellCount, n);
}

/// The same walk, cells listed instead of written: segment i fills cells[i * cap ...] (row, col pairs) in Bresenham order and
/// reports how many (in-map cells only, as RasterizeLine appends them)
__global__ void __launch_bounds__(64) k_rasterize_list(MapView m, int nSegments, const double* __restrict__ p0, const double* __restrict__ p1, int cap, int32_t* __restrict__ cells,
	int32_t* __restrict__ count)
{
	const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (i >= nSegments)
		return;
	int x0, y0, x1, y1;
	world_to_cell(m, p0[2 * i], p0[2 * i + 1], x0, y0);
	world_to_cell(m, p1[2 * i], p1[2 * i + 1], x1, y1);
	const bool steep = abs(y1 - y0) > abs(x1 - x0);
	if (steep) {
		int t = x0;
		x0 = y0;
		y0 = t;
		t = x1;
		x1 = y1;
		y1 = t;
	}
	if (x0 > x1) {
		int t = x0;
		x0 = x1;
		x1 = t;
		t = y0;
		y0 = y1;
		y1 = t;
	}
	const int dx = x1 - x0, dy = abs(y1 - y0);
	int err = dx / 2;
	const int ystep = y0 < y1 ? 1 : -1;
	int y = y0, n = 0;
	int32_t* out = cells + (int64_t)i * cap * 2;
	for (int x = x0; x <= x1; x++) {
		const int row = steep ? y : x, col = steep ? x : y;
		if (row >= 0 && row < m.rows && col >= 0 && col < m.cols && n < cap) {
			out[2 * n] = row;
			out[2 * n + 1] = col;
			n++;
		}
		err -= dy;
		if (err < 0) {
			y += ystep;
			err += dx;
		}
	}
	count[i] = n;
}

__global__ void __launch_bounds__(kBlock) k_set_cells(int rows, int cols, int64_t n, const int32_t* __restrict__ cells, int32_t value, int32_t* __restrict__ occ)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int r = cells[2 * i], c = cells[2 * i + 1];
		if (r >= 0 && r < rows && c >= 0 && c < cols)
			occ[(int64_t)r * cols + c] = value;
	}
}

// ------------------------------------------------------------------------------------------------ propagation --
__device__ __forceinline__ int sq_dist(uint32_t label, int r, int c)
{
	const int dr = (int)(label >> 16) - r, dc = (int)(label & 0xFFFFu) - c;
	return dr * dr + dc * dc; // SquaredDistance, gvd.cpp:12-17
}

/// seeds: a cell that is a source labels itself
__global__ void __launch_bounds__(kBlock) k_seed(int rows, int cols, const int32_t* __restrict__ occ, const uint8_t* __restrict__ edge, uint32_t* __restrict__ label)
{
	const int64_t n = (int64_t)rows * cols;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
		const bool src = occ ? occ[i] >= 0 : edge[i] != 0;
		label[i] = src ? ((uint32_t)r << 16) | (uint32_t)c : kNone;
	}
}

/// One Jacobi step of the 8-neighbour vector propagation: label <- the strictly nearest among the neighbours' labels, else
/// its own.  The neighbour order is GetNeighbors' (utils/grid.cpp:29-47); `changed` is raised when any label moved.
__global__ void __launch_bounds__(kBlock) k_propagate(int rows, int cols, const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int32_t* __restrict__ changed)
{
	const int64_t n = (int64_t)rows * cols;
	bool any = false;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
		uint32_t best = in[i];
		int bestD = best == kNone ? INT_MAX : sq_dist(best, r, c);
		const int dR[8] = { 0, -1, 1, 0, -1, 1, -1, 1 }, dC[8] = { -1, -1, -1, 1, 1, 1, 0, 0 };
#pragma unroll
		for (int j = 0; j < 8; j++) {
			const int rr = r + dR[j], cc = c + dC[j];
			if (rr < 0 || rr >= rows || cc < 0 || cc >= cols)
				continue;
			const uint32_t l = in[(int64_t)rr * cols + cc];
			if (l == kNone)
				continue;
			const int d = sq_dist(l, r, c);
			if (d < bestD) {
				bestD = d;
				best = l;
			}
		}
		out[i] = best;
		any = any || best != in[i];
	}
	if (__ballot(any) && (threadIdx.x & 63) == 0)
		atomicOr(changed, 1);
}

__global__ void __launch_bounds__(kBlock) k_labels_to_d2(int rows, int cols, const uint32_t* __restrict__ label, int32_t* __restrict__ d2)
{
	const int64_t n = (int64_t)rows * cols;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
		d2[i] = label[i] == kNone ? INT_MAX : sq_dist(label[i], r, c);
	}
}

// --------------------------------------------------------------------------------------------- Voronoi edges --
/// CheckVoro (gvd.cpp:105-131) for the pair (s, n) on final labels: is s to be marked?
__device__ __forceinline__ bool voro_marks_s(int rows, int cols, const uint32_t* __restrict__ label, const int32_t* __restrict__ occ, int sr, int sc, int nr, int nc)
{
	const uint32_t oS = label[(int64_t)sr * cols + sc], oN = label[(int64_t)nr * cols + nc];
	if (oS == kNone || oN == kNone)
		return false;
	const int oSr = (int)(oS >> 16), oSc = (int)(oS & 0xFFFFu), oNr = (int)(oN >> 16), oNc = (int)(oN & 0xFFFFu);
	if (occ[(int64_t)oSr * cols + oSc] == occ[(int64_t)oNr * cols + oNc])
		return false; // same obstacle id
	const int dS = sq_dist(oS, sr, sc), dN = sq_dist(oN, nr, nc);
	if (!(dS > 1 || dN > 1))
		return false;
	if (!(abs(oSr - oNr) > 1 || abs(oSc - oNc) > 1))
		return false;
	const int sStability = sq_dist(oN, sr, sc) - dS, nStability = sq_dist(oS, nr, nc) - dN;
	if (sStability < 0 || nStability < 0)
		return false;
	return sStability <= nStability;
}

__global__ void __launch_bounds__(kBlock) k_voronoi_edges(int rows, int cols, const uint32_t* __restrict__ label, const int32_t* __restrict__ occ, uint8_t* __restrict__ edge)
{
	const int64_t n = (int64_t)rows * cols;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
		bool mark = false;
		const int dR[8] = { 0, -1, 1, 0, -1, 1, -1, 1 }, dC[8] = { -1, -1, -1, 1, 1, 1, 0, 0 };
#pragma unroll
		for (int j = 0; j < 8; j++) {
			const int rr = r + dR[j], cc = c + dC[j];
			if (rr < 0 || rr >= rows || cc < 0 || cc >= cols)
				continue;
			mark = mark || voro_marks_s(rows, cols, label, occ, r, c, rr, cc);
		}
		edge[i] = mark ? 1 : 0;
	}
}

// ------------------------------------------------------------------------------------------------- path cost --
/// PathCostMap::Update, gvd.cpp:266-283, with the accessors of gvd.h:38 and :77: float(sqrt(int) * resolution).
/// Types as in the reference: the two quotients are float, pow(float, int) promotes to double, the product with it is
/// double, the store converts to float.  (x - dMax) is a float, its square is exact in double.
__global__ void __launch_bounds__(kBlock) k_path_cost(int64_t n, const int32_t* __restrict__ obstD2, const int32_t* __restrict__ voroD2, float resolution, float alpha, float dMax,
	float* __restrict__ cost)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const float obstDist = (float)(sqrt((double)obstD2[i]) * (double)resolution);
		const float voroDist = (float)(sqrt((double)voroD2[i]) * (double)resolution);
		float out = 0.0f;
		if (!(obstDist >= dMax || voroDist == __builtin_huge_valf())) {
			const float a = alpha / (alpha + obstDist);
			const float b = voroDist / (obstDist + voroDist);
			const double dm = (double)(obstDist - dMax), dq = (double)dMax;
			out = (float)((double)(a * b) * ((dm * dm) / (dq * dq)));
		}
		cost[i] = out;
	}
}

hipError_t ensure(void** p, size_t bytes)
{
	return *p ? hipSuccess : hipMalloc(p, bytes ? bytes : 1);
}

/// runs k_propagate to its fixed point; the result is in label[result]
int propagate_to_fixpoint(pp_map* map, uint32_t* label[2], int& result, int& iterations)
{
	hipStream_t s = map->ctx->stream;
	const int rows = map->desc.rows, cols = map->desc.cols;
	const int64_t n = (int64_t)rows * cols;
	int cur = 0;
	iterations = 0;
	constexpr int kChunk = 16; // Jacobi steps between two looks at the flag
	const int maxIter = 2 * (rows + cols) + 16;
	for (;;) {
		PP_HIP_TRY(hipMemsetAsync(map->gvdFlag, 0, 4, s));
		for (int k = 0; k < kChunk; k++) {
			hipLaunchKernelGGL(k_propagate, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, rows, cols, label[cur], label[cur ^ 1], map->gvdFlag);
			cur ^= 1;
		}
		PP_HIP_TRY(hipGetLastError());
		int32_t flag = 0;
		PP_HIP_TRY(hipMemcpyAsync(&flag, map->gvdFlag, 4, hipMemcpyDeviceToHost, s));
		PP_HIP_TRY(hipStreamSynchronize(s));
		iterations += kChunk;
		if (!flag || iterations > maxIter)
			break;
	}
	result = cur;
	return PP_OK;
}

} // namespace

extern "C" {

int pp_map_rasterize_segments(pp_map* map, int32_t n_segments, const double* p0_xy_host, const double* p1_xy_host, int32_t value, int32_t* n_cells_out)
{
	if (!map || n_segments < 0 || (n_segments > 0 && (!p0_xy_host || !p1_xy_host))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipStream_t s = map->ctx->stream;
	const size_t n = map->cells();
	if (!map->occ32) {
		PP_HIP_TRY(hipMalloc((void**)&map->occ32, n * 4));
		PP_HIP_TRY(hipMemsetAsync(map->occ32, 0xFF, n * 4, s)); // -1: free (occupancy_map.cpp:12)
	}
	PP_HIP_TRY(ensure((void**)&map->gvdFlag, 64));
	if (n_segments > 0) {
		double *d0 = nullptr, *d1 = nullptr;
		PP_HIP_TRY(hipMalloc((void**)&d0, (size_t)n_segments * 16));
		hipError_t e = hipMalloc((void**)&d1, (size_t)n_segments * 16);
		if (e == hipSuccess)
			e = hipMemcpyAsync(d0, p0_xy_host, (size_t)n_segments * 16, hipMemcpyHostToDevice, s);
		if (e == hipSuccess)
			e = hipMemcpyAsync(d1, p1_xy_host, (size_t)n_segments * 16, hipMemcpyHostToDevice, s);
		if (e == hipSuccess)
			e = hipMemsetAsync(map->gvdFlag + 1, 0, 4, s);
		if (e == hipSuccess) {
			hipLaunchKernelGGL(k_rasterize, dim3((n_segments + 63) / 64), dim3(64), 0, s, map->view(), n_segments, d0, d1, value, map->occ32, map->gvdFlag + 1);
			e = hipGetLastError();
		}
		int32_t cnt = 0;
		if (e == hipSuccess)
			e = hipMemcpyAsync(&cnt, map->gvdFlag + 1, 4, hipMemcpyDeviceToHost, s);
		if (e == hipSuccess)
			e = hipStreamSynchronize(s);
		(void)hipFree(d0);
		(void)hipFree(d1);
		if (e != hipSuccess)
			return pph::hip_fail(e, "pp_map_rasterize_segments");
		if (n_cells_out)
			*n_cells_out = cnt;
	} else if (n_cells_out) {
		*n_cells_out = 0;
	}
	// the wavefront / search kernels read the packed occupancy
	if (!map->occ8)
		PP_HIP_TRY(hipMalloc((void**)&map->occ8, n));
	PP_HIP_TRY(pph::launch_occ_to_u8(s, map->occ32, map->occ8, (int64_t)n));
	PP_HIP_TRY(hipStreamSynchronize(s));
	return PP_OK;
}

int pp_rasterize_cells(pp_map* map, int32_t n_segments, const double* p0_xy_host, const double* p1_xy_host, int32_t cap_per_segment, int32_t* cells_host, int32_t* count_host)
{
	if (!map || n_segments < 0 || cap_per_segment < 1 || (n_segments > 0 && (!p0_xy_host || !p1_xy_host || !cells_host || !count_host))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	if (n_segments == 0)
		return PP_OK;
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipStream_t s = map->ctx->stream;
	double *d0 = nullptr, *d1 = nullptr;
	int32_t *dc = nullptr, *dn = nullptr;
	const size_t cellBytes = (size_t)n_segments * cap_per_segment * 8;
	hipError_t e = hipMalloc((void**)&d0, (size_t)n_segments * 16);
	if (e == hipSuccess)
		e = hipMalloc((void**)&d1, (size_t)n_segments * 16);
	if (e == hipSuccess)
		e = hipMalloc((void**)&dc, cellBytes);
	if (e == hipSuccess)
		e = hipMalloc((void**)&dn, (size_t)n_segments * 4);
	if (e == hipSuccess)
		e = hipMemcpyAsync(d0, p0_xy_host, (size_t)n_segments * 16, hipMemcpyHostToDevice, s);
	if (e == hipSuccess)
		e = hipMemcpyAsync(d1, p1_xy_host, (size_t)n_segments * 16, hipMemcpyHostToDevice, s);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_rasterize_list, dim3((n_segments + 63) / 64), dim3(64), 0, s, map->view(), n_segments, d0, d1, cap_per_segment, dc, dn);
		e = hipGetLastError();
	}
	if (e == hipSuccess)
		e = hipMemcpyAsync(cells_host, dc, cellBytes, hipMemcpyDeviceToHost, s);
	if (e == hipSuccess)
		e = hipMemcpyAsync(count_host, dn, (size_t)n_segments * 4, hipMemcpyDeviceToHost, s);
	if (e == hipSuccess)
		e = hipStreamSynchronize(s);
	for (void* q : { (void*)d0, (void*)d1, (void*)dc, (void*)dn })
		if (q)
			(void)hipFree(q);
	if (e != hipSuccess)
		return pph::hip_fail(e, "pp_rasterize_cells");
	return PP_OK;
}

int pp_map_set_cells(pp_map* map, int64_t n_cells, const int32_t* cells_host, int32_t value)
{
	if (!map || n_cells < 0 || (n_cells > 0 && !cells_host)) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipStream_t s = map->ctx->stream;
	const size_t n = map->cells();
	if (!map->occ32) {
		PP_HIP_TRY(hipMalloc((void**)&map->occ32, n * 4));
		PP_HIP_TRY(hipMemsetAsync(map->occ32, 0xFF, n * 4, s));
	}
	if (n_cells > 0) {
		int32_t* dc = nullptr;
		PP_HIP_TRY(hipMalloc((void**)&dc, (size_t)n_cells * 8));
		hipError_t e = hipMemcpyAsync(dc, cells_host, (size_t)n_cells * 8, hipMemcpyHostToDevice, s);
		if (e == hipSuccess) {
			hipLaunchKernelGGL(k_set_cells, dim3(grid_for(n_cells, kBlock)), dim3(kBlock), 0, s, map->desc.rows, map->desc.cols, n_cells, dc, value, map->occ32);
			e = hipGetLastError();
		}
		if (e == hipSuccess)
			e = hipStreamSynchronize(s);
		(void)hipFree(dc);
		if (e != hipSuccess)
			return pph::hip_fail(e, "pp_map_set_cells");
	}
	if (!map->occ8)
		PP_HIP_TRY(hipMalloc((void**)&map->occ8, n));
	PP_HIP_TRY(pph::launch_occ_to_u8(s, map->occ32, map->occ8, (int64_t)n));
	PP_HIP_TRY(hipStreamSynchronize(s));
	return PP_OK;
}

int pp_map_download_occupancy(pp_map* map, int32_t* occ_host)
{
	if (!map || !occ_host || !map->occ32) {
		set_error("no device occupancy grid (pp_map_rasterize_segments / pp_map_upload_occupancy first)");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	PP_HIP_TRY(hipMemcpy(occ_host, map->occ32, map->cells() * 4, hipMemcpyDeviceToHost));
	return PP_OK;
}

int pp_map_update_gvd(pp_map* map, float alpha, float d_max, int32_t* iterations_out)
{
	if (!map || !map->occ32) {
		set_error("no device occupancy grid (pp_map_rasterize_segments / pp_map_upload_occupancy first)");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipStream_t s = map->ctx->stream;
	const int rows = map->desc.rows, cols = map->desc.cols;
	const size_t n = map->cells();
	if (rows > 65535 || cols > 65535) {
		set_error("grids beyond 65535 cells a side are not supported by the label encoding");
		return PP_ERR_CAPACITY;
	}
	PP_HIP_TRY(ensure((void**)&map->gvdFlag, 64));
	for (int k = 0; k < 2; k++) {
		PP_HIP_TRY(ensure((void**)&map->obstLabel[k], n * 4));
		PP_HIP_TRY(ensure((void**)&map->voroLabel[k], n * 4));
	}
	PP_HIP_TRY(ensure((void**)&map->d2, n * 4));
	PP_HIP_TRY(ensure((void**)&map->voroD2, n * 4));
	PP_HIP_TRY(ensure((void**)&map->voroEdge, n));
	PP_HIP_TRY(ensure((void**)&map->dist, n * 4));
	PP_HIP_TRY(ensure((void**)&map->pathcost, n * 4));
	PP_HIP_TRY(ensure((void**)&map->validBits, ((n + 63) / 64) * 8));
	const int grid = grid_for((int64_t)n, kBlock);
	// ---- ObstacleDistanceMap::Update
	hipLaunchKernelGGL(k_seed, dim3(grid), dim3(kBlock), 0, s, rows, cols, map->occ32, (const uint8_t*)nullptr, map->obstLabel[0]);
	int it1 = 0, it2 = 0;
	if (int rc = propagate_to_fixpoint(map, map->obstLabel, map->obstResult, it1))
		return rc;
	hipLaunchKernelGGL(k_labels_to_d2, dim3(grid), dim3(kBlock), 0, s, rows, cols, map->obstLabel[map->obstResult], map->d2);
	// ---- Voronoi edges + VoronoiDistanceMap::Update
	hipLaunchKernelGGL(k_voronoi_edges, dim3(grid), dim3(kBlock), 0, s, rows, cols, map->obstLabel[map->obstResult], map->occ32, map->voroEdge);
	hipLaunchKernelGGL(k_seed, dim3(grid), dim3(kBlock), 0, s, rows, cols, (const int32_t*)nullptr, map->voroEdge, map->voroLabel[0]);
	PP_HIP_TRY(hipGetLastError());
	if (int rc = propagate_to_fixpoint(map, map->voroLabel, map->voroResult, it2))
		return rc;
	hipLaunchKernelGGL(k_labels_to_d2, dim3(grid), dim3(kBlock), 0, s, rows, cols, map->voroLabel[map->voroResult], map->voroD2);
	// ---- PathCostMap::Update, then what the validator reads: float distances and the validity bitmap
	hipLaunchKernelGGL(k_path_cost, dim3(grid), dim3(kBlock), 0, s, (int64_t)n, map->d2, map->voroD2, map->desc.resolution, alpha, d_max, map->pathcost);
	PP_HIP_TRY(hipGetLastError());
	PP_HIP_TRY(pph::launch_d2_to_distance(s, map->d2, map->dist, (int64_t)n, map->desc.resolution));
	PP_HIP_TRY(pph::launch_valid_bits(s, map->dist, (int64_t)n, map->minSafeRadius, map->validBits));
	PP_HIP_TRY(hipStreamSynchronize(s));
	if (iterations_out)
		*iterations_out = it1 + it2;
	return PP_OK;
}

int pp_map_download_gvd(pp_map* map, int32_t* d2_host, int32_t* nearest_obstacle_host, uint8_t* voronoi_edge_host, int32_t* voronoi_d2_host, int32_t* nearest_edge_host, float* path_cost_host)
{
	if (!map || !map->d2 || !map->voroD2 || !map->pathcost || !map->obstLabel[0]) {
		set_error("fields not built (pp_map_update_gvd first)");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	const size_t n = map->cells();
	if (d2_host)
		PP_HIP_TRY(hipMemcpy(d2_host, map->d2, n * 4, hipMemcpyDeviceToHost));
	if (voronoi_edge_host)
		PP_HIP_TRY(hipMemcpy(voronoi_edge_host, map->voroEdge, n, hipMemcpyDeviceToHost));
	if (voronoi_d2_host)
		PP_HIP_TRY(hipMemcpy(voronoi_d2_host, map->voroD2, n * 4, hipMemcpyDeviceToHost));
	if (path_cost_host)
		PP_HIP_TRY(hipMemcpy(path_cost_host, map->pathcost, n * 4, hipMemcpyDeviceToHost));
	auto labels = [&](const uint32_t* dev, int32_t* out) -> int {
		std::vector<uint32_t> tmp(n);
		PP_HIP_TRY(hipMemcpy(tmp.data(), dev, n * 4, hipMemcpyDeviceToHost));
		for (size_t i = 0; i < n; i++) {
			out[2 * i] = tmp[i] == kNone ? -1 : (int32_t)(tmp[i] >> 16);
			out[2 * i + 1] = tmp[i] == kNone ? -1 : (int32_t)(tmp[i] & 0xFFFFu);
		}
		return PP_OK;
	};
	if (nearest_obstacle_host)
		if (int rc = labels(map->obstLabel[map->obstResult], nearest_obstacle_host))
			return rc;
	if (nearest_edge_host)
		if (int rc = labels(map->voroLabel[map->voroResult], nearest_edge_host))
			return rc;
	return PP_OK;
}

int pp_map_upload_nearest_cells(pp_map* map, const int32_t* nearest_obstacle_host, const int32_t* nearest_edge_host)
{
	if (!map || !nearest_obstacle_host || !nearest_edge_host) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	if (map->desc.rows > 65535 || map->desc.cols > 65535) {
		set_error("grids beyond 65535 cells a side are not supported by the label encoding");
		return PP_ERR_CAPACITY;
	}
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	const size_t n = map->cells();
	PP_HIP_TRY(ensure((void**)&map->obstLabel[0], n * 4));
	PP_HIP_TRY(ensure((void**)&map->voroLabel[0], n * 4));
	std::vector<uint32_t> a(n), b(n);
	for (size_t i = 0; i < n; i++) {
		a[i] = nearest_obstacle_host[2 * i] < 0 ? kNone : ((uint32_t)nearest_obstacle_host[2 * i] << 16) | (uint32_t)nearest_obstacle_host[2 * i + 1];
		b[i] = nearest_edge_host[2 * i] < 0 ? kNone : ((uint32_t)nearest_edge_host[2 * i] << 16) | (uint32_t)nearest_edge_host[2 * i + 1];
	}
	PP_HIP_TRY(hipMemcpy(map->obstLabel[0], a.data(), n * 4, hipMemcpyHostToDevice));
	PP_HIP_TRY(hipMemcpy(map->voroLabel[0], b.data(), n * 4, hipMemcpyHostToDevice));
	map->obstResult = 0;
	map->voroResult = 0;
	return PP_OK;
}

/// PathCostMap::Update alone, over grids supplied by the caller (e.g. the reference's own two brushfire maps): the
/// elementwise step of gvd.cpp:266-283 with the reference's exact type mix.
int pp_path_cost_update(pp_map* map, const int32_t* obstacle_d2_host, const int32_t* voronoi_d2_host, float alpha, float d_max, float* path_cost_host)
{
	if (!map || !obstacle_d2_host || !voronoi_d2_host) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipStream_t s = map->ctx->stream;
	const size_t n = map->cells();
	int32_t *a = nullptr, *b = nullptr;
	PP_HIP_TRY(hipMalloc((void**)&a, n * 4));
	hipError_t e = hipMalloc((void**)&b, n * 4);
	if (e == hipSuccess)
		e = ensure((void**)&map->pathcost, n * 4);
	if (e == hipSuccess)
		e = hipMemcpyAsync(a, obstacle_d2_host, n * 4, hipMemcpyHostToDevice, s);
	if (e == hipSuccess)
		e = hipMemcpyAsync(b, voronoi_d2_host, n * 4, hipMemcpyHostToDevice, s);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_path_cost, dim3(grid_for((int64_t)n, kBlock)), dim3(kBlock), 0, s, (int64_t)n, a, b, map->desc.resolution, alpha, d_max, map->pathcost);
		e = hipGetLastError();
	}
	if (e == hipSuccess && path_cost_host)
		e = hipMemcpyAsync(path_cost_host, map->pathcost, n * 4, hipMemcpyDeviceToHost, s);
	if (e == hipSuccess)
		e = hipStreamSynchronize(s);
	(void)hipFree(a);
	(void)hipFree(b);
	if (e != hipSuccess)
		return pph::hip_fail(e, "pp_path_cost_update");
	return PP_OK;
}

} // extern "C"
