// Map authoring and field construction on the device: everything the hot path READS is built here
// from obstacle outlines, so no grid has to come from the host.
//   state_validator/obstacle.cpp:7-61              Shape::RasterizeLine (Bresenham over cells)          -> k_rasterize_list
//   obstacle_list_occupancy_map.cpp:29-61          AddObstacle / RemoveObstacle (id or -1 per boundary cell)
//   state_validator/gvd.cpp:30-72                  ObstacleDistanceMap::Update                          -> host brushfire | k_edt_rows + k_edt_cols
//   state_validator/gvd.cpp:100-131                CheckVoro / CheckVoroConditions                      -> k_voronoi_edges
//   state_validator/gvd.cpp:200-237                VoronoiDistanceMap::Update                           -> host brushfire | k_edt_rows + k_edt_cols
//   state_validator/gvd.cpp:266-283                PathCostMap::Update                                  -> k_path_cost
//
// Two modes for GVD::Update's two distance maps (pp_map_update_gvd_ex):
//   PP_GVD_REFERENCE_ORDER  the reference's own algorithm -- Lau's dynamic brushfire over a std::priority_queue, where the order among
//                           equal keys decides which source a tie keeps -- replayed on the host over the recorded sequence of cell
//                           edits (pp_brushfire_host.hpp): d2, nearest cells, Voronoi edges, Voronoi d2 are the reference's bit for
//                           bit, and edits after the first build are incremental (its raise / lower waves).
//   PP_GVD_EXACT_EDT        the throughput mode: an exact separable Euclidean transform on the device (k_edt_rows, k_edt_cols; no
//                           host round trip), Voronoi edges from gvd.cpp's CheckVoro rule on the FINAL labels of every neighbouring
//                           pair.  Its distances are the true minima; the brushfire's are not always (it over-estimates a few cells
//                           in ten thousand), so the two modes differ there and in the ties -- tests/test_gpu_gvd.py measures it.
// Rasterisation is integer arithmetic on cells and PathCostMap::Update is elementwise with the reference's type mix: both exact.
#include "pp_internal.hpp"
#include "pp_brushfire_host.hpp"

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
/// One thread per segment: Shape::RasterizeLine, obstacle.cpp:7-61 (endpoints through WorldPositionToGridCell(p, false)).  Segment i
/// fills cells[i * cap ...] (row, col pairs) in Bresenham order and reports how many (in-map cells only, as RasterizeLine appends them)
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

// ------------------------------------------------------------------------ exact Euclidean distance transform --
__device__ __forceinline__ int sq_dist(uint32_t label, int r, int c)
{
	const int dr = (int)(label >> 16) - r, dc = (int)(label & 0xFFFFu) - c;
	return dr * dr + dc * dc; // SquaredDistance, gvd.cpp:12-17
}

/// Pass 1, along a row (the contiguous direction): near[r][c] = column of the nearest source cell IN ROW r, -1 if the row has
/// none; of two equally near ones the left.  One workgroup per row walks it in tiles of kBlock columns, once left to right
/// carrying the last source seen (an inclusive max-scan of "my column if I am a source": DPP-free wave shuffles, the four
/// wave totals through LDS), once right to left carrying the next one (min-scan).  Sources: occ >= 0, or edge != 0.
__global__ void __launch_bounds__(kBlock) k_edt_rows(int rows, int cols, const int32_t* __restrict__ occ, const uint8_t* __restrict__ edge, int32_t* __restrict__ near, int32_t* __restrict__ anySource)
{
	__shared__ int s_wave[kBlock / 64];
	__shared__ int s_carry;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	bool found = false;
	for (int r = blockIdx.x; r < rows; r += gridDim.x) {
		const int64_t base = (int64_t)r * cols;
		// ---- left to right: last source at or before c
		if (tid == 0)
			s_carry = -1;
		__syncthreads();
		for (int c0 = 0; c0 < cols; c0 += kBlock) {
			const int c = c0 + tid;
			const bool src = c < cols && (occ ? occ[base + c] >= 0 : edge[base + c] != 0);
			int v = src ? c : -1;
			for (int d = 1; d < 64; d <<= 1) {
				const int o = __shfl_up(v, d);
				if (lane >= d && o > v)
					v = o;
			}
			if (lane == 63)
				s_wave[wv] = v;
			__syncthreads();
			int before = s_carry;
			for (int k = 0; k < wv; k++)
				before = s_wave[k] > before ? s_wave[k] : before;
			v = before > v ? before : v;
			if (c < cols)
				near[base + c] = v;
			__syncthreads();
			if (tid == kBlock - 1)
				s_carry = v;
			__syncthreads();
		}
		found = found || s_carry >= 0;
		__syncthreads();
		// ---- right to left: first source at or after c, merged with the left one
		if (tid == 0)
			s_carry = INT_MAX;
		__syncthreads();
		const int tiles = (cols + kBlock - 1) / kBlock;
		for (int t = tiles - 1; t >= 0; t--) {
			const int c = t * kBlock + tid;
			const bool src = c < cols && (occ ? occ[base + c] >= 0 : edge[base + c] != 0);
			int v = src ? c : INT_MAX;
			for (int d = 1; d < 64; d <<= 1) {
				const int o = __shfl_down(v, d);
				if (lane + d < 64 && o < v)
					v = o;
			}
			if (lane == 0)
				s_wave[wv] = v;
			__syncthreads();
			int after = s_carry;
			for (int k = wv + 1; k < kBlock / 64; k++)
				after = s_wave[k] < after ? s_wave[k] : after;
			v = after < v ? after : v;
			if (c < cols) {
				const int left = near[base + c];
				int best = left;
				if (v != INT_MAX && (left < 0 || v - c < c - left))
					best = v;
				near[base + c] = best;
			}
			__syncthreads();
			if (tid == 0)
				s_carry = v;
			__syncthreads();
		}
	}
	if (found && tid == 0)
		atomicOr(anySource, 1);
}

/// Pass 2, across rows: the nearest source of (r, c) is (r', near[r'][c]) for the r' that minimises (r - r')^2 + (c - near[r'][c])^2.
/// Rows are visited outwards from r (r, r-1, r+1, r-2, ...) and the walk ends once k^2 alone reaches the best value, so a cell
/// looks at about as many rows as it is cells away from its source; a wave covers 64 consecutive columns, every step one coalesced
/// load per direction.  Ties keep the candidate seen first (smaller |r' - r|, then the lower row).  Exact by construction --
/// every cell's minimum over ALL sources -- which the 8-neighbour propagation this replaces is not in general.
__global__ void __launch_bounds__(kBlock) k_edt_cols(int rows, int cols, const int32_t* __restrict__ near, const int32_t* __restrict__ anySource, uint32_t* __restrict__ label, int32_t* __restrict__ d2)
{
	const int64_t n = (int64_t)rows * cols;
	const bool any = *anySource != 0;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
		int best = INT_MAX, bestR = -1, bestC = -1;
		if (any) {
			const int g0 = near[i];
			if (g0 >= 0) {
				best = (c - g0) * (c - g0);
				bestR = r;
				bestC = g0;
			}
			for (int k = 1; k < rows; k++) {
				const long long kk = (long long)k * k;
				if (kk >= (long long)best || (r - k < 0 && r + k >= rows))
					break;
				if (r - k >= 0) {
					const int g = near[i - (int64_t)k * cols];
					if (g >= 0) {
						const int d = (int)kk + (c - g) * (c - g);
						if (d < best) {
							best = d;
							bestR = r - k;
							bestC = g;
						}
					}
				}
				if (r + k < rows) {
					const int g = near[i + (int64_t)k * cols];
					if (g >= 0) {
						const int d = (int)kk + (c - g) * (c - g);
						if (d < best) {
							best = d;
							bestR = r + k;
							bestC = g;
						}
					}
				}
			}
		}
		label[i] = bestR < 0 ? kNone : ((uint32_t)bestR << 16) | (uint32_t)bestC;
		d2[i] = best;
	}
}

__global__ void __launch_bounds__(kBlock) k_edges_from_labels(int64_t n, int cols, const uint32_t* __restrict__ label, uint8_t* __restrict__ edge)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
		edge[i] = label[i] == (((uint32_t)r << 16) | (uint32_t)c) ? 1 : 0;
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

/// exact transform of one source set: labels into label[0], squared distances into d2; label[1] is the row pass's scratch
hipError_t launch_edt(pp_map* map, const int32_t* occ, const uint8_t* edge, uint32_t* label[2], int32_t* d2)
{
	hipStream_t s = map->ctx->stream;
	const int rows = map->desc.rows, cols = map->desc.cols;
	hipError_t e = hipMemsetAsync(map->gvdFlag + 2, 0, 4, s);
	if (e != hipSuccess)
		return e;
	hipLaunchKernelGGL(k_edt_rows, dim3(rows < 4096 ? rows : 4096), dim3(kBlock), 0, s, rows, cols, occ, edge, (int32_t*)label[1], map->gvdFlag + 2);
	hipLaunchKernelGGL(k_edt_cols, dim3(grid_for((int64_t)rows * cols, kBlock)), dim3(kBlock), 0, s, rows, cols, (const int32_t*)label[1], map->gvdFlag + 2, label[0], d2);
	return hipGetLastError();
}

/// applies the edits recorded since the last reference-order update to the host brushfire (creating it, or rebuilding it
/// from the device grid when the record is not usable), runs GVD::Update's two sweeps there and uploads the four grids
int update_reference_order(pp_map* map)
{
	hipStream_t s = map->ctx->stream;
	const int rows = map->desc.rows, cols = map->desc.cols;
	const size_t n = map->cells();
	if (map->journalReset || map->journalLost || !map->gvdRef) {
		delete map->gvdRef;
		map->gvdRef = new pph::GvdReference(rows, cols);
		if (map->journalLost) { // edits were dropped: the sequence becomes "every occupied cell of the grid, row-major"
			std::vector<int32_t> occ(n);
			PP_HIP_TRY(hipMemcpyAsync(occ.data(), map->occ32, n * 4, hipMemcpyDeviceToHost, s));
			PP_HIP_TRY(hipStreamSynchronize(s));
			map->journal.clear();
			for (size_t i = 0; i < n; i++)
				if (occ[i] >= 0) {
					map->journal.push_back((int32_t)i);
					map->journal.push_back(occ[i]);
				}
		}
		map->journalReset = map->journalLost = false;
	}
	pph::GvdReference& G = *map->gvdRef;
	for (size_t k = 0; k + 1 < map->journal.size(); k += 2)
		G.edit(map->journal[k], map->journal[k + 1]);
	map->journal.clear();
	map->journal.shrink_to_fit();
	G.update();
	std::vector<uint32_t> lab(n);
	auto pack = [&](const std::vector<int32_t>& src) {
		for (size_t i = 0; i < n; i++)
			lab[i] = src[i] < 0 ? kNone : ((uint32_t)(src[i] / cols) << 16) | (uint32_t)(src[i] % cols);
	};
	PP_HIP_TRY(hipMemcpyAsync(map->d2, G.obstacles.dist.data(), n * 4, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(hipMemcpyAsync(map->voroD2, G.edges.dist.data(), n * 4, hipMemcpyHostToDevice, s));
	pack(G.obstacles.source);
	PP_HIP_TRY(hipMemcpyAsync(map->obstLabel[0], lab.data(), n * 4, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(hipStreamSynchronize(s)); // `lab` is reused
	pack(G.edges.source);
	PP_HIP_TRY(hipMemcpyAsync(map->voroLabel[0], lab.data(), n * 4, hipMemcpyHostToDevice, s));
	hipLaunchKernelGGL(k_edges_from_labels, dim3(grid_for((int64_t)n, kBlock)), dim3(kBlock), 0, s, (int64_t)n, cols, map->voroLabel[0], map->voroEdge);
	PP_HIP_TRY(hipGetLastError());
	PP_HIP_TRY(hipStreamSynchronize(s));
	return PP_OK;
}

/// an ordered edit (cell, value) for the reference-order mode; the record is bounded: beyond 8 entries per cell it is dropped
/// and the next reference-order update starts again from the grid itself
void journal_edit(pp_map* map, int32_t cell, int32_t value)
{
	if (map->journalLost)
		return;
	if (map->journal.size() > 16 * map->cells()) {
		map->journal.clear();
		map->journal.shrink_to_fit();
		map->journalLost = true;
		return;
	}
	map->journal.push_back(cell);
	map->journal.push_back(value);
}

} // namespace

void pph::gvd_reference_free(pp_map* map)
{
	delete map->gvdRef;
	map->gvdRef = nullptr;
}

extern "C" {

int pp_map_rasterize_segments(pp_map* map, int32_t n_segments, const double* p0_xy_host, const double* p1_xy_host, int32_t value, int32_t* n_cells_out)
{
	if (!map || n_segments < 0 || (n_segments > 0 && (!p0_xy_host || !p1_xy_host))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	// = the cell lists of the segments (device walk), written in list order: the same cells whichever way they are written, and
	// the ORDER -- segment after segment, Bresenham order inside one -- is recorded for the reference-order field update, as
	// AddObstacle's loop over GetBoundaryGridCellPosition fixes it (obstacle_list_occupancy_map.cpp:36-39)
	const int cap = map->desc.rows + map->desc.cols + 2; // a line holds at most max(rows, columns) in-map cells
	std::vector<int32_t> rc((size_t)n_segments * cap * 2), count((size_t)n_segments), cells;
	if (n_segments > 0)
		if (int e = pp_rasterize_cells(map, n_segments, p0_xy_host, p1_xy_host, cap, rc.data(), count.data()))
			return e;
	for (int k = 0; k < n_segments; k++)
		cells.insert(cells.end(), rc.begin() + (size_t)k * cap * 2, rc.begin() + ((size_t)k * cap + count[k]) * 2);
	if (n_cells_out)
		*n_cells_out = (int32_t)(cells.size() / 2);
	return pp_map_set_cells(map, (int64_t)(cells.size() / 2), cells.data(), value);
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
	for (int64_t i = 0; i < n_cells; i++) { // the order of the edits is part of the reference's result (reference-order mode)
		const int r = cells_host[2 * i], c = cells_host[2 * i + 1];
		if (r >= 0 && r < map->desc.rows && c >= 0 && c < map->desc.cols)
			journal_edit(map, r * map->desc.cols + c, value);
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
	if (int rc = pph::refresh_occupancy_views(map, s))
		return rc;
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
	return pp_map_update_gvd_ex(map, alpha, d_max, PP_GVD_EXACT_EDT, iterations_out);
}

int pp_map_update_gvd_ex(pp_map* map, float alpha, float d_max, int32_t mode, int32_t* iterations_out)
{
	if (!map || !map->occ32) {
		set_error("no device occupancy grid (pp_map_set_cells / pp_map_rasterize_segments / pp_map_upload_occupancy first)");
		return PP_ERR_INVALID;
	}
	if (mode != PP_GVD_EXACT_EDT && mode != PP_GVD_REFERENCE_ORDER) {
		set_error("unknown GVD update mode");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipStream_t s = map->ctx->stream;
	const int rows = map->desc.rows, cols = map->desc.cols;
	const size_t n = map->cells();
	if (rows > 32767 || cols > 32767) {
		set_error("grids beyond 32767 cells a side are not supported (squared distances are int32, as in the reference)");
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
	long long work = 0;
	if (mode == PP_GVD_REFERENCE_ORDER) {
		// ObstacleDistanceMap::Update + CheckVoro + VoronoiDistanceMap::Update in the reference's own order (host, pp_brushfire_host.hpp)
		if (int rc = update_reference_order(map))
			return rc;
		work = map->gvdRef->pops;
	} else {
		// ---- ObstacleDistanceMap::Update as an exact transform
		PP_HIP_TRY(launch_edt(map, map->occ32, nullptr, map->obstLabel, map->d2));
		// ---- Voronoi edges (CheckVoro on the final labels) + VoronoiDistanceMap::Update
		hipLaunchKernelGGL(k_voronoi_edges, dim3(grid), dim3(kBlock), 0, s, rows, cols, map->obstLabel[0], map->occ32, map->voroEdge);
		PP_HIP_TRY(hipGetLastError());
		PP_HIP_TRY(launch_edt(map, nullptr, map->voroEdge, map->voroLabel, map->voroD2));
		work = 2;
	}
	map->obstResult = map->voroResult = 0;
	// ---- PathCostMap::Update, then what the validator reads: float distances and the validity bitmap
	hipLaunchKernelGGL(k_path_cost, dim3(grid), dim3(kBlock), 0, s, (int64_t)n, map->d2, map->voroD2, map->desc.resolution, alpha, d_max, map->pathcost);
	PP_HIP_TRY(hipGetLastError());
	PP_HIP_TRY(pph::launch_d2_to_distance(s, map->d2, map->dist, (int64_t)n, map->desc.resolution));
	PP_HIP_TRY(pph::launch_valid_bits(s, map->dist, (int64_t)n, map->minSafeRadius, map->validBits));
	PP_HIP_TRY(hipStreamSynchronize(s));
	if (iterations_out)
		*iterations_out = (int32_t)(work > INT_MAX ? INT_MAX : work);
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
