// Batched per-pose / per-path kernels of the hot path (gfx950):
//   check_states      a1  IsStateValid                (HBM-bound stream + cached gather)
//   check_arcs        a2+a3 IsPathValid over constant-steer arcs
//   check_segments    a2  IsPathValid over R2 segments (RRT edge check)
//   rollout_children  a4  GetConstantSteerChild for parents x primitives
//   rs_solve          a6  Reeds-Shepp GetOptimalPath
//   nonholo_build     a10 NonHolonomicHeuristic::Build
//   knn               a14 exact k nearest neighbours (flann replacement)
#include "pp_internal.hpp"

#include <mutex>

#include <cstdlib>
#include "pp_rs_device.hpp"

using namespace ppd;

namespace {

constexpr int kBlock = 256;

inline int grid_for(int64_t n, int block, int maxBlocks = 256 * 16)
{
	int64_t b = (n + block - 1) / block;
	if (b < 1)
		b = 1;
	if (b > maxBlocks)
		b = maxBlocks;
	return (int)b;
}

// ---------------------------------------------------------------- map prep --
__global__ void k_d2_to_distance(const int32_t* __restrict__ d2, float* __restrict__ dist, int64_t n, float res)
{
	// GVD::ObstacleDistanceMap::GetDistanceToNearestObstacle, gvd.h:38:
	// std::sqrt(int) -> double sqrt; times the float resolution in double; returned as float.
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
		dist[i] = (float)(sqrt((double)d2[i]) * (double)res);
}

__global__ void k_occ_to_u8(const int32_t* __restrict__ occ, uint8_t* __restrict__ occ8, int64_t n)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
		occ8[i] = occ[i] >= 0 ? 1 : 0; // obstacle_list_occupancy_map.cpp:63-69
}

// ------------------------------------------------------------ check_states --
// One pose per thread.  A block stages its 256 poses (6 KiB, contiguous in the
// Pose2d AoS layout) through LDS with 16-byte coalesced loads, so HBM sees whole
// lines once; the distance grid is a cached gather (4 MiB at 1024^2: L2 resident).
#ifndef PP_CS_GRID_PER_CU
#define PP_CS_GRID_PER_CU 16 // workgroups per CU of the grid-stride launch
#endif
#ifndef PP_CS_PER
#define PP_CS_PER 4
#endif
constexpr int kCsPer = PP_CS_PER; // poses per thread and tile: 4 x 256 x 24 B = 24 KiB in flight per workgroup before the barrier
__global__ void __launch_bounds__(kBlock) k_check_states(MapView m, int64_t n, const double* __restrict__ poses, uint8_t* __restrict__ valid, int aligned16)
{
	constexpr int kTile = kBlock * kCsPer;
	__shared__ double2 tile[kTile * 3 / 2];
	const int64_t nTiles = (n + kTile - 1) / kTile;
	for (int64_t tileIdx = blockIdx.x; tileIdx < nTiles; tileIdx += gridDim.x) {
		const int64_t base = tileIdx * kTile;
		const int count = (int)min((int64_t)kTile, n - base);
		const double* src = poses + base * 3;
		const int nd = count * 3; // doubles in this tile
		if (count == kTile && aligned16) {
			const double2* src2 = reinterpret_cast<const double2*>(src); // base*24 bytes is 16-byte aligned
			double2 v[kCsPer * 3 / 2];
#pragma unroll
			for (int k = 0; k < kCsPer * 3 / 2; k++)
				v[k] = src2[k * kBlock + threadIdx.x]; // all loads of the tile are in flight together
#pragma unroll
			for (int k = 0; k < kCsPer * 3 / 2; k++)
				tile[k * kBlock + threadIdx.x] = v[k];
		} else {
			double* t = reinterpret_cast<double*>(tile);
			for (int i = threadIdx.x; i < nd; i += kBlock)
				t[i] = src[i];
		}
		__syncthreads();
#pragma unroll
		for (int k = 0; k < kCsPer; k++) {
			const int i = k * kBlock + threadIdx.x;
			if (i < count) {
				const double* t = reinterpret_cast<const double*>(tile) + 3 * i;
				valid[base + i] = is_state_valid_bit(m, t[0], t[1], t[2]) ? 1 : 0;
			}
		}
		__syncthreads();
	}
}

// The pipelined form for large aligned batches (16-byte aligned poses, 4-byte aligned flags; tails and unaligned views go
// through the kernel above).  Same LDS staging -- a thread-per-group layout without it (each lane reading its own 96 bytes)
// was measured at HALF the rate: a wave's 16-byte loads then spread over 48 cache lines instead of 8; staging per wave without
// workgroup barriers measured 10 % slower -- with four changes: (0) a branch-free validity predicate (is_state_valid_bit_flat:
// the early exits and wrap loops of four inlined checks were 36 exec-mask branches per tile);
// (1) the NEXT tile's loads are issued into registers before the current tile is evaluated, so a workgroup always has 24 KiB
// in flight instead of alternating between a load phase and a compute phase; (2) the pose stream is loaded non-temporally
// (read once; the 128 KiB validity bitmap is what should stay cached); (3) the four flags of a thread's column go through LDS
// and leave as ONE 32-bit word per thread: 256 contiguous bytes per wave store instead of 64.
typedef double dvec2 __attribute__((ext_vector_type(2))); // (the non-temporal builtins take native vectors, not HIP's double2 struct)
__global__ void __launch_bounds__(kBlock) k_check_states_pipe(MapView m, int64_t nTiles, const dvec2* __restrict__ src, uint32_t* __restrict__ valid4)
{
	constexpr int kTile = kBlock * kCsPer;          // poses per tile
	constexpr int kVec = kCsPer * 3 / 2;            // 16-byte vectors per thread and tile
	static_assert(kCsPer == 4, "four flags per 32-bit word");
	__shared__ dvec2 tile[kTile * 3 / 2];
	__shared__ uint8_t flags[2][kTile];
	dvec2 v[kVec];
	int64_t tileIdx = blockIdx.x;
	if (tileIdx < nTiles) {
		const dvec2* s2 = src + tileIdx * (kTile * 3 / 2);
#pragma unroll
		for (int k = 0; k < kVec; k++)
			v[k] = __builtin_nontemporal_load(&s2[k * kBlock + threadIdx.x]);
	}
	int par = 0;
	for (; tileIdx < nTiles; tileIdx += gridDim.x, par ^= 1) {
#pragma unroll
		for (int k = 0; k < kVec; k++)
			tile[k * kBlock + threadIdx.x] = v[k];
		__syncthreads();
		const int64_t next = tileIdx + gridDim.x;
		if (next < nTiles) { // in flight while this tile is evaluated
			const dvec2* s2 = src + next * (kTile * 3 / 2);
#pragma unroll
			for (int k = 0; k < kVec; k++)
				v[k] = __builtin_nontemporal_load(&s2[k * kBlock + threadIdx.x]);
		}
#pragma unroll
		for (int k = 0; k < kCsPer; k++) {
			const int i = k * kBlock + threadIdx.x;
			const double* t = reinterpret_cast<const double*>(tile) + 3 * i;
			flags[par][i] = is_state_valid_bit_flat(m, t[0], t[1], t[2], m.validBits) ? 1 : 0;
		}
		__syncthreads(); // flags complete; the tile may be overwritten by the next iteration
		__builtin_nontemporal_store(reinterpret_cast<const uint32_t*>(flags[par])[threadIdx.x], &valid4[tileIdx * (kTile / 4) + threadIdx.x]);
	}
}

// With the pose stream at the read-only rate, what is left is the bitmap look-up: 2^26 random 4-byte gathers through the
// 32 KiB vector L1 are L2 round trips, 0.13 of the kernel's 0.42 ms (measured by pinning every look-up to word 0).  A bitmap of up
// to 128 KiB (grids up to 1024 x 1024) fits in a CU's 160 KiB of LDS next to a 24 KiB pose tile: ONE workgroup of 1024 threads
// per CU copies it in once and every look-up is a ds_read.  Two tiles of loads stay in flight in registers.
constexpr int kLdsBlock = 1024;
constexpr int kLdsTile = kLdsBlock; // one pose per thread and tile
__global__ void __launch_bounds__(kLdsBlock) k_check_states_lds(MapView m, int64_t nTiles, const dvec2* __restrict__ src, uint32_t* __restrict__ valid4, int bitmapWords)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	uint32_t* const bits = reinterpret_cast<uint32_t*>(lds);
	dvec2* const tile = reinterpret_cast<dvec2*>(lds + (size_t)bitmapWords * 4);
	uint8_t* const flags = reinterpret_cast<uint8_t*>(tile + kLdsTile * 3 / 2); // [2][kLdsTile]
	const int tid = threadIdx.x;
	for (int i = tid; i < bitmapWords / 4; i += kLdsBlock)
		reinterpret_cast<uint4*>(bits)[i] = reinterpret_cast<const uint4*>(m.validBits)[i];
	constexpr int kVecPerTile = kLdsTile * 3 / 2; // 1536 16-byte vectors: every thread loads one, the first half a second one
	const bool second = tid < kVecPerTile - kLdsBlock;
	dvec2 a0 = {}, b0 = {}, a1 = {}, b1 = {};
	int64_t t0 = blockIdx.x, t1 = t0 + gridDim.x;
	auto fetch = [&](int64_t tileIdx, dvec2& a, dvec2& b) {
		if (tileIdx < nTiles) {
			const dvec2* s2 = src + tileIdx * kVecPerTile;
			a = __builtin_nontemporal_load(&s2[tid]);
			if (second)
				b = __builtin_nontemporal_load(&s2[kLdsBlock + tid]);
		}
	};
	fetch(t0, a0, b0);
	fetch(t1, a1, b1);
	int par = 0;
	for (int64_t tileIdx = blockIdx.x; tileIdx < nTiles; tileIdx += gridDim.x, par ^= 1) {
		tile[tid] = a0;
		if (second)
			tile[kLdsBlock + tid] = b0;
		__syncthreads(); // (the first pass also completes the bitmap copy)
		a0 = a1;
		b0 = b1;
		fetch(tileIdx + 2 * (int64_t)gridDim.x, a1, b1); // two tiles ahead
		const double* t = reinterpret_cast<const double*>(tile) + 3 * tid;
		flags[par * kLdsTile + tid] = is_state_valid_bit_flat(m, t[0], t[1], t[2], bits) ? 1 : 0;
		__syncthreads();
		if (tid < kLdsTile / 4)
			__builtin_nontemporal_store(reinterpret_cast<const uint32_t*>(flags + par * kLdsTile)[tid], &valid4[tileIdx * (kLdsTile / 4) + tid]);
	}
}

/// validity bitmap: one ballot per wave = two 32-bit words
__global__ void __launch_bounds__(kBlock) k_valid_bits(const float* __restrict__ dist, int64_t cells, float minSafeRadius, uint32_t* __restrict__ bits)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const bool ok = i < cells && dist[i] >= minSafeRadius; // the comparison of state_validator_occupancy_map.cpp:25
	const unsigned long long b = __ballot(ok);
	const int lane = threadIdx.x & 63;
	const int64_t w = (i - lane) >> 5; // first word of this wave (cells padded to a multiple of 64 by the allocation)
	if (lane == 0)
		bits[w] = (uint32_t)b;
	if (lane == 32)
		bits[w + 1] = (uint32_t)(b >> 32);
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}
__device__ __forceinline__ double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

// Fused form: pose i = lb + (ub - lb) * u01(splitmix64(seed + 3i + c)); only a count leaves.
__global__ void __launch_bounds__(kBlock) k_check_states_fused(MapView m, int64_t n, uint64_t seed, unsigned long long* __restrict__ count)
{
	unsigned long long local = 0;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const uint64_t k = seed + 3ull * (uint64_t)i;
		double x = m.lbx + (m.ubx - m.lbx) * u01(splitmix64(k));
		double y = m.lby + (m.uby - m.lby) * u01(splitmix64(k + 1));
		double t = m.lbt + (m.ubt - m.lbt) * u01(splitmix64(k + 2));
		local += is_state_valid_bit(m, x, y, t) ? 1 : 0;
	}
	// wave reduce (64 lanes), then one atomic per wave
	for (int off = 32; off > 0; off >>= 1)
		local += __shfl_down(local, off, 64);
	if ((threadIdx.x & 63) == 0 && local)
		atomicAdd(count, local);
}

// -------------------------------------------------------------- check_arcs --
__global__ void __launch_bounds__(kBlock) k_check_arcs(MapView m, int64_t n, const double* __restrict__ from, const double* __restrict__ kappa,
	const double* __restrict__ length, const int32_t* __restrict__ dir, uint8_t* __restrict__ valid, float* __restrict__ last)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		Arc a;
		a.init = { from[3 * i], from[3 * i + 1], from[3 * i + 2] };
		a.kappa = kappa[i];
		a.length = length[i];
		a.backward = dir[i] == 1;
		float l = -1.0f;
		int checks = 0;
		bool ok = is_path_valid(m, a, a.init, l, checks);
		valid[i] = ok ? 1 : 0;
		if (last)
			last[i] = l;
	}
}

__global__ void __launch_bounds__(kBlock) k_check_segments(MapView m, int64_t n, const double* __restrict__ from, const double* __restrict__ to,
	uint8_t* __restrict__ valid)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		Segment sg;
		sg.x0 = from[2 * i];
		sg.y0 = from[2 * i + 1];
		sg.x1 = to[2 * i];
		sg.y1 = to[2 * i + 1];
		const double dx = sg.x1 - sg.x0, dy = sg.y1 - sg.y0;
		sg.length = sqrt(dx * dx + dy * dy); // PathR2 ctor: (to - from).norm(), paths/path_r2.cpp:5-9
		Pose init = { sg.x0, sg.y0, 0.0 };
		float l;
		int checks = 0;
		valid[i] = is_path_valid(m, sg, init, l, checks) ? 1 : 0;
	}
}

// -------------------------------------------------------- rollout_children --
// thread = (parent, primitive).  GetConstantSteerChild, algo/hybrid_a_star.cpp:111-147.
__global__ void __launch_bounds__(kBlock) k_rollout(MapView m, pph::RolloutParams rp, pph::PrimTable prims, int64_t nParents, const double* __restrict__ parents,
	uint8_t* __restrict__ valid, double* __restrict__ pose, int32_t* __restrict__ key, double* __restrict__ cost, double* __restrict__ lengthOut)
{
	const int P = prims.n;
	const int64_t total = nParents * P;
	for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
		const int64_t parent = idx / P;
		const int prim = (int)(idx - parent * P);
		Arc a;
		a.init = { parents[3 * parent], parents[3 * parent + 1], parents[3 * parent + 2] };
		a.kappa = prims.kappa[prim];
		a.length = rp.arcLength;
		a.backward = prims.backward[prim];
		int pix, piy, pit;
		discretize_pose(a.init, rp.lat, rp.headingAlias, pix, piy, pit);
		Pose child = a.interpolate(1.0);
		int ix, iy, it;
		discretize_pose(child, rp.lat, rp.headingAlias, ix, iy, it);
		float lastValidRatio;
		int checks = 0;
		bool ok = true;
		if (!is_path_valid(m, a, a.init, lastValidRatio, checks)) {
			// PathConstantSteer::Truncate, paths/path_constant_steer.cpp:16-20
			child = a.interpolate((double)lastValidRatio);
			a.length *= (double)lastValidRatio;
			discretize_pose(child, rp.lat, rp.headingAlias, ix, iy, it);
			if (ix == pix && iy == piy && it == pit)
				ok = false;
		}
		double c = 0.0;
		if (ok) {
			double pathCost = (a.backward ? rp.reverseMult : rp.forwardMult) * a.length;
			double switchingCost = 0.0; // hybrid_a_star.cpp:142 compares a direction with itself
			double voro = voronoi_cost(m, a, rp.voroDiagRes, rp.voronoiMult);
			c = pathCost + switchingCost + voro;
		}
		valid[idx] = ok ? 1 : 0;
		pose[3 * idx] = child.x;
		pose[3 * idx + 1] = child.y;
		pose[3 * idx + 2] = child.t;
		key[3 * idx] = ix;
		key[3 * idx + 1] = iy;
		key[3 * idx + 2] = it;
		cost[idx] = c;
		lengthOut[idx] = ok ? a.length : 0.0;
	}
}

// ---------------------------------------------------------------- rs_solve --
__global__ void __launch_bounds__(kBlock) k_rs_solve(int64_t n, const double* __restrict__ from, const double* __restrict__ to, double rmin, float rev, float fwd,
	float sw, int32_t* __restrict__ word, double* __restrict__ tuv, float* __restrict__ cost, double* __restrict__ segLength)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		Pose a = { from[3 * i], from[3 * i + 1], from[3 * i + 2] };
		Pose b = { to[3 * i], to[3 * i + 1], to[3 * i + 2] };
		double t, u, v, sl;
		float c;
		int w = rs::optimal_word(a, b, rmin, rev, fwd, sw, t, u, v, c, sl);
		word[i] = w;
		if (tuv) {
			tuv[3 * i] = t;
			tuv[3 * i + 1] = u;
			tuv[3 * i + 2] = v;
		}
		if (cost)
			cost[i] = c;
		if (segLength)
			segLength[i] = sl;
	}
}

// ----------------------------------------------------------- nonholo_build --
// NonHolonomicHeuristic::Build, algo/heuristics.cpp:62-73: entry (i, j, k) = float cost of the
// optimal RS path from (i*res - offX, j*res - offY, k*angRes) to the origin.
__global__ void __launch_bounds__(kBlock) k_nonholo_build(pph::NonHoloDesc d, double* __restrict__ table)
{
	const int64_t total = (int64_t)d.nx * d.ny * d.na;
	for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
		const int k = (int)(idx % d.na);
		const int64_t ij = idx / d.na;
		const int j = (int)(ij % d.ny);
		const int i = (int)(ij / d.ny);
		Pose p;
		p.x = i * d.spatialRes - d.offX;
		p.y = j * d.spatialRes - d.offY;
		p.t = wrap_theta(k * d.angularRes); // Pose2d constructor
		Pose g = { 0.0, 0.0, 0.0 };
		double t, u, v, sl;
		float c;
		int w = rs::optimal_word(p, g, d.rmin, d.reverseCost, d.forwardCost, d.switchCost, t, u, v, c, sl);
		// GetOptimalPath returns an empty PathSegment when no word is valid; its ComputeCost is +inf
		table[idx] = w < 0 ? (double)__builtin_huge_valf() : (double)c;
	}
}

// --------------------------------------------------------------------- knn --
// One query per thread; points streamed through LDS tiles shared by the block.
// Squared L2 in double, ascending; ties keep the lower point index.
constexpr int kKnnMaxK = 16;
constexpr int kKnnTile = 1024;
__global__ void __launch_bounds__(kBlock) k_knn(int64_t nPoints, const double* __restrict__ pts, int64_t nQueries, const double* __restrict__ q, int k,
	int32_t* __restrict__ idxOut, double* __restrict__ d2Out)
{
	__shared__ double2 tile[kKnnTile];
	const int64_t qi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	double qx = 0, qy = 0;
	if (qi < nQueries) {
		qx = q[2 * qi];
		qy = q[2 * qi + 1];
	}
	double bd[kKnnMaxK];
	int bi[kKnnMaxK];
#pragma unroll
	for (int s = 0; s < kKnnMaxK; s++) {
		bd[s] = __builtin_huge_val();
		bi[s] = -1;
	}
	for (int64_t base = 0; base < nPoints; base += kKnnTile) {
		const int cnt = (int)min((int64_t)kKnnTile, nPoints - base);
		for (int i = threadIdx.x; i < cnt; i += kBlock)
			tile[i] = reinterpret_cast<const double2*>(pts)[base + i];
		__syncthreads();
		if (qi < nQueries) {
			for (int i = 0; i < cnt; i++) {
				const double dx = tile[i].x - qx, dy = tile[i].y - qy;
				const double d = dx * dx + dy * dy;
				if (d < bd[k - 1]) {
					// insert keeping ascending order; equal distances stay behind earlier indices
					double cd = d;
					int ci = (int)(base + i);
					bool inserted = false;
#pragma unroll
					for (int s = 0; s < kKnnMaxK; s++) {
						if (s < k && (inserted || cd < bd[s])) {
							double td = bd[s];
							int ti = bi[s];
							bd[s] = cd;
							bi[s] = ci;
							cd = td;
							ci = ti;
							inserted = true;
						}
					}
				}
			}
		}
		__syncthreads();
	}
	if (qi < nQueries) {
		for (int s = 0; s < k; s++) {
			idxOut[qi * k + s] = bi[s];
			d2Out[qi * k + s] = bd[s];
		}
	}
}

} // namespace

namespace pph {

hipError_t launch_d2_to_distance(hipStream_t s, const int32_t* d2, float* dist, int64_t n, float res)
{
	hipLaunchKernelGGL(k_d2_to_distance, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, d2, dist, n, res);
	return hipGetLastError();
}
hipError_t launch_occ_to_u8(hipStream_t s, const int32_t* occ, uint8_t* occ8, int64_t n)
{
	hipLaunchKernelGGL(k_occ_to_u8, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, occ, occ8, n);
	return hipGetLastError();
}
hipError_t launch_check_states(hipStream_t s, const MapView& m, int64_t n, const double* poses, uint8_t* valid)
{
	if (n <= 0)
		return hipSuccess;
	static const bool staged = getenv("PP_CS_STAGED") && getenv("PP_CS_STAGED")[0] == '1'; // measurement switch: the round-1 kernel for everything
	int64_t done = 0;
	constexpr int kTile = kBlock * kCsPer;
	const int64_t cells = (int64_t)m.rows * m.cols;
	const int bitmapWords = (int)(((cells + 63) / 64) * 2);
	static const int ldsMode = getenv("PP_CS_LDS") ? atoi(getenv("PP_CS_LDS")) : 1; // measurement switch: 0 = never use the LDS-resident bitmap
	if (!staged && ldsMode && (((uintptr_t)poses) & 15) == 0 && (((uintptr_t)valid) & 3) == 0 && bitmapWords * 4 <= 128 * 1024 && bitmapWords % 4 == 0 && n >= (1 << 20)) {
		const size_t ldsBytes = (size_t)bitmapWords * 4 + (size_t)kLdsTile * 24 + 2 * kLdsTile;
		// more than 64 KiB of LDS per workgroup has to be asked for, once PER DEVICE (the attribute belongs to the function on a
		// device): a flag and the CU count per device id, set under a lock
		int dev = 0;
		(void)hipGetDevice(&dev);
		static std::mutex attrLock;
		static int cusOf[64] = {}; // 0: attribute not yet set on that device
		int cus = 0;
		{
			std::lock_guard<std::mutex> g(attrLock);
			const int slot = dev >= 0 && dev < 64 ? dev : 63;
			if (!cusOf[slot] || slot == 63) {
				hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_check_states_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
				if (e != hipSuccess)
					return e;
				hipDeviceProp_t prop;
				cusOf[slot] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
			}
			cus = cusOf[slot];
		}
		const int64_t tiles = n / kLdsTile;
		hipLaunchKernelGGL(k_check_states_lds, dim3((unsigned)(tiles < cus ? tiles : cus)), dim3(kLdsBlock), ldsBytes, s, m, tiles, reinterpret_cast<const dvec2*>(poses),
			reinterpret_cast<uint32_t*>(valid), bitmapWords);
		done = tiles * kLdsTile;
	} else if (!staged && (((uintptr_t)poses) & 15) == 0 && (((uintptr_t)valid) & 3) == 0 && n >= 64 * kTile) {
		const int64_t tiles = n / kTile;
		hipLaunchKernelGGL(k_check_states_pipe, dim3(grid_for(tiles, 1, 256 * PP_CS_GRID_PER_CU)), dim3(kBlock), 0, s, m, tiles, reinterpret_cast<const dvec2*>(poses),
			reinterpret_cast<uint32_t*>(valid));
		done = tiles * kTile;
	}
	if (done < n) { // the last partial tile, small batches, or views the pipelined form cannot take (done * 24 bytes keeps the 16-byte alignment)
		const double* p = poses + 3 * done;
		hipLaunchKernelGGL(k_check_states, dim3(grid_for(n - done, kBlock * kCsPer, 256 * PP_CS_GRID_PER_CU)), dim3(kBlock), 0, s, m, n - done, p, valid + done,
			(int)((((uintptr_t)p) & 15) == 0));
	}
	return hipGetLastError();
}
hipError_t launch_valid_bits(hipStream_t s, const float* dist, int64_t cells, float minSafeRadius, uint32_t* bits)
{
	if (cells <= 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_valid_bits, dim3((unsigned)((cells + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, dist, cells, minSafeRadius, bits);
	return hipGetLastError();
}
hipError_t launch_check_states_fused(hipStream_t s, const MapView& m, int64_t n, uint64_t seed, uint64_t* count)
{
	hipError_t e = hipMemsetAsync(count, 0, sizeof(uint64_t), s);
	if (e != hipSuccess)
		return e;
	if (n <= 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_check_states_fused, dim3(grid_for(n, kBlock, 256 * 8)), dim3(kBlock), 0, s, m, n, seed, (unsigned long long*)count);
	return hipGetLastError();
}
hipError_t launch_check_arcs(hipStream_t s, const MapView& m, int64_t n, const double* from, const double* kappa, const double* length, const int32_t* dir,
	uint8_t* valid, float* last)
{
	if (n <= 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_check_arcs, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, m, n, from, kappa, length, dir, valid, last);
	return hipGetLastError();
}
hipError_t launch_check_segments(hipStream_t s, const MapView& m, int64_t n, const double* from, const double* to, uint8_t* valid)
{
	if (n <= 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_check_segments, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, m, n, from, to, valid);
	return hipGetLastError();
}
hipError_t launch_rollout(hipStream_t s, const MapView& m, const RolloutParams& rp, const PrimTable& prims, int64_t nParents, const double* parents,
	uint8_t* valid, double* pose, int32_t* key, double* cost, double* length)
{
	if (nParents <= 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_rollout, dim3(grid_for(nParents * prims.n, kBlock)), dim3(kBlock), 0, s, m, rp, prims, nParents, parents, valid, pose, key, cost, length);
	return hipGetLastError();
}
hipError_t launch_rs_solve(hipStream_t s, int64_t n, const double* from, const double* to, double rmin, float rev, float fwd, float sw, int32_t* word, double* tuv,
	float* cost, double* segLength)
{
	if (n <= 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_rs_solve, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, from, to, rmin, rev, fwd, sw, word, tuv, cost, segLength);
	return hipGetLastError();
}
hipError_t launch_nonholo_build(hipStream_t s, const NonHoloDesc& d, double* table)
{
	const int64_t total = (int64_t)d.nx * d.ny * d.na;
	hipLaunchKernelGGL(k_nonholo_build, dim3(grid_for(total, kBlock)), dim3(kBlock), 0, s, d, table);
	return hipGetLastError();
}
hipError_t launch_knn(hipStream_t s, int64_t nPoints, const double* pts, int64_t nQueries, const double* q, int k, int32_t* idx, double* d2)
{
	if (nQueries <= 0)
		return hipSuccess;
	if (k < 1 || k > kKnnMaxK)
		return hipErrorInvalidValue;
	int64_t blocks = (nQueries + kBlock - 1) / kBlock;
	hipLaunchKernelGGL(k_knn, dim3((unsigned)blocks), dim3(kBlock), 0, s, nPoints, pts, nQueries, q, k, idx, d2);
	return hipGetLastError();
}

} // namespace pph
