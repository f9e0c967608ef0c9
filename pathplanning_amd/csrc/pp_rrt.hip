// rrt_tree -- RRT::SearchPath (algo/rrt.h:55-95) and RRTStar::SearchPath (algo/rrt_star.h:53-112)
// with Tree::GetNearestNode(s) (utils/tree.h:73-116, flann exact kNN) on gfx950.
//
// The planners are strictly sequential (sample i sees the tree after i-1 insertions), so one
// workgroup owns one tree and runs the whole loop on the device: no host round trip per sample.
//   * the sample stream is the query's own mt19937_64 (utils/random.h), drawn on the device;
//   * nearest / k-nearest = block-wide brute-force scan of the tree's points (double2, coalesced;
//     1.6 MB at 1e5 nodes: L2 resident), squared L2 in double, ties -> lower node index;
//   * SteerTowards / SteerExactly / PathR2 (rrt_star.h:143-160, paths/path_r2.cpp) and the edge
//     check (IsPathValid on the occupancy validator with theta = 0, or StateValidatorFree) are
//     evaluated by the lanes, the reference's sequential choose-parent scan by thread 0.
// Independent problems (different seeds / start-goal pairs) map to different workgroups.
#include "pp_search_device.hpp"

#include <algorithm>
#include <cstring>
#include <memory>

using namespace ppd;

namespace {

constexpr int RT = 512;          // threads per tree
constexpr int kMaxNear = 16;     // k = max(1, (unsigned)log(N)) <= 16 up to N = e^17 = 2.4e7 nodes

struct RrtArgs {
	MapView m;
	int useMap;        // 0 = StateValidatorFree
	double lbx, lby, ubx, uby;
	unsigned int maxIteration, maxNumberTreeNode;
	double maxConnectionDistance, goalBias;
	double initx, inity, goalx, goaly;
	unsigned long long seed;
	int star;
	int capacity;      // allocated nodes
};

struct RrtOut {
	int32_t status, nNodes, solution, pad;
	long long iterations, nKnn, nEdge;
};

struct Cand {
	double d;
	int idx;
};
__device__ __forceinline__ bool cand_before(double d1, int i1, double d2, int i2) { return d1 < d2 || (d1 == d2 && i1 < i2); }

/// block-wide arg-min of (d, idx); result broadcast through LDS
__device__ __forceinline__ void block_argmin(double& d, int& idx, double* sd, int* si)
{
	for (int off = 32; off > 0; off >>= 1) {
		const double od = __shfl_xor(d, off, 64);
		const int oi = __shfl_xor(idx, off, 64);
		if (cand_before(od, oi, d, idx)) {
			d = od;
			idx = oi;
		}
	}
	const int wave = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 0) {
		sd[wave] = d;
		si[wave] = idx;
	}
	__syncthreads();
	double bd = sd[0];
	int bi = si[0];
	for (int w = 1; w < RT / 64; w++)
		if (cand_before(sd[w], si[w], bd, bi)) {
			bd = sd[w];
			bi = si[w];
		}
	__syncthreads();
	d = bd;
	idx = bi;
}

__device__ __forceinline__ bool edge_valid(const RrtArgs& A, double x0, double y0, double x1, double y1, double length)
{
	if (!A.useMap)
		return true; // StateValidatorFree::IsPathValid, state_validator_free.h:24-29
	Segment sg;
	sg.x0 = x0;
	sg.y0 = y0;
	sg.x1 = x1;
	sg.y1 = y1;
	sg.length = length;
	Pose init = { x0, y0, 0.0 };
	float l;
	int checks = 0;
	return is_path_valid(A.m, sg, init, l, checks);
}

__global__ void __launch_bounds__(RT) k_rrt(RrtArgs A, double2* __restrict__ pts, int32_t* __restrict__ parent, double* __restrict__ cost, RrtOut* __restrict__ out)
{
	__shared__ unsigned long long mt[Mt64::N];
	__shared__ double s_d[RT / 64];
	__shared__ int s_i[RT / 64];
	__shared__ double nearD[kMaxNear];
	__shared__ int nearI[kMaxNear];
	__shared__ double candCost[kMaxNear];
	__shared__ uint8_t candValid[kMaxNear];
	__shared__ int s_best, s_flag;
	__shared__ double s_bestCost;

	const int tid = threadIdx.x;
	// the RNG helpers are written for a 64-lane block: the first wave drives them
	if (tid == 0)
		Mt64::seed(mt, A.seed);
	int mtIdx = Mt64::N;
	if (tid == 0) {
		pts[0] = make_double2(A.initx, A.inity); // Tree::CreateRootNode, tree.h:59-65
		parent[0] = -1;
		cost[0] = 0.0;
	}
	__syncthreads();
	int n = 1;
	long long iterations = 0, nKnn = 0, nEdge = 0;
	int status = -1, solution = -1;

	auto draw = [&](double lb, double ub) -> double {
		// Random<double>::SampleUniform, utils/random.h:23-27 (block-uniform result)
		if (mtIdx >= Mt64::N) {
			// in-place regeneration, all threads take part in the barriers
			const unsigned long long UM = 0xFFFFFFFF80000000ull, LM = 0x7FFFFFFFull, AA = 0xB5026F5AA96619E9ull;
			unsigned long long v = 0;
			if (tid < Mt64::M) {
				const unsigned long long x = (mt[tid] & UM) | (mt[tid + 1] & LM);
				v = mt[tid + Mt64::M] ^ (x >> 1) ^ ((x & 1ull) ? AA : 0ull);
			}
			__syncthreads();
			if (tid < Mt64::M)
				mt[tid] = v;
			__syncthreads();
			if (tid >= Mt64::M && tid < Mt64::N - 1) {
				const unsigned long long x = (mt[tid] & UM) | (mt[tid + 1] & LM);
				v = mt[tid - Mt64::M] ^ (x >> 1) ^ ((x & 1ull) ? AA : 0ull);
			}
			__syncthreads();
			if (tid >= Mt64::M && tid < Mt64::N - 1)
				mt[tid] = v;
			__syncthreads();
			if (tid == 0) {
				const unsigned long long x = (mt[Mt64::N - 1] & UM) | (mt[0] & LM);
				mt[Mt64::N - 1] = mt[Mt64::M - 1] ^ (x >> 1) ^ ((x & 1ull) ? AA : 0ull);
			}
			__syncthreads();
			mtIdx = 0;
		}
		const double u = Mt64::uniform01(Mt64::temper(mt[mtIdx]));
		mtIdx++;
		const double range = ub - lb;
		return lb + range * u;
	};

	int count = -1;
	while (true) {
		count++;
		if (count > (int)A.maxIteration) // rrt.h:62-65 / rrt_star.h:62-65
			break;
		if ((unsigned int)n > A.maxNumberTreeNode)
			break;
		if (n >= A.capacity)
			break;
		iterations++;
		// ---- sample: goal with probability goalBias, else uniform in the bounds (x then y)
		double rx, ry;
		if (draw(0, 1) < A.goalBias) {
			rx = A.goalx;
			ry = A.goaly;
		} else {
			rx = draw(A.lbx, A.ubx); // StateSpaceR2::SampleUniform, state_space_r2.cpp:25-35
			ry = draw(A.lby, A.uby);
		}
		// ---- nearest node (Tree::GetNearestNode)
		double bd = __builtin_huge_val();
		int bi = 0x7FFFFFFF;
		for (int i = tid; i < n; i += RT) {
			const double2 p = pts[i];
			const double dx = p.x - rx, dy = p.y - ry;
			const double d = dx * dx + dy * dy;
			if (cand_before(d, i, bd, bi)) {
				bd = d;
				bi = i;
			}
		}
		block_argmin(bd, bi, s_d, s_i);
		nKnn++;
		const int nearest = bi;
		const double2 pn = pts[nearest];
		// ---- SteerTowards, rrt_star.h:143-151 (PathR2 + Truncate, paths/path_r2.cpp)
		double fx = rx, fy = ry;
		double len;
		{
			const double dx = rx - pn.x, dy = ry - pn.y;
			len = sqrt(dx * dx + dy * dy);
			if (len > 0) {
				double ratio = A.maxConnectionDistance / len;
				ratio = ratio < 0.0 ? 0.0 : (ratio > 1.0 ? 1.0 : ratio); // std::clamp
				fx = (1 - ratio) * pn.x + ratio * rx;
				fy = (1 - ratio) * pn.y + ratio * ry;
				len *= ratio;
			}
		}
		nEdge++;
		if (tid == 0)
			s_flag = edge_valid(A, pn.x, pn.y, fx, fy, len) ? 1 : 0;
		__syncthreads();
		const bool ok = s_flag != 0;
		__syncthreads();
		if (!ok)
			continue;
		const double nx = fx, ny = fy;

		int newNode = -1;
		if (!A.star) {
			// ---- RRT: Extend(newState, nearestNode), rrt.h:80-82; an existing state returns its node (tree.h:127-129)
			double ed = __builtin_huge_val();
			int ei = 0x7FFFFFFF;
			for (int i = tid; i < n; i += RT) {
				const double2 p = pts[i];
				if (p.x == nx && p.y == ny && i < ei) {
					ed = 0.0;
					ei = i;
				}
			}
			block_argmin(ed, ei, s_d, s_i);
			if (ei != 0x7FFFFFFF) {
				newNode = ei;
			} else {
				newNode = n;
				if (tid == 0) {
					pts[n] = make_double2(nx, ny);
					parent[n] = nearest;
					cost[n] = 0.0;
				}
				n++;
				__syncthreads();
			}
			const double gx = nx - A.goalx, gy = ny - A.goaly;
			if (sqrt(gx * gx + gy * gy) < 1) { // RRT::IsSolution, rrt.h:125-128
				status = 0;
				solution = newNode;
				break;
			}
			continue;
		}

		// ---- RRT*: k nearest of the new state, k = max(1, (unsigned)log(size)), rrt_star.h:84-85
		unsigned int nnU = (unsigned int)log((double)(unsigned long long)n);
		int k = (int)(nnU < 1u ? 1u : nnU);
		if (k > kMaxNear)
			k = kMaxNear;
		if (k > n)
			k = n;
		nKnn++;
		// local top-k per thread (ascending), then k rounds of block arg-min over the heads
		double ld[kMaxNear];
		int li[kMaxNear];
		const int kk = k;
#pragma unroll
		for (int s = 0; s < kMaxNear; s++) {
			ld[s] = __builtin_huge_val();
			li[s] = 0x7FFFFFFF;
		}
		for (int i = tid; i < n; i += RT) {
			const double2 p = pts[i];
			const double dx = p.x - nx, dy = p.y - ny;
			double cd = dx * dx + dy * dy;
			int ci = i;
			if (cand_before(cd, ci, ld[kk - 1], li[kk - 1])) {
				bool ins = false;
#pragma unroll
				for (int s = 0; s < kMaxNear; s++) {
					if (s < kk && (ins || cand_before(cd, ci, ld[s], li[s]))) {
						const double td = ld[s];
						const int ti = li[s];
						ld[s] = cd;
						li[s] = ci;
						cd = td;
						ci = ti;
						ins = true;
					}
				}
			}
		}
		int head = 0;
		for (int r = 0; r < kk; r++) {
			double hd = __builtin_huge_val();
			int hi = 0x7FFFFFFF;
#pragma unroll
			for (int s = 0; s < kMaxNear; s++)
				if (s == head) {
					hd = ld[s];
					hi = li[s];
				}
			const int mine = hi;
			block_argmin(hd, hi, s_d, s_i);
			if (tid == 0) {
				nearD[r] = hd;
				nearI[r] = hi;
			}
			if (mine == hi && hi != 0x7FFFFFFF)
				head++;
			__syncthreads();
		}
		// ---- choose parent: lanes evaluate SteerExactly + IsPathValid of every candidate
		if (tid < kk) {
			const int node = nearI[tid];
			const double2 p = pts[node];
			const double dx = nx - p.x, dy = ny - p.y;
			const double plen = sqrt(dx * dx + dy * dy); // PathR2 length
			candCost[tid] = cost[node] + plen;
			candValid[tid] = edge_valid(A, p.x, p.y, nx, ny, plen) ? 1 : 0;
		}
		__syncthreads();
		if (tid == 0) {
			// the reference's sequential scan, rrt_star.h:89-97: IsPathValid only runs when cost < bestCost
			int best = -1;
			double bestCost = __builtin_huge_val();
			int edges = 0;
			for (int r = 0; r < kk; r++) {
				if (candCost[r] < bestCost) {
					edges++;
					if (candValid[r]) {
						best = nearI[r];
						bestCost = candCost[r];
					}
				}
			}
			s_best = best;
			s_bestCost = bestCost;
			s_flag = edges;
		}
		__syncthreads();
		nEdge += s_flag;
		const int bestParent = s_best;
		const double bestCost = s_bestCost;
		// ---- Extend(newState, bestParentNode), rrt_star.h:100-102 / tree.h:124-146
		const int nn0 = nearI[0];
		const double2 p0 = pts[nn0];
		if (p0.x == nx && p0.y == ny) {
			newNode = nn0; // already in the tree: its cost is overwritten (Q15)
		} else {
			newNode = n;
			if (tid == 0) {
				pts[n] = make_double2(nx, ny);
				parent[n] = bestParent >= 0 ? bestParent : nn0; // null parent -> nearest node (tree.h:131)
			}
			n++;
			if (bestParent < 0)
				nKnn++; // GetNearestNode inside Extend
		}
		if (tid == 0)
			cost[newNode] = bestCost;
		__syncthreads();
		if (nx == A.goalx && ny == A.goaly) { // RRTStar::IsSolution: exact equality, rrt_star.h:136-139
			status = 0;
			solution = newNode;
			break;
		}
	}
	if (tid == 0) {
		RrtOut o;
		o.status = status;
		o.nNodes = n;
		o.solution = solution;
		o.pad = 0;
		o.iterations = iterations;
		o.nKnn = nKnn;
		o.nEdge = nEdge;
		*out = o;
	}
}

} // namespace

struct pp_rrt {
	std::vector<double> nodes, costs, path;
	std::vector<int32_t> parents;
};

extern "C" {

int pp_rrt_run(pp_ctx* ctx, pp_map* map, const double lower[2], const double upper[2], const double params[4], const double init[2], const double goal[2],
	uint64_t seed, int32_t star, pp_rrt** out, pp_rrt_result* result)
{
	using pph::set_error;
	if (!ctx || !lower || !upper || !params || !init || !goal || !out || !result) {
		set_error("null argument");
		return PP_ERR_INVALID;
	}
	if (map && !map->dist) {
		set_error("distance grid not uploaded (pp_map_upload_dist2)");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(ctx->device));
	RrtArgs A;
	std::memset(&A, 0, sizeof(A));
	if (map)
		A.m = map->view();
	A.useMap = map ? 1 : 0;
	A.lbx = lower[0];
	A.lby = lower[1];
	A.ubx = upper[0];
	A.uby = upper[1];
	A.maxIteration = (unsigned int)params[0];
	A.maxNumberTreeNode = (unsigned int)params[1];
	A.maxConnectionDistance = params[2];
	A.goalBias = params[3];
	A.initx = init[0];
	A.inity = init[1];
	A.goalx = goal[0];
	A.goaly = goal[1];
	A.seed = seed;
	A.star = star;
	// the loop stops once size > maxNumberTreeNode or after maxIteration + 1 iterations
	const unsigned long long cap = std::min<unsigned long long>((unsigned long long)A.maxNumberTreeNode + 2ull, (unsigned long long)A.maxIteration + 3ull);
	if (cap > (1ull << 28)) {
		set_error("tree too large");
		return PP_ERR_INVALID;
	}
	A.capacity = (int)cap;
	double2* pts = nullptr;
	int32_t* parent = nullptr;
	double* cost = nullptr;
	RrtOut* dout = nullptr;
	hipError_t e = hipMalloc((void**)&pts, cap * sizeof(double2));
	if (e == hipSuccess)
		e = hipMalloc((void**)&parent, cap * 4);
	if (e == hipSuccess)
		e = hipMalloc((void**)&cost, cap * 8);
	if (e == hipSuccess)
		e = hipMalloc((void**)&dout, sizeof(RrtOut));
	RrtOut ho;
	std::memset(&ho, 0, sizeof(ho));
	auto r = std::make_unique<pp_rrt>();
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_rrt, dim3(1), dim3(RT), 0, ctx->stream, A, pts, parent, cost, dout);
		e = hipGetLastError();
	}
	if (e == hipSuccess)
		e = hipMemcpyAsync(&ho, dout, sizeof(RrtOut), hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess)
		e = hipStreamSynchronize(ctx->stream);
	if (e == hipSuccess) {
		const int n = ho.nNodes;
		r->nodes.resize((size_t)n * 2);
		r->parents.resize(n);
		r->costs.resize(n);
		e = hipMemcpy(r->nodes.data(), pts, (size_t)n * 16, hipMemcpyDeviceToHost);
		if (e == hipSuccess)
			e = hipMemcpy(r->parents.data(), parent, (size_t)n * 4, hipMemcpyDeviceToHost);
		if (e == hipSuccess)
			e = hipMemcpy(r->costs.data(), cost, (size_t)n * 8, hipMemcpyDeviceToHost);
	}
	(void)hipFree(pts);
	(void)hipFree(parent);
	(void)hipFree(cost);
	(void)hipFree(dout);
	if (e != hipSuccess)
		return pph::hip_fail(e, "pp_rrt_run");
	// GetPath, rrt.h:97-115: states from the root to the solution node
	if (ho.solution >= 0) {
		std::vector<int> chain;
		for (int k = ho.solution; k >= 0; k = r->parents[k])
			chain.push_back(k);
		for (size_t i = chain.size(); i-- > 0;) {
			r->path.push_back(r->nodes[2 * chain[i]]);
			r->path.push_back(r->nodes[2 * chain[i] + 1]);
		}
	}
	result->status = ho.status;
	result->n_nodes = ho.nNodes;
	result->n_path = (int32_t)(r->path.size() / 2);
	result->iterations = ho.iterations;
	result->n_knn_queries = ho.nKnn;
	result->n_edge_checks = ho.nEdge;
	*out = r.release();
	return PP_OK;
}

int pp_rrt_get(pp_rrt* r, double* nodes_xy, int32_t* parents, double* costs, double* path_xy)
{
	if (!r) {
		pph::set_error("null result");
		return PP_ERR_INVALID;
	}
	if (nodes_xy)
		std::memcpy(nodes_xy, r->nodes.data(), r->nodes.size() * 8);
	if (parents)
		std::memcpy(parents, r->parents.data(), r->parents.size() * 4);
	if (costs)
		std::memcpy(costs, r->costs.data(), r->costs.size() * 8);
	if (path_xy)
		std::memcpy(path_xy, r->path.data(), r->path.size() * 8);
	return PP_OK;
}

int pp_rrt_destroy(pp_rrt* r)
{
	delete r;
	return PP_OK;
}

} // extern "C"
