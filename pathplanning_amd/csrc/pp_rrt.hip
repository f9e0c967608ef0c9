// rrt_tree -- RRT::SearchPath (algo/rrt.h:55-95) and RRTStar::SearchPath (algo/rrt_star.h:53-112)
// with Tree::GetNearestNode(s) (utils/tree.h:73-116, flann exact kNN) on gfx950.
//
// The planners are strictly sequential (sample i sees the tree after i-1 insertions), so one
// workgroup owns one tree and runs the whole loop on the device: no host round trip per sample.
//   * the sample stream is the query's own mt19937_64 (utils/random.h), drawn on the device;
//   * nearest / k-nearest: exact, squared L2 in double, ties -> lower node index.  Small trees are scanned
//     by the whole block (double2, coalesced); beyond 2048 nodes a uniform grid over the bounds (linked cell
//     lists in HBM, ~4 nodes per cell at capacity) limits the scan to a square window of cells that doubles
//     until the k-th candidate is provably closer than anything outside it; the k best of the window's
//     candidates are picked by rank counting in LDS;
//   * SteerTowards / SteerExactly / PathR2 (rrt_star.h:143-160, paths/path_r2.cpp) and the edge
//     check (IsPathValid on the occupancy validator with theta = 0, or StateValidatorFree) are
//     evaluated by the lanes, the reference's sequential choose-parent scan by thread 0.
// Independent problems (different seeds / start-goal pairs) map to different workgroups (pp_rrt_run_batch).
//
// Beyond the reference: the reference's RRT* only chooses the best parent among the k = ln N nearest
// nodes -- "FIXME" at rrt_star.h:83; Node::Reparent exists (utils/node.h:203-225) but no planner calls it.  star = 2 / 3 add
// the missing half of RRT* (Karaman & Frazzoli): after the new node is linked, every near node that gets cheaper through it
// (and whose connecting edge is valid) is re-parented to it and the saving is carried down its subtree (children lists in
// HBM, level-synchronous walk by the block).  star = 3 also replaces the k-nearest near-set by a radius search:
// the <= 16 nearest nodes within gamma * sqrt(ln(n + 1) / (n + 1)).  Parity for these modes is against the oracle's own
// definition of the same steps (tests/test_gpu_rrt.py); with star = 1 nothing changes.
#include "pp_search_device.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

using namespace ppd;

namespace {

constexpr int RT = 512;          // threads per tree
constexpr int kMaxNear = 16;     // k = max(1, (unsigned)log(N)) <= 16 up to N = e^17 = 2.4e7 nodes
constexpr int kBruteMax = 2048;  // trees up to this size are scanned whole
constexpr int kCandMax = 2048;   // candidates of one grid window held in LDS

struct RrtArgs {
	MapView m;
	int useMap;        // 0 = StateValidatorFree
	double lbx, lby, ubx, uby;
	unsigned int maxIteration, maxNumberTreeNode;
	double maxConnectionDistance, goalBias;
	int star;          // 0 RRT, 1 RRT* as the reference (choose-parent only), 2 + rewire, 3 + rewire with a radius near-set
	double gamma;      // star = 3: near-set radius = gamma * sqrt(ln(n + 1) / (n + 1))
	int capacity;      // allocated nodes per tree
	int G;             // the spatial index has G x G cells over the bounds
	double invHx, invHy, hMin; // cells per metre in x / y, smaller cell side
};

struct RrtProblem {
	double initx, inity, goalx, goaly;
	unsigned long long seed;
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

__global__ void __launch_bounds__(RT) k_rrt(RrtArgs A, const RrtProblem* __restrict__ problems, double2* __restrict__ ptsBase, int32_t* __restrict__ parentBase,
	double* __restrict__ costBase, int32_t* __restrict__ cellHeadBase, int32_t* __restrict__ cellNextBase, RrtOut* __restrict__ outs, int32_t* __restrict__ childBase,
	double* __restrict__ edgeLenBase, int32_t* __restrict__ queueBase)
{
	// one workgroup = one tree
	const RrtProblem prob = problems[blockIdx.x];
	double2* const pts = ptsBase + (size_t)blockIdx.x * A.capacity;
	int32_t* const parent = parentBase + (size_t)blockIdx.x * A.capacity;
	double* const cost = costBase + (size_t)blockIdx.x * A.capacity;
	int32_t* const cellHead = cellHeadBase + (size_t)blockIdx.x * A.G * A.G; // -1 = empty (set by the host)
	int32_t* const cellNext = cellNextBase + (size_t)blockIdx.x * A.capacity;
	RrtOut* const out = outs + blockIdx.x;
	// rewire modes only: children lists (first child, next / previous sibling), length of the edge to the parent, two
	// queues for the subtree walk
	int32_t* const firstChild = childBase ? childBase + (size_t)blockIdx.x * A.capacity * 3 : nullptr;
	int32_t* const nextSib = firstChild ? firstChild + A.capacity : nullptr;
	int32_t* const prevSib = firstChild ? firstChild + 2 * (size_t)A.capacity : nullptr;
	double* const edgeLen = edgeLenBase ? edgeLenBase + (size_t)blockIdx.x * A.capacity : nullptr;
	int32_t* const queue0 = queueBase ? queueBase + (size_t)blockIdx.x * A.capacity * 2 : nullptr;
	int32_t* const queue1 = queue0 ? queue0 + A.capacity : nullptr;
	const bool rewire = A.star >= 2;
	__shared__ int s_qn;
	__shared__ double candD[kCandMax];
	__shared__ int candI[kCandMax];
	__shared__ int s_cnt;
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
		Mt64::seed(mt, prob.seed);
	int mtIdx = Mt64::N;
	if (tid == 0) {
		pts[0] = make_double2(prob.initx, prob.inity); // Tree::CreateRootNode, tree.h:59-65
		parent[0] = -1;
		cost[0] = 0.0;
		if (rewire) {
			firstChild[0] = nextSib[0] = prevSib[0] = -1;
			edgeLen[0] = 0.0;
		}
	}
	auto link_child = [&](int p, int c) { // thread 0
		nextSib[c] = firstChild[p];
		prevSib[c] = -1;
		if (firstChild[p] >= 0)
			prevSib[firstChild[p]] = c;
		firstChild[p] = c;
	};
	auto unlink_child = [&](int c) { // thread 0
		const int p = parent[c];
		if (prevSib[c] >= 0)
			nextSib[prevSib[c]] = nextSib[c];
		else
			firstChild[p] = nextSib[c];
		if (nextSib[c] >= 0)
			prevSib[nextSib[c]] = prevSib[c];
	};
	auto cell_x = [&](double x) { return min(A.G - 1, max(0, (int)((x - A.lbx) * A.invHx))); };
	auto cell_y = [&](double y) { return min(A.G - 1, max(0, (int)((y - A.lby) * A.invHy))); };
	auto index_insert = [&](int node, double x, double y) { // thread 0
		const int c = cell_x(x) * A.G + cell_y(y);
		cellNext[node] = cellHead[c];
		cellHead[c] = node;
	};
	if (tid == 0)
		index_insert(0, prob.initx, prob.inity);
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

	// Exact k nearest of (px, py) by a scan of the whole tree -> nearD / nearI (ascending (d, index)).
	auto brute_knn = [&](double px, double py, int k) {
		// local top-k per thread (ascending), then k rounds of block arg-min over the heads
		double ld[kMaxNear];
		int li[kMaxNear];
	#pragma unroll
		for (int s = 0; s < kMaxNear; s++) {
			ld[s] = __builtin_huge_val();
			li[s] = 0x7FFFFFFF;
		}
		for (int i = tid; i < n; i += RT) {
			const double2 p = pts[i];
			const double dx = p.x - px, dy = p.y - py;
			double cd = dx * dx + dy * dy;
			int ci = i;
			if (cand_before(cd, ci, ld[k - 1], li[k - 1])) {
				bool ins = false;
	#pragma unroll
				for (int s = 0; s < kMaxNear; s++) {
					if (s < k && (ins || cand_before(cd, ci, ld[s], li[s]))) {
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
		for (int r = 0; r < k; r++) {
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
	};

	// Exact k nearest of (px, py) through the cell lists -> nearD / nearI (ascending (d, index)); n > kBruteMax >= k.
	// A window of (2r+1)^2 cells around the query's cell holds every node closer than r cell sides, so the search ends
	// as soon as the k-th candidate is within that distance (or the window is the whole grid).
	auto grid_knn = [&](double px, double py, int k) {
		const int ci = cell_x(px), cj = cell_y(py);
		for (int r = 1;; r *= 2) {
			if (tid == 0)
				s_cnt = 0;
			__syncthreads();
			const int i0 = max(0, ci - r), i1 = min(A.G - 1, ci + r), j0 = max(0, cj - r), j1 = min(A.G - 1, cj + r);
			const int wj = j1 - j0 + 1, W = (i1 - i0 + 1) * wj;
			for (int c = tid; c < W; c += RT) {
				const int ii = i0 + c / wj, jj = j0 + c % wj;
				for (int node = cellHead[ii * A.G + jj]; node >= 0; node = cellNext[node]) {
					const double2 p = pts[node];
					const double dx = p.x - px, dy = p.y - py;
					const int slot = atomicAdd(&s_cnt, 1);
					if (slot < kCandMax) {
						candD[slot] = dx * dx + dy * dy;
						candI[slot] = node;
					}
				}
			}
			__syncthreads();
			const int M = s_cnt;
			const bool whole = i0 == 0 && j0 == 0 && i1 == A.G - 1 && j1 == A.G - 1;
			if (M > kCandMax) {
				// more candidates than the LDS buffer (dense cluster): scan the whole tree with the block-wide top-k
				__syncthreads();
				brute_knn(px, py, k);
				return;
			}
			// the k best by rank counting: candidate c is preceded by `rank` others in (d, index) order
			for (int c = tid; c < M; c += RT) {
				const double d = candD[c];
				const int idx = candI[c];
				int rank = 0;
				for (int j = 0; j < M; j++)
					rank += cand_before(candD[j], candI[j], d, idx) ? 1 : 0;
				if (rank < k) {
					nearD[rank] = d;
					nearI[rank] = idx;
				}
			}
			__syncthreads();
			const double reach = (double)r * A.hMin * (1.0 - 1e-9); // nodes outside the window are farther than this
			const bool done = whole || (M >= k && nearD[k - 1] <= reach * reach);
			__syncthreads();
			if (done)
				return;
		}
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
			rx = prob.goalx;
			ry = prob.goaly;
		} else {
			rx = draw(A.lbx, A.ubx); // StateSpaceR2::SampleUniform, state_space_r2.cpp:25-35
			ry = draw(A.lby, A.uby);
		}
		// ---- nearest node (Tree::GetNearestNode)
		int bi;
		if (n <= kBruteMax) {
			double bd = __builtin_huge_val();
			bi = 0x7FFFFFFF;
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
		} else {
			grid_knn(rx, ry, 1);
			bi = nearI[0];
			__syncthreads();
		}
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
			if (n <= kBruteMax) {
				for (int i = tid; i < n; i += RT) {
					const double2 p = pts[i];
					if (p.x == nx && p.y == ny && i < ei) {
						ed = 0.0;
						ei = i;
					}
				}
				block_argmin(ed, ei, s_d, s_i);
			} else {
				// an identical state can only sit in the new state's own cell
				if (tid == 0) {
					int found = 0x7FFFFFFF;
					for (int node = cellHead[cell_x(nx) * A.G + cell_y(ny)]; node >= 0; node = cellNext[node]) {
						const double2 p = pts[node];
						if (p.x == nx && p.y == ny && node < found)
							found = node;
					}
					s_best = found;
				}
				__syncthreads();
				ei = s_best;
				__syncthreads();
			}
			if (ei != 0x7FFFFFFF) {
				newNode = ei;
			} else {
				newNode = n;
				if (tid == 0) {
					pts[n] = make_double2(nx, ny);
					parent[n] = nearest;
					cost[n] = 0.0;
					index_insert(n, nx, ny);
				}
				n++;
				__syncthreads();
			}
			const double gx = nx - prob.goalx, gy = ny - prob.goaly;
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
		if (A.star == 3) { // radius search: the <= kMaxNear nearest within the shrinking ball (at least the nearest itself)
			k = n < kMaxNear ? n : kMaxNear;
		}
		nKnn++;
		if (n <= kBruteMax)
			brute_knn(nx, ny, k);
		else
			grid_knn(nx, ny, k);
		int kk = k;
		if (A.star == 3) {
			const double np1 = (double)n + 1.0;
			const double radius = A.gamma * sqrt(log(np1) / np1);
			int m = 0;
			while (m < k && nearD[m] <= radius * radius) // ascending distances: the ball's members are a prefix
				m++;
			kk = m < 1 ? 1 : m;
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
				index_insert(n, nx, ny);
			}
			n++;
			if (bestParent < 0)
				nKnn++; // GetNearestNode inside Extend
		}
		const bool existed = newNode != n - 1 || (p0.x == nx && p0.y == ny);
		if (tid == 0) {
			cost[newNode] = bestCost;
			if (rewire && !existed) {
				const int par = parent[newNode];
				const double2 pp = pts[par];
				const double ex = nx - pp.x, ey = ny - pp.y;
				edgeLen[newNode] = sqrt(ex * ex + ey * ey);
				firstChild[newNode] = -1;
				link_child(par, newNode);
			}
		}
		__syncthreads();
		if (rewire && !existed && bestParent >= 0) {
			// ---- rewire (beyond the reference): near nodes that get cheaper through the new node, in near-set order
			const int par = parent[newNode];
			if (tid < kk) {
				const int node = nearI[tid];
				const double2 p = pts[node];
				const double dx = p.x - nx, dy = p.y - ny;
				const double plen = sqrt(dx * dx + dy * dy);
				candCost[tid] = plen; // edge new -> near
				candValid[tid] = (node != newNode && node != par && edge_valid(A, nx, ny, p.x, p.y, plen)) ? 1 : 0;
			}
			__syncthreads();
			for (int r = 0; r < kk; r++) {
				const int node = nearI[r];
				if (node == newNode || node == par)
					continue; // (block-uniform)
				const double through = cost[newNode] + candCost[r];
				const bool cheaper = through < cost[node];
				if (cheaper)
					nEdge++;
				if (!(cheaper && candValid[r]))
					continue;
				if (tid == 0) {
					unlink_child(node);
					parent[node] = newNode;
					link_child(newNode, node);
					edgeLen[node] = candCost[r];
					cost[node] = through;
					queue0[0] = node;
					s_qn = 0;
				}
				__syncthreads();
				// carry the new cost down the subtree, one level per pass: cost[c] = cost[parent] + edgeLen[c]
				int cur = 1;
				int32_t *qa = queue0, *qb = queue1;
				while (cur > 0) {
					for (int i = tid; i < cur; i += RT) {
						const int p = qa[i];
						const double cp = cost[p];
						for (int c = firstChild[p]; c >= 0; c = nextSib[c]) {
							cost[c] = cp + edgeLen[c];
							qb[atomicAdd(&s_qn, 1)] = c;
						}
					}
					__syncthreads();
					cur = s_qn;
					__syncthreads();
					if (tid == 0)
						s_qn = 0;
					int32_t* t = qa;
					qa = qb;
					qb = t;
					__syncthreads();
				}
			}
		}
		if (nx == prob.goalx && ny == prob.goaly) { // RRTStar::IsSolution: exact equality, rrt_star.h:136-139
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

int pp_rrt_run_batch(pp_ctx* ctx, pp_map* map, const double lower[2], const double upper[2], const double params[4], int32_t n_problems,
	const double* inits_xy, const double* goals_xy, const uint64_t* seeds, int32_t star, pp_rrt** outs, pp_rrt_result* results)
{
	using pph::set_error;
	if (!ctx || !lower || !upper || !params || n_problems < 0 || (n_problems > 0 && (!inits_xy || !goals_xy || !seeds || !outs || !results))) {
		set_error("null argument");
		return PP_ERR_INVALID;
	}
	if (n_problems == 0)
		return PP_OK;
	if (map && !map->dist) {
		set_error("distance grid not uploaded (pp_map_upload_dist2)");
		return PP_ERR_INVALID;
	}
	if (star < 0 || star > 3 || (star == 3 && !(params[4] > 0.0))) {
		set_error("star: 0 RRT, 1 RRT* (reference), 2 RRT* + rewire, 3 RRT* + rewire + radius near-set (params[4] = gamma > 0)");
		return PP_ERR_INVALID;
	}
	if (!(upper[0] > lower[0]) || !(upper[1] > lower[1])) {
		set_error("empty bounds");
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
	A.star = star;
	A.gamma = star == 3 ? params[4] : 0.0;
	// the loop stops once size > maxNumberTreeNode or after maxIteration + 1 iterations
	const unsigned long long cap = std::min<unsigned long long>((unsigned long long)A.maxNumberTreeNode + 2ull, (unsigned long long)A.maxIteration + 3ull);
	if (cap > (1ull << 28) || cap * (unsigned long long)n_problems > (1ull << 32)) {
		set_error("tree(s) too large");
		return PP_ERR_INVALID;
	}
	A.capacity = (int)cap;
	// spatial index: about four nodes per cell when the tree is full
	int G = (int)std::ceil(std::sqrt((double)cap / 4.0));
	A.G = G < 1 ? 1 : (G > 2048 ? 2048 : G);
	A.invHx = A.G / (upper[0] - lower[0]);
	A.invHy = A.G / (upper[1] - lower[1]);
	A.hMin = std::min((upper[0] - lower[0]) / A.G, (upper[1] - lower[1]) / A.G);
	const size_t np = (size_t)n_problems;
	std::vector<RrtProblem> probs(np);
	for (size_t i = 0; i < np; i++) {
		probs[i].initx = inits_xy[2 * i];
		probs[i].inity = inits_xy[2 * i + 1];
		probs[i].goalx = goals_xy[2 * i];
		probs[i].goaly = goals_xy[2 * i + 1];
		probs[i].seed = seeds[i];
	}
	RrtProblem* dprob = nullptr;
	double2* pts = nullptr;
	int32_t *parent = nullptr, *cellHead = nullptr, *cellNext = nullptr;
	double* cost = nullptr;
	RrtOut* dout = nullptr;
	int32_t *child = nullptr, *queue = nullptr;
	double* edgeLen = nullptr;
	const size_t cells = (size_t)A.G * A.G;
	hipError_t e = hipMalloc((void**)&pts, np * cap * sizeof(double2));
	if (e == hipSuccess)
		e = hipMalloc((void**)&parent, np * cap * 4);
	if (e == hipSuccess)
		e = hipMalloc((void**)&cost, np * cap * 8);
	if (e == hipSuccess)
		e = hipMalloc((void**)&cellNext, np * cap * 4);
	if (e == hipSuccess)
		e = hipMalloc((void**)&cellHead, np * cells * 4);
	if (e == hipSuccess)
		e = hipMalloc((void**)&dout, np * sizeof(RrtOut));
	if (e == hipSuccess)
		e = hipMalloc((void**)&dprob, np * sizeof(RrtProblem));
	if (star >= 2) { // children lists, edge lengths and the subtree-walk queues of the rewire modes
		if (e == hipSuccess)
			e = hipMalloc((void**)&child, np * cap * 3 * 4);
		if (e == hipSuccess)
			e = hipMalloc((void**)&edgeLen, np * cap * 8);
		if (e == hipSuccess)
			e = hipMalloc((void**)&queue, np * cap * 2 * 4);
	}
	std::vector<RrtOut> ho(np);
	std::vector<std::unique_ptr<pp_rrt>> rs(np);
	if (e == hipSuccess)
		e = hipMemsetAsync(cellHead, 0xFF, np * cells * 4, ctx->stream); // -1 = empty cell
	if (e == hipSuccess)
		e = hipMemcpyAsync(dprob, probs.data(), np * sizeof(RrtProblem), hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_rrt, dim3(n_problems), dim3(RT), 0, ctx->stream, A, dprob, pts, parent, cost, cellHead, cellNext, dout, child, edgeLen, queue);
		e = hipGetLastError();
	}
	if (e == hipSuccess)
		e = hipMemcpyAsync(ho.data(), dout, np * sizeof(RrtOut), hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess)
		e = hipStreamSynchronize(ctx->stream);
	for (size_t i = 0; i < np && e == hipSuccess; i++) {
		rs[i] = std::make_unique<pp_rrt>();
		pp_rrt* r = rs[i].get();
		const int n = ho[i].nNodes;
		r->nodes.resize((size_t)n * 2);
		r->parents.resize(n);
		r->costs.resize(n);
		e = hipMemcpy(r->nodes.data(), pts + i * cap, (size_t)n * 16, hipMemcpyDeviceToHost);
		if (e == hipSuccess)
			e = hipMemcpy(r->parents.data(), parent + i * cap, (size_t)n * 4, hipMemcpyDeviceToHost);
		if (e == hipSuccess)
			e = hipMemcpy(r->costs.data(), cost + i * cap, (size_t)n * 8, hipMemcpyDeviceToHost);
	}
	(void)hipFree(pts);
	(void)hipFree(parent);
	(void)hipFree(cost);
	(void)hipFree(cellNext);
	(void)hipFree(cellHead);
	(void)hipFree(dout);
	(void)hipFree(dprob);
	if (child)
		(void)hipFree(child);
	if (edgeLen)
		(void)hipFree(edgeLen);
	if (queue)
		(void)hipFree(queue);
	if (e != hipSuccess)
		return pph::hip_fail(e, "pp_rrt_run_batch");
	for (size_t i = 0; i < np; i++) {
		pp_rrt* r = rs[i].get();
		// GetPath, rrt.h:97-115: states from the root to the solution node
		if (ho[i].solution >= 0) {
			std::vector<int> chain;
			for (int k = ho[i].solution; k >= 0; k = r->parents[k])
				chain.push_back(k);
			for (size_t j = chain.size(); j-- > 0;) {
				r->path.push_back(r->nodes[2 * chain[j]]);
				r->path.push_back(r->nodes[2 * chain[j] + 1]);
			}
		}
		results[i].status = ho[i].status;
		results[i].n_nodes = ho[i].nNodes;
		results[i].n_path = (int32_t)(r->path.size() / 2);
		results[i].iterations = ho[i].iterations;
		results[i].n_knn_queries = ho[i].nKnn;
		results[i].n_edge_checks = ho[i].nEdge;
	}
	for (size_t i = 0; i < np; i++)
		outs[i] = rs[i].release();
	return PP_OK;
}

int pp_rrt_run(pp_ctx* ctx, pp_map* map, const double lower[2], const double upper[2], const double params[4], const double init[2], const double goal[2],
	uint64_t seed, int32_t star, pp_rrt** out, pp_rrt_result* result)
{
	if (!init || !goal || !out || !result) {
		pph::set_error("null argument");
		return PP_ERR_INVALID;
	}
	return pp_rrt_run_batch(ctx, map, lower, upper, params, 1, init, goal, &seed, star, out, result);
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
