// Device-side pieces of the batched Hybrid-A* graph search: heuristics, open list, RNG.
#pragma once

#include "pp_internal.hpp"
#include "pp_rs_device.hpp"

namespace ppd {

// ------------------------------------------------------------- heuristics --
struct HeurView {
	// NonHolonomicHeuristic (algo/heuristics.cpp:7-95)
	int nx, ny, na;
	double spatialRes, angularRes, offX, offY, minMult;
	const double* table; // [(i*ny + j)*na + k]
	int negativeKRead;
	// ObstaclesHeuristic (algo/heuristics.cpp:97-166)
	float obstCostMult; // min(rev, fwd) * resolution, as float (heuristics.cpp:99)
	float obstDiagRes;  // sqrt(2) * resolution, as float (heuristics.cpp:98)
};

/// m_values[i][j][k] with the out-of-bounds rows the reference reads for k < 0 (SURVEY
/// Appendix A Q7): sequential glibc chunks of 8*na+8 bytes rounded to 16 (74 doubles for
/// na = 73).  k == -1 -> the chunk-size word; k <= -2 -> row j-1 at stride+k; j == 0 -> the
/// row-pointer array (denormal-sized heap addresses), modelled as +0.0.
PPD_INLINE double nonholo_lookup(const HeurView& h, int i, int j, int k)
{
	const size_t na = (size_t)h.na;
	if (k >= 0)
		return h.table[((size_t)i * h.ny + j) * na + k];
	if (!h.negativeKRead)
		return h.table[((size_t)i * h.ny + j) * na + (k + h.na)];
	const unsigned long long chunk = ((unsigned long long)(na * 8 + 8 + 15) / 16) * 16;
	if (k == -1)
		return __longlong_as_double((long long)(chunk | 1ull));
	if (j >= 1) {
		const int kk = (int)(chunk / 8) + k;
		if (kk >= 0 && kk < h.na)
			return h.table[((size_t)i * h.ny + (j - 1)) * na + kk];
		return 0.0;
	}
	return 0.0;
}

/// NonHolonomicHeuristic::GetHeuristicValue, algo/heuristics.cpp:78-95
PPD_INLINE double nonholo_heuristic(const HeurView& h, const Pose& goal, const Pose& state)
{
	Pose delta = rs::between(goal, state);
	int i = trunc_to_int(round((delta.x + h.offX) / h.spatialRes));
	int j = trunc_to_int(round((delta.y + h.offY) / h.spatialRes));
	int k = trunc_to_int(round(delta.t / h.angularRes));
	if (k == h.na)
		k = 0;
	if (i < 0 || i >= h.nx || j < 0 || j >= h.ny) {
		double euclideanDistance = sqrt(delta.x * delta.x + delta.y * delta.y);
		return h.minMult * euclideanDistance;
	}
	return nonholo_lookup(h, i, j, k);
}

/// ObstaclesHeuristic::GetHeuristicValue, algo/heuristics.cpp:155-165.  `field` is the
/// wavefront result of this query's goal; +inf marks cells the reference leaves unexplored.
PPD_INLINE double obstacle_heuristic(const HeurView& h, const MapView& m, const float* field, const Pose& goal, const Pose& state)
{
	const double dx = goal.x - state.x, dy = goal.y - state.y;
	const double euclidean = sqrt(dx * dx + dy * dy);
	int row, col;
	world_to_cell(m, state.x, state.y, row, col);
	if (!inside_map(m, row, col))
		return euclidean;
	const float c = field[(size_t)row * m.cols + col];
	if (c == __builtin_huge_valf())
		return euclidean;
	const double heuristic = (double)(c * h.obstCostMult - h.obstDiagRes);
	return heuristic < euclidean ? euclidean : heuristic; // std::max(heuristic, euclidean)
}

/// AStarCombinedHeuristic::GetHeuristicValue, algo/a_star.h:102-109
PPD_INLINE double combined_heuristic(const HeurView& h, const MapView& m, const float* field, const Pose& goal, const Pose& state)
{
	double value = -__builtin_huge_val();
	const double a = nonholo_heuristic(h, goal, state);
	value = value < a ? a : value;
	const double b = obstacle_heuristic(h, m, field, goal, state);
	value = value < b ? b : value;
	return value;
}

/// HybridAStar::GraphSearch::IdenticalPoses, algo/hybrid_a_star.h:208-211
PPD_INLINE bool identical_poses(const Pose& a, const Pose& b)
{
	const double tol = 1e-3;
	const double dx = a.x - b.x, dy = a.y - b.y;
	return sqrt(dx * dx + dy * dy) < tol && fabs(a.t - b.t) < tol * kPi / 180.0;
}

// -------------------------------------------------------------- open list --
// 64-ary min-heap ordered like the reference Frontier pops (utils/frontier.h:39-48,83-91):
// smallest totalCost first, and among equal costs the most recently pushed (Appendix A Q1).
struct HeapEntry {
	unsigned long long ckey; // order-preserving image of the double totalCost
	unsigned int nseq;       // 0xFFFFFFFF - pushSequence: smaller = pushed later
	unsigned int node;
};

PPD_INLINE unsigned long long cost_key(double c)
{
	unsigned long long b = (unsigned long long)__double_as_longlong(c);
	return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
PPD_INLINE bool heap_before(const HeapEntry& a, const HeapEntry& b) { return a.ckey < b.ckey || (a.ckey == b.ckey && a.nseq < b.nseq); }

/// lane 0 only
PPD_INLINE void heap_push(HeapEntry* heap, int& size, const HeapEntry& e)
{
	int i = size++;
	while (i > 0) {
		const int p = (i - 1) >> 6;
		const HeapEntry pe = heap[p];
		if (!heap_before(e, pe))
			break;
		heap[i] = pe;
		i = p;
	}
	heap[i] = e;
}

/// whole wave (64 lanes); `size` is wave-uniform.  Returns the popped entry.
PPD_INLINE HeapEntry heap_pop_wave(HeapEntry* heap, int& size, int lane)
{
	const HeapEntry top = heap[0];
	const int hs = size - 1;
	size = hs;
	if (hs > 0) {
		const HeapEntry last = heap[hs];
		int i = 0;
		for (;;) {
			const int first = (i << 6) + 1;
			if (first >= hs)
				break;
			const int c = first + lane;
			HeapEntry e;
			if (c < hs)
				e = heap[c];
			else {
				e.ckey = ~0ull;
				e.nseq = ~0u;
				e.node = 0;
			}
			// wave arg-min of (ckey, nseq)
			unsigned long long mk = e.ckey;
			unsigned int ms = e.nseq;
#pragma unroll
			for (int off = 32; off > 0; off >>= 1) {
				const unsigned int lo = __shfl_xor((int)(unsigned int)mk, off, 64);
				const unsigned int hi = __shfl_xor((int)(unsigned int)(mk >> 32), off, 64);
				const unsigned long long ok = ((unsigned long long)hi << 32) | lo;
				const unsigned int os = __shfl_xor((int)ms, off, 64);
				if (ok < mk || (ok == mk && os < ms)) {
					mk = ok;
					ms = os;
				}
			}
			HeapEntry best;
			best.ckey = mk;
			best.nseq = ms;
			if (!heap_before(best, last))
				break;
			const unsigned long long match = __ballot(e.ckey == mk && e.nseq == ms);
			const int minLane = __ffsll((long long)match) - 1;
			if (lane == minLane)
				heap[i] = e;
			i = first + minLane;
		}
		if (lane == 0)
			heap[i] = last;
	}
	return top;
}

// ------------------------------------------------------------------- RNG --
// std::mt19937_64 + std::uniform_real_distribution<double>(0, nextafter(1, max)) exactly as
// Random<double>::SampleUniform draws (utils/random.h:12-27) with libstdc++'s
// generate_canonical<double, 53> (one 64-bit draw: double(u64) / 2^64, clamped below 1).
struct Mt64 {
	static constexpr int N = 312, M = 156;
	PPD_INLINE static void seed(unsigned long long* mt, unsigned long long s)
	{
		mt[0] = s;
		for (int i = 1; i < N; i++)
			mt[i] = 6364136223846793005ull * (mt[i - 1] ^ (mt[i - 1] >> 62)) + (unsigned long long)i;
	}
	/// whole wave; regenerates all 312 words in place
	PPD_INLINE static void twist_wave(unsigned long long* mt, int lane)
	{
		const unsigned long long UM = 0xFFFFFFFF80000000ull, LM = 0x7FFFFFFFull, A = 0xB5026F5AA96619E9ull;
		// i in [0, 156): inputs are all old values
		for (int base = 0; base < M; base += 64) {
			const int i = base + lane;
			unsigned long long v = 0;
			if (i < M) {
				const unsigned long long x = (mt[i] & UM) | (mt[i + 1] & LM);
				v = mt[i + M] ^ (x >> 1) ^ ((x & 1ull) ? A : 0ull);
			}
			__syncthreads();
			if (i < M)
				mt[i] = v;
			__syncthreads();
		}
		// i in [156, 311): mt[i - 156] is already new, mt[i], mt[i+1] old
		for (int base = M; base < N - 1; base += 64) {
			const int i = base + lane;
			unsigned long long v = 0;
			if (i < N - 1) {
				const unsigned long long x = (mt[i] & UM) | (mt[i + 1] & LM);
				v = mt[i - M] ^ (x >> 1) ^ ((x & 1ull) ? A : 0ull);
			}
			__syncthreads();
			if (i < N - 1)
				mt[i] = v;
			__syncthreads();
		}
		if (lane == 0) {
			const unsigned long long x = (mt[N - 1] & UM) | (mt[0] & LM);
			mt[N - 1] = mt[M - 1] ^ (x >> 1) ^ ((x & 1ull) ? A : 0ull);
		}
		__syncthreads();
	}
	PPD_INLINE static unsigned long long temper(unsigned long long y)
	{
		y ^= (y >> 29) & 0x5555555555555555ull;
		y ^= (y << 17) & 0x71D67FFFEDA60000ull;
		y ^= (y << 37) & 0xFFF7EEE000000000ull;
		y ^= (y >> 43);
		return y;
	}
	/// Random<double>::SampleUniform(0.0, 1.0) from one raw 64-bit output
	PPD_INLINE static double uniform01(unsigned long long raw)
	{
		double r = (double)raw / 18446744073709551616.0; // generate_canonical
		if (r >= 1.0)
			r = 0.99999999999999988897769753748434595763683319091796875; // nextafter(1, 0)
		const double b = 1.0000000000000002220446049250313080847263336181640625; // nextafter(1, max)
		const double u = (b - 0.0) * r + 0.0; // uniform_real_distribution::operator()
		return 0.0 + (1.0 - 0.0) * u;         // lb + range * u
	}
};

} // namespace ppd
