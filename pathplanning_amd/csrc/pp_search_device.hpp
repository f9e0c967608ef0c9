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
	Resolutions lat; // spatialRes / angularRes with their reciprocals (div_by)
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
	int i = trunc_to_int(round(div_by(delta.x + h.offX, h.lat.spatial, h.lat.invSpatial)));
	int j = trunc_to_int(round(div_by(delta.y + h.offY, h.lat.spatial, h.lat.invSpatial)));
	int k = trunc_to_int(round(div_by(delta.t, h.lat.angular, h.lat.invAngular)));
	if (k == h.na)
		k = 0;
	if (i < 0 || i >= h.nx || j < 0 || j >= h.ny) {
		double euclideanDistance = sqrt(delta.x * delta.x + delta.y * delta.y);
		return h.minMult * euclideanDistance;
	}
	return nonholo_lookup(h, i, j, k);
}

/// Same as nonholo_heuristic with sin/cos of state.t supplied by the caller (they were computed when
/// the pose was produced): between(goal, state) rotates by -state.t, whose sine is -sinS and cosine cosS.
PPD_INLINE double nonholo_heuristic_sc(const HeurView& h, const Pose& goal, const Pose& state, double sinS, double cosS)
{
	const double dx = goal.x - state.x, dy = goal.y - state.y;
	const double s = -sinS, c = cosS;
	Pose delta;
	delta.x = c * dx + (-s) * dy;
	delta.y = s * dx + c * dy;
	delta.t = wrap_theta(wrap_theta(goal.t - state.t));
	int i = trunc_to_int(round(div_by(delta.x + h.offX, h.lat.spatial, h.lat.invSpatial)));
	int j = trunc_to_int(round(div_by(delta.y + h.offY, h.lat.spatial, h.lat.invSpatial)));
	int k = trunc_to_int(round(div_by(delta.t, h.lat.angular, h.lat.invAngular)));
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
	const float c = field[field_tiled_index(m.cols, row, col)]; // tiled layout, see pp_device.hpp
	if (c == __builtin_huge_valf())
		return euclidean;
	const double heuristic = (double)(c * h.obstCostMult - h.obstDiagRes);
	return heuristic < euclidean ? euclidean : heuristic; // std::max(heuristic, euclidean)
}

/// combined_heuristic_sc in two halves: `issue` computes the indices and starts the two table reads (non-holonomic
/// table, obstacle field), `finish` does the arithmetic on the loaded values -- the caller runs the validity march in
/// between, so the reads overlap it instead of stalling the wave one after the other.  Same value as
/// combined_heuristic_sc.
struct HeurLoads {
	double aRaw;    // non-holonomic table entry (when aFromTable)
	double aConst;  // value that needs no read: outside the table, or one of the k < 0 cases
	float cRaw;     // obstacle field entry (when cInside)
	double euclid;
	bool aFromTable, cInside;
};
PPD_INLINE void combined_heuristic_issue(const HeurView& h, const MapView& m, const float* field, const Pose& goal, const Pose& state, double sinS, double cosS, HeurLoads& L)
{
	const double dx = goal.x - state.x, dy = goal.y - state.y;
	L.aFromTable = false;
	L.aConst = 0.0;
	L.aRaw = 0.0;
	{ // nonholo_heuristic_sc
		const double s = -sinS, c = cosS;
		Pose delta;
		delta.x = c * dx + (-s) * dy;
		delta.y = s * dx + c * dy;
		delta.t = wrap_theta(wrap_theta(goal.t - state.t));
		int i = trunc_to_int(round(div_by(delta.x + h.offX, h.lat.spatial, h.lat.invSpatial)));
		int j = trunc_to_int(round(div_by(delta.y + h.offY, h.lat.spatial, h.lat.invSpatial)));
		int k = trunc_to_int(round(div_by(delta.t, h.lat.angular, h.lat.invAngular)));
		if (k == h.na)
			k = 0;
		if (i < 0 || i >= h.nx || j < 0 || j >= h.ny) {
			double euclideanDistance = sqrt(delta.x * delta.x + delta.y * delta.y);
			L.aConst = h.minMult * euclideanDistance;
		} else {
			// nonholo_lookup
			const size_t na = (size_t)h.na;
			const double* p = nullptr;
			if (k >= 0)
				p = &h.table[((size_t)i * h.ny + j) * na + k];
			else if (!h.negativeKRead)
				p = &h.table[((size_t)i * h.ny + j) * na + (k + h.na)];
			else {
				const unsigned long long chunk = ((unsigned long long)(na * 8 + 8 + 15) / 16) * 16;
				if (k == -1)
					L.aConst = __longlong_as_double((long long)(chunk | 1ull));
				else if (j >= 1) {
					const int kk = (int)(chunk / 8) + k;
					if (kk >= 0 && kk < h.na)
						p = &h.table[((size_t)i * h.ny + (j - 1)) * na + kk];
				}
			}
			if (p) {
				L.aRaw = *p;
				L.aFromTable = true;
			}
		}
	}
	{ // obstacle_heuristic
		L.euclid = sqrt(dx * dx + dy * dy);
		int row, col;
		world_to_cell(m, state.x, state.y, row, col);
		L.cInside = inside_map(m, row, col);
		L.cRaw = 0.0f;
		if (L.cInside)
			L.cRaw = field[field_tiled_index(m.cols, row, col)];
	}
}
PPD_INLINE double combined_heuristic_finish(const HeurView& h, const HeurLoads& L)
{
	double value = -__builtin_huge_val();
	const double a = L.aFromTable ? L.aRaw : L.aConst;
	value = value < a ? a : value;
	double b = L.euclid;
	if (L.cInside && !(L.cRaw == __builtin_huge_valf())) {
		const double heuristic = (double)(L.cRaw * h.obstCostMult - h.obstDiagRes);
		b = heuristic < L.euclid ? L.euclid : heuristic;
	}
	value = value < b ? b : value;
	return value;
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

PPD_INLINE double combined_heuristic_sc(const HeurView& h, const MapView& m, const float* field, const Pose& goal, const Pose& state, double sinS, double cosS)
{
	double value = -__builtin_huge_val();
	const double a = nonholo_heuristic_sc(h, goal, state, sinS, cosS);
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

// ------------------------------------------------------ cross-lane helpers --
// A __shfl is a ds_bpermute: an LDS round trip (~100+ cycles) that a single latency-bound wave
// cannot hide.  These use the gfx9 data-parallel primitives (DPP) and v_readlane instead.
PPD_INLINE unsigned int lane_read(unsigned int v, int l) { return (unsigned int)__builtin_amdgcn_readlane((int)v, l); } // l wave-uniform
PPD_INLINE unsigned long long lane_read64(unsigned long long v, int l)
{
	return ((unsigned long long)lane_read((unsigned int)(v >> 32), l) << 32) | lane_read((unsigned int)v, l);
}
/// lane i receives lane i-1's value; lane 0 receives `fill`  (DPP wave_shr:1)
PPD_INLINE unsigned int wave_shr1(unsigned int v, unsigned int fill) { return (unsigned int)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xF, 0xF, false); }
/// lane i receives lane i+1's value; lane 63 receives `fill`  (DPP wave_shl:1)
PPD_INLINE unsigned int wave_shl1(unsigned int v, unsigned int fill) { return (unsigned int)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xF, 0xF, false); }
PPD_INLINE unsigned long long wave_shr1_64(unsigned long long v, unsigned long long fill)
{
	return ((unsigned long long)wave_shr1((unsigned int)(v >> 32), (unsigned int)(fill >> 32)) << 32) | wave_shr1((unsigned int)v, (unsigned int)fill);
}
PPD_INLINE unsigned long long wave_shl1_64(unsigned long long v, unsigned long long fill)
{
	return ((unsigned long long)wave_shl1((unsigned int)(v >> 32), (unsigned int)(fill >> 32)) << 32) | wave_shl1((unsigned int)v, (unsigned int)fill);
}
/// minimum over the 64 lanes, wave-uniform result: row_shr 1,2,4,8 then row_bcast 15 / 31 (lane 63 ends up with the total)
PPD_INLINE unsigned int wave_min_u32(unsigned int v)
{
	const int id = (int)0xFFFFFFFF;
	unsigned int x = v;
	x = min(x, (unsigned int)__builtin_amdgcn_update_dpp(id, (int)x, 0x111, 0xF, 0xF, false));
	x = min(x, (unsigned int)__builtin_amdgcn_update_dpp(id, (int)x, 0x112, 0xF, 0xF, false));
	x = min(x, (unsigned int)__builtin_amdgcn_update_dpp(id, (int)x, 0x114, 0xF, 0xF, false));
	x = min(x, (unsigned int)__builtin_amdgcn_update_dpp(id, (int)x, 0x118, 0xF, 0xF, false));
	x = min(x, (unsigned int)__builtin_amdgcn_update_dpp(id, (int)x, 0x142, 0xA, 0xF, false));
	x = min(x, (unsigned int)__builtin_amdgcn_update_dpp(id, (int)x, 0x143, 0xC, 0xF, false));
	return lane_read(x, 63);
}
/// lane of the lexicographically smallest (k, s) over the wave (lowest lane among equals); also returns that key
PPD_INLINE int wave_argmin_key(unsigned long long k, unsigned int s, unsigned long long& mk, unsigned int& ms)
{
	const unsigned int hi = (unsigned int)(k >> 32), lo = (unsigned int)k;
	const unsigned int mhi = wave_min_u32(hi);
	bool cand = hi == mhi;
	const unsigned int mlo = wave_min_u32(cand ? lo : 0xFFFFFFFFu);
	cand = cand && lo == mlo;
	ms = wave_min_u32(cand ? s : 0xFFFFFFFFu);
	cand = cand && s == ms;
	mk = ((unsigned long long)mhi << 32) | mlo;
	return __ffsll((long long)__ballot(cand)) - 1;
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

/// whole wave (64 lanes); `size` and `cachedTop` (a register copy of heap[0]) are wave-uniform.
/// Returns the popped entry and leaves the new root in `cachedTop`: no load of heap[0] before or after,
/// so a pop costs one memory round trip per level below the root.
PPD_INLINE HeapEntry heap_pop_wave(HeapEntry* heap, int& size, int lane, HeapEntry& cachedTop)
{
	const HeapEntry top = cachedTop;
	const int hs = size - 1;
	size = hs;
	if (hs <= 0) {
		cachedTop.ckey = ~0ull;
		cachedTop.nseq = ~0u;
		cachedTop.node = 0;
		return top;
	}
	const HeapEntry last = heap[hs];
	HeapEntry newRoot = last;
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
		unsigned long long mk;
		unsigned int ms;
		const int minLane = wave_argmin_key(e.ckey, e.nseq, mk, ms);
		HeapEntry best;
		best.ckey = mk;
		best.nseq = ms;
		if (!heap_before(best, last))
			break;
		if (lane == minLane)
			heap[i] = e;
		if (i == 0) {
			best.node = lane_read(e.node, minLane);
			newRoot = best;
		}
		i = first + minLane;
	}
	if (lane == 0)
		heap[i] = last;
	cachedTop = newRoot;
	return top;
}

// ----------------------------------------------------------- front buffer --
// The best entries of the open list live in registers, one per lane, sorted: lane l holds the
// (l+1)-th best; lanes >= count hold the MAX sentinel.  Insert and pop are a ballot + one
// lane shift -- no memory traffic.  Entries that fall off the end spill to the 64-ary heap in
// HBM; a pop takes whichever of (front[0], heap[0]) comes first in the reference's pop order, so
// the pair (front, heap) pops exactly like one Frontier.
struct FrontLane {
	unsigned long long ckey;
	unsigned int nseq;
	unsigned int node;
};
PPD_INLINE void front_clear(FrontLane& f)
{
	f.ckey = ~0ull;
	f.nseq = ~0u;
	f.node = 0;
}
PPD_INLINE bool key_before(unsigned long long ak, unsigned int as, unsigned long long bk, unsigned int bs) { return ak < bk || (ak == bk && as < bs); }

#ifndef PP_SEARCH_SETPRIO
#define PP_SEARCH_SETPRIO 1 // search waves raise their issue priority (s_setprio 3) over the wavefront kernels' waves
#endif
#ifndef PP_ROWS_PRIO
#define PP_ROWS_PRIO 1 // rows kernel: above the wavefront kernels, below the one-query-per-wave kernel (which runs the longest queries)
#endif
#ifndef PP_FRONT_CAP
#define PP_FRONT_CAP 64 // entries of the open list kept in registers (tuning experiments: 16 / 32)
#endif
/// Inserts e (wave-uniform).  Returns true when an entry left the buffer (written to `spilled`).
PPD_INLINE bool front_insert(FrontLane& f, int& count, const HeapEntry& e, int lane, HeapEntry& spilled)
{
	constexpr int kCap = PP_FRONT_CAP;
	const bool mineFirst = lane < count && key_before(f.ckey, f.nseq, e.ckey, e.nseq);
	const int pos = __popcll(__ballot(mineFirst));
	if (pos >= kCap) {
		spilled = e;
		return true;
	}
	bool spill = false;
	if (count == kCap) {
		spilled.ckey = lane_read64(f.ckey, kCap - 1);
		spilled.nseq = lane_read(f.nseq, kCap - 1);
		spilled.node = lane_read(f.node, kCap - 1);
		spill = true;
	}
	const unsigned long long uk = wave_shr1_64(f.ckey, ~0ull);
	const unsigned int us = wave_shr1(f.nseq, ~0u);
	const unsigned int un = wave_shr1(f.node, 0u);
	if (lane > pos && lane < kCap) {
		f.ckey = uk;
		f.nseq = us;
		f.node = un;
	} else if (lane == pos) {
		f.ckey = e.ckey;
		f.nseq = e.nseq;
		f.node = e.node;
	}
	if (count < kCap)
		count++;
	return spill;
}

/// Removes and returns front[0] (count > 0, wave-uniform).
PPD_INLINE HeapEntry front_pop(FrontLane& f, int& count, int lane)
{
	HeapEntry top;
	top.ckey = lane_read64(f.ckey, 0);
	top.nseq = lane_read(f.nseq, 0);
	top.node = lane_read(f.node, 0);
	// every lane takes its right neighbour's entry; lanes >= count hold the sentinel, lane 63 is refilled with it
	f.ckey = wave_shl1_64(f.ckey, ~0ull);
	f.nseq = wave_shl1(f.nseq, ~0u);
	f.node = wave_shl1(f.node, 0u);
	count--;
	return top;
}

// ------------------------------------------------------- open list: f-bands --
// Outside the register front buffer the open list is kept in BANDS of the total cost f: band = floor(f / W).  A ring of
// kBands slots x kBandCap entries per query in HBM holds the bands being filled (slot = band mod kBands, a slot serves
// one band at a time); what does not fit goes to the 64-ary heap.  The front buffer always holds the globally best
// entries (an entry enters a non-full front only if it beats a lower bound of everything outside), so a pop never needs
// the heap, and an empty front is refilled with the whole lowest band: ONE coalesced load, sorted in the wave / row.
// Measured motivation (tools/study_open_list.py): with the plain "most recent 64" buffer 70 % of the pops came from the
// heap (two to three dependent HBM round trips each).
constexpr int kBandShift = 10, kBands = 1 << kBandShift, kBandCap = 16;
// 1024 slots x 16 entries (256 KiB per query), W = 1/64: a window of 16 cost units that starts at `bandLo`, the lowest band
// that may hold entries (slot = band & 1023 is unique inside the window; bands outside it go to the heap).  The CPU model of
// the policy on oracle traces (tools/study_open_list.py) sends 0.65 % of the entries to the heap and refills a 16-entry
// buffer once per ~6 pops.  A band is one entry per lane of a 16-lane row (k_hybrid_search_rows); the 64-lane kernel
// loads the aligned group of four consecutive bands.  Slot fill counts (u8) live in LDS, 1 KiB per query.
/// inverse of cost_key
PPD_INLINE double key_cost(unsigned long long k)
{
	const unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
	return __longlong_as_double((long long)b);
}
PPD_INLINE long long band_of_key(unsigned long long ckey, double invW)
{
	double b = floor(key_cost(ckey) * invW);
	b = b < -1.0e15 ? -1.0e15 : (b > 1.0e15 ? 1.0e15 : b); // (NaN -> comparisons false -> unchanged -> cast below is bounded by the heap path)
	return (long long)b;
}
/// sorts the 64 lanes' entries ascending in (ckey, nseq) -- bitonic network over __shfl_xor; keys are unique
PPD_INLINE void wave_sort_entries(unsigned long long& ckey, unsigned int& nseq, unsigned int& node, int lane)
{
#pragma unroll
	for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
		for (int j = k >> 1; j > 0; j >>= 1) {
			const unsigned long long ok = ((unsigned long long)(unsigned int)__shfl_xor((int)(ckey >> 32), j, 64) << 32) | (unsigned int)__shfl_xor((int)ckey, j, 64);
			const unsigned int os = (unsigned int)__shfl_xor((int)nseq, j, 64), on = (unsigned int)__shfl_xor((int)node, j, 64);
			const bool up = (lane & k) == 0, lower = (lane & j) == 0;
			const bool otherBefore = ok < ckey || (ok == ckey && os < nseq);
			if ((lower == up) == otherBefore) {
				ckey = ok;
				nseq = os;
				node = on;
			}
		}
	}
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
