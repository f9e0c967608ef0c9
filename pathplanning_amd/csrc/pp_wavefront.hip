// obst_wavefront -- ObstaclesHeuristic::Update (algo/heuristics.cpp:106-153) on gfx950,
// EXACT reference order: first discovery wins, no relaxation, LIFO among equal costs
// (utils/frontier.h:39-48,83-91).
//
// The reference pops a sequential open list.  Restated as data-parallel rounds:
//   * every edge costs >= 1 (1.0f or sqrtf(2.0f)), and f32 rounding is monotone, so a cell
//     pushed while popping cost c has cost >= fl(c + 1).  Hence, with L the smallest cost
//     in the open list, every cell whose cost is < fl(L + 1) is ALREADY in the open list:
//     the "window" of this round.
//   * the reference pops the window in the order (cost ascending, push order descending);
//     a bitonic sort gives each window cell its rank i.
//   * popping cell i pushes its undiscovered neighbours in the fixed enumeration order
//     j = 0..7 (utils/grid.cpp:29-47).  A neighbour n is discovered by the smallest (i, j)
//     that reaches it; the winner writes cost = cost_i + edge (same f32 add as the
//     reference) and the push order roundBase + i*8 + j, monotone in the reference's push time.
//
// What bounds this kernel (measured, profiles/r01_wavefront_*): not HBM bytes but the per-CU
// rate of uncoalesced cache-line requests -- window cells are ordered by cost, not by position,
// so every lane touches its own line.  The layout therefore minimises REQUESTS per cell:
//   * working grid = padded (rows+2) x (cols+2) f32 costs; occupied cells and the border hold
//     a NaN pattern, undiscovered cells +inf: one 12-byte row load yields occupancy AND
//     discovery state of three neighbours (3 requests per 3x3 neighbourhood, no bounds tests);
//   * discovery claims of a round are resolved in an LDS hash table (cell -> min (i*8+j)) with
//     LDS atomics -- no global traffic -- whenever the window fits (w <= 2048);
//   * (cost - L, ~pushOrder, cell) is packed into ONE 64-bit key, so the sort moves 8 B/element.
// Larger windows fall back to publishing (round, rank) per cell and gathering the seven other
// neighbours of n (exact, atomic-free, more requests); windows beyond LDS sort in HBM.
// One workgroup per goal.  Algorithmic bytes: 9 B/cell.
#include "pp_internal.hpp"

using namespace ppd;

namespace {

constexpr int WF_T = 512;          // 8 waves per goal
constexpr int WF_LCAP = 4096;      // LDS sort capacity (packed u64 keys): 32 KiB
constexpr int WF_HCAP = 4096;      // LDS hash slots (reuses the sort buffer): windows up to 2048 cells
constexpr uint32_t kInfBits = 0x7F800000u;
constexpr uint32_t kOccBits = 0x7FC00000u; // NaN pattern marking occupied / border cells in the working grid

struct WfSlot {
	uint32_t* grid;      // [(rows+2)*(cols+2)] working costs (padded)
	uint32_t* tag;       // [(rows+2)*(cols+2)] (round+1) << 17 | rank, fallback claim resolution
	uint32_t* fcell[2];  // open list ping-pong, [fcap]: padded cell index
	uint32_t* fcost[2];
	uint32_t* ford[2];
	uint64_t* gkeys;     // fallback sort buffers in HBM, [gcap] (gcap = pow2 >= fcap)
	uint32_t* gvals;
	uint32_t fcap, gcap;
};

__host__ __device__ inline int64_t padded_cells(int rows, int cols) { return (int64_t)(rows + 2) * ((cols + 2 + 7) & ~7); }

__device__ __forceinline__ WfSlot slot_view(void* base, int64_t bytesPerSlot, int slot, int64_t pcells, uint32_t fcap, uint32_t gcap)
{
	char* p = (char*)base + (int64_t)slot * bytesPerSlot;
	WfSlot s;
	const int64_t gbytes = (pcells * 4 + 255) / 256 * 256;
	s.grid = (uint32_t*)p;
	p += gbytes;
	s.tag = (uint32_t*)p;
	p += gbytes;
	for (int k = 0; k < 2; k++) {
		s.fcell[k] = (uint32_t*)p;
		p += (int64_t)fcap * 4;
		s.fcost[k] = (uint32_t*)p;
		p += (int64_t)fcap * 4;
		s.ford[k] = (uint32_t*)p;
		p += (int64_t)fcap * 4;
	}
	s.gkeys = (uint64_t*)p;
	p += (int64_t)gcap * 8;
	s.gvals = (uint32_t*)p;
	s.fcap = fcap;
	s.gcap = gcap;
	return s;
}

inline uint32_t next_pow2(uint32_t v)
{
	uint32_t p = 1;
	while (p < v)
		p <<= 1;
	return p;
}

// ---- wave-level ballot / prefix compaction: the lanes of a wave that want a slot reserve a contiguous range
// with ONE atomic on the shared counter (a per-lane atomicAdd on one LDS word serialises ~2.4k times per round).
__device__ __forceinline__ uint32_t wave_alloc(uint32_t* counter, bool want)
{
	const unsigned long long mask = __ballot(want);
	if (mask == 0ull)
		return 0u;
	const int lane = threadIdx.x & 63;
	const uint32_t prefix = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
	uint32_t base = 0;
	if (lane == __ffsll((long long)mask) - 1)
		base = atomicAdd(counter, (uint32_t)__popcll(mask));
	base = (uint32_t)__builtin_amdgcn_readlane((int)base, __ffsll((long long)mask) - 1);
	return base + prefix;
}

/// LDS position of sort element i: one slot of skew per 32 elements, so that power-of-two strides (the only
/// ones a bitonic network uses) spread over all banks instead of hitting 2-4 of them.
__device__ __forceinline__ int SK(int i) { return i + (i >> 5); }
constexpr int kSkewed(int n) { return n + (n >> 5); }

// ---- bitonic sorts (ascending), whole block.  Wave w owns the contiguous chunk of E = P/8 elements; a
// stage whose partner distance j satisfies 2*j <= E only moves data inside each wave's chunk, so two such
// consecutive stages need no block barrier (a wave runs in lockstep, its LDS operations complete in order).
__device__ __forceinline__ void cmpex1(uint64_t* keys, int t, int j, int lj, int k)
{
	const int i1 = ((t >> lj) << (lj + 1)) | (t & (j - 1)), i2 = i1 + j;
	const bool up = (i1 & k) == 0;
	const uint64_t a = keys[i1], b = keys[i2];
	const bool sw = (a > b) == up;
	keys[i1] = sw ? b : a;
	keys[i2] = sw ? a : b;
}
__device__ __forceinline__ void cmpex2(uint64_t* keys, uint32_t* vals, int t, int j, int lj, int k)
{
	const int i1 = ((t >> lj) << (lj + 1)) | (t & (j - 1)), i2 = i1 + j;
	const bool up = (i1 & k) == 0;
	const uint64_t a = keys[i1], b = keys[i2];
	const uint32_t va = vals[i1], vb = vals[i2];
	const bool sw = (a > b) == up;
	keys[i1] = sw ? b : a;
	keys[i2] = sw ? a : b;
	vals[i1] = sw ? vb : va;
	vals[i2] = sw ? va : vb;
}

template <bool kLds, bool kVals>
__device__ __forceinline__ void bitonic_sort(uint64_t* keys, uint32_t* vals, int P)
{
	const int tid = threadIdx.x;
	const int half = P >> 1;
	const int wave = tid >> 6, lane = tid & 63;
	const int E = P / (WF_T / 64);
	const bool canLocal = kLds && E >= 128;
	for (int k = 2; k <= P; k <<= 1) {
		for (int j = k >> 1, lj = 31 - __clz(k >> 1); j > 0; j >>= 1, lj--) {
			const bool local = canLocal && 2 * j <= E;
			if (local) {
				for (int t = wave * (E / 2) + lane; t < (wave + 1) * (E / 2); t += 64) {
					if (kVals)
						cmpex2(keys, vals, t, j, lj, k);
					else
						cmpex1(keys, t, j, lj, k);
				}
			} else {
				for (int t = tid; t < half; t += WF_T) {
					if (kVals)
						cmpex2(keys, vals, t, j, lj, k);
					else
						cmpex1(keys, t, j, lj, k);
				}
			}
			int nk = k, nj = j >> 1;
			if (nj == 0) {
				nk = k << 1;
				nj = k;
			}
			const bool nextLocal = nk <= P && canLocal && 2 * nj <= E;
			if (local && nextLocal) {
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				__builtin_amdgcn_wave_barrier();
			} else {
				__syncthreads();
			}
		}
	}
}

// ---- packed-key sort in LDS, several strides per pass.  A thread takes 2^S elements spaced by the smallest
// stride of the pass and applies S consecutive strides of the bitonic network in registers: one LDS round trip
// per pass instead of one per stride (the sort is LDS-latency bound: ~600 cycles per round trip at 8 waves).
template <int S>
__device__ __forceinline__ void bitonic_pass(uint64_t* keys, int g, int ljmin, int k)
{
	constexpr int N = 1 << S;
	const int jmin = 1 << ljmin;
	const int base = ((g >> ljmin) << (ljmin + S)) | (g & (jmin - 1));
	const bool up = (base & k) == 0; // all N elements lie in one 2*jmax block, so they share the direction
	uint64_t v[N];
#pragma unroll
	for (int e = 0; e < N; e++)
		v[e] = keys[SK(base + e * jmin)];
#pragma unroll
	for (int st = S - 1; st >= 0; st--) { // stride jmin << st
#pragma unroll
		for (int e = 0; e < N; e++) {
			if (!(e & (1 << st))) {
				const int f = e | (1 << st);
				const uint64_t a = v[e], b = v[f];
				const bool sw = (a > b) == up;
				v[e] = sw ? b : a;
				v[f] = sw ? a : b;
			}
		}
	}
#pragma unroll
	for (int e = 0; e < N; e++)
		keys[SK(base + e * jmin)] = v[e];
}

__device__ __forceinline__ void bitonic_sort_packed(uint64_t* keys, int P)
{
	const int tid = threadIdx.x;
	const int wave = tid >> 6, lane = tid & 63;
	const int E = P / (WF_T / 64);      // elements owned by one wave when a pass is wave-local
	const bool canLocal = E >= 512;     // >= 64 groups of 8 per wave
	bool prevLocal = false;
	for (int k = 2; k <= P; k <<= 1) {
		int lj = 31 - __clz(k >> 1); // log2 of the first (largest) stride of this merge step
		while (lj >= 0) {
			const int S = lj >= 2 ? 3 : lj + 1;      // strides 2^lj .. 2^(lj-S+1)
			const int ljmin = lj - S + 1;
			const int groups = P >> S;
			const bool local = canLocal && (2 << lj) <= E; // 2*jmax <= E: the pass stays inside each wave's chunk
			if (!(local && prevLocal))
				__syncthreads();
			else {
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				__builtin_amdgcn_wave_barrier();
			}
			if (local) {
				const int gpw = E >> S; // groups per wave
				for (int g = wave * gpw + lane; g < (wave + 1) * gpw; g += 64) {
					if (S == 3)
						bitonic_pass<3>(keys, g, ljmin, k);
					else if (S == 2)
						bitonic_pass<2>(keys, g, ljmin, k);
					else
						bitonic_pass<1>(keys, g, ljmin, k);
				}
			} else {
				for (int g = tid; g < groups; g += WF_T) {
					if (S == 3)
						bitonic_pass<3>(keys, g, ljmin, k);
					else if (S == 2)
						bitonic_pass<2>(keys, g, ljmin, k);
					else
						bitonic_pass<1>(keys, g, ljmin, k);
				}
			}
			prevLocal = local;
			lj -= S;
		}
	}
	__syncthreads();
}

// neighbour offsets in the reference's enumeration order (utils/grid.cpp:29-47)
constexpr int kDr[8] = { 0, -1, 1, 0, -1, 1, -1, 1 };
constexpr int kDc[8] = { -1, -1, -1, 1, 1, 1, 0, 0 };

/// three consecutive state bytes (c-1, c, c+1) of one padded row, fetched as ONE aligned 8-byte word:
/// base4 = (c-1) & ~3 always covers c-1 .. c+1 (offset <= 3, so offset + 2 <= 5 < 8)
struct Row3 {
	uint32_t a, b, c;
};
__device__ __forceinline__ Row3 load_state3(const uint8_t* rowBase, int cm1)
{
	const int base4 = cm1 & ~3;
	struct __attribute__((aligned(4))) U2 {
		uint32_t x, y;
	};
	const U2 v = *reinterpret_cast<const U2*>(rowBase + base4); // 4-byte aligned: row stride is a multiple of 8 bytes
	const unsigned long long w = ((unsigned long long)v.y << 32) | v.x;
	const int sh = (cm1 - base4) * 8;
	Row3 r;
	r.a = (uint32_t)(w >> sh) & 0xFFu;
	r.b = (uint32_t)(w >> (sh + 8)) & 0xFFu;
	r.c = (uint32_t)(w >> (sh + 16)) & 0xFFu;
	return r;
}
/// three consecutive 32-bit words (fallback path: rank words)
__device__ __forceinline__ Row3 load_row3(const uint32_t* p)
{
	const float3 v = *reinterpret_cast<const float3*>(p);
	Row3 r;
	r.a = __float_as_uint(v.x);
	r.b = __float_as_uint(v.y);
	r.c = __float_as_uint(v.z);
	return r;
}

constexpr uint32_t ST_FREE = 0u, ST_OCC = 1u, ST_SEEN = 2u; // state grid values (border = ST_OCC)

/// Candidate mask of a window cell from the 3x3 block of states around it (heuristics.cpp:127-136):
/// neighbour j is offered iff it is free and undiscovered and the corner rule allows the move.
__device__ __forceinline__ uint32_t candidate_mask(const Row3& up, const Row3& mid, const Row3& dn)
{
	const uint32_t nb[8] = { mid.a, up.a, dn.a, mid.c, up.c, dn.c, up.b, dn.b }; // order of kDr/kDc
	const bool oL = mid.a == ST_OCC, oR = mid.c == ST_OCC, oU = up.b == ST_OCC, oD = dn.b == ST_OCC;
	uint32_t mk = 0;
#pragma unroll
	for (int j = 0; j < 8; j++) {
		bool ok = nb[j] == ST_FREE; // not occupied, not yet in the open list nor explored
		// diagonal: blocked only if BOTH (n.row, cell.col) and (cell.row, n.col) are occupied (heuristics.cpp:130-132)
		if (j == 1)
			ok = ok && !(oU && oL);
		if (j == 2)
			ok = ok && !(oD && oL);
		if (j == 4)
			ok = ok && !(oU && oR);
		if (j == 5)
			ok = ok && !(oD && oR);
		if (ok)
			mk |= 1u << j;
	}
	return mk;
}

// kProfile: diagnostic build -- per goal {init, min, partition, sort, offer, push, tail} shader-clock sums + rounds, sum(w), sum(P)
enum { WP_INIT = 0, WP_MIN, WP_PART, WP_SORT, WP_OFFER, WP_PUSH, WP_TAIL, WP_ROUNDS, WP_SUMW, WP_SUMP, WP_FBROUNDS, WP_FBCYC, WP_COUNT };

constexpr int WF_LIST = 3072; // open-list entries kept in LDS (36 KiB); beyond that the list lives in HBM

#ifndef PP_WF_WAVES_PER_SIMD
#define PP_WF_WAVES_PER_SIMD 4 // 2 workgroups of 8 waves per CU: <= 128 VGPRs
#endif
template <bool kProfile>
__global__ void __launch_bounds__(WF_T, PP_WF_WAVES_PER_SIMD) k_wavefront(MapView m, int nGoals, const int32_t* __restrict__ goalCells, float* __restrict__ costOut,
	void* workspace, int64_t bytesPerSlot, uint32_t fcap, uint32_t gcap, int32_t* errorFlag, unsigned long long* __restrict__ prof, int* __restrict__ goalCounter)
{
	unsigned long long ph[WP_COUNT];
	unsigned long long tl = 0;
#define WF_STAMP(i)                                \
	if (kProfile) {                                \
		const unsigned long long now_ = clock64(); \
		ph[i] += now_ - tl;                        \
		tl = now_;                                 \
	}
	__shared__ uint64_t skey[kSkewed(WF_LCAP)]; // sort buffer (skewed layout), then (as two uint32 arrays) the claim hash table
	__shared__ uint64_t lco[WF_LIST];            // open list in LDS: cost bits << 32 | push order
	__shared__ uint32_t lcell[WF_LIST];          //                   padded cell index
	__shared__ uint32_t s_min, s_minNext, s_ordMin, s_ordMinNext, s_w, s_b, s_new, s_packFail, s_cand, s_distinct;
	__shared__ int s_goal;
	uint32_t* const hcell = reinterpret_cast<uint32_t*>(skey);          // [WF_HCAP] padded cell index + 1, 0 = empty
	uint32_t* const hkey = reinterpret_cast<uint32_t*>(skey) + WF_HCAP; // [WF_HCAP] min (i*8+j)

	const int tid = threadIdx.x;
	const int cols = m.cols, rows = m.rows;
	const int pc = (cols + 2 + 7) & ~7; // padded row stride of the state grid (bytes), multiple of 8
	const int64_t cells = (int64_t)rows * cols;
	const int64_t pcells = (int64_t)(rows + 2) * pc;
	WfSlot S = slot_view(workspace, bytesPerSlot, blockIdx.x, pcells, fcap, gcap);
	uint8_t* const state = reinterpret_cast<uint8_t*>(S.grid); // [(rows+2) * pc] bytes
	const float kDiag = sqrtf(2.0f); // std::sqrt(2.0f), heuristics.cpp:134
	const int nbOff[8] = { -1, -pc - 1, pc - 1, 1, -pc + 1, pc + 1, -pc, pc }; // padded-index offsets of kDr/kDc

	int tagGoal = -1; // goal whose fallback rounds the tag grid currently describes (it is cleared lazily)
	// goals are handed out dynamically: a workgroup that finishes early takes the next one (balanced tail)
	for (;;) {
		__syncthreads();
		if (tid == 0)
			s_goal = atomicAdd(goalCounter, 1);
		__syncthreads();
		const int g = s_goal;
		if (g >= nGoals)
			break;
		if (kProfile) {
			for (int i = 0; i < WP_COUNT; i++)
				ph[i] = 0;
			tl = clock64();
		}
		float* cost = costOut + (int64_t)g * cells;
		uint32_t* costBits = reinterpret_cast<uint32_t*>(cost);
		// ---- every cell starts at +inf / unexplored (heuristics.cpp:108-113); state: occupied cells and the border
		for (int64_t i = tid; i < cells; i += WF_T)
			costBits[i] = kInfBits;
		for (int pr = 0; pr < rows + 2; pr++) {
			const bool brow = pr == 0 || pr == rows + 1;
			for (int pcc = tid; pcc < pc; pcc += WF_T) {
				uint8_t v = (uint8_t)ST_OCC;
				if (!brow && pcc >= 1 && pcc <= cols)
					v = m.occ8[(int64_t)(pr - 1) * cols + (pcc - 1)] ? (uint8_t)ST_OCC : (uint8_t)ST_FREE;
				state[(int64_t)pr * pc + pcc] = v;
			}
		}
		const int32_t start = goalCells[g];
		if (tid == 0) {
			s_min = 0u; // cost bits of the start cell
			s_minNext = 0xFFFFFFFFu;
			s_ordMin = 0u;
			s_ordMinNext = 0xFFFFFFFFu;
			s_w = 0;
			s_b = 0;
			s_new = 0;
			s_packFail = 0;
			s_cand = 0;
			s_distinct = 0;
		}
		__syncthreads();
		if (start < 0)
			continue; // goal outside the map (heuristics.cpp:115-117): the field stays +inf
		if (tid == 0) {
			const int sr = start / cols, sc = start - sr * cols;
			const uint32_t sp = (uint32_t)((sr + 1) * pc + (sc + 1));
			state[sp] = (uint8_t)ST_SEEN; // the reference pushes the goal cell even when it is occupied
			cost[start] = 0.0f;
			lco[0] = 0ull; // cost 0, order 0
			lcell[0] = sp;
		}
		__syncthreads();

		WF_STAMP(WP_INIT);
		uint32_t n = 1;         // open-list size
		uint32_t roundBase = 1; // next push-order value
		uint32_t round = 0;
		bool inLds = true;      // where the open list lives
		int cur = 0;            // ping-pong index of the HBM list
		bool overflow = false;

		while (n > 0) {
			const int nxt = cur ^ 1;
			// ---- L = smallest cost in the open list (tracked while the list was written in the previous round)
			const uint32_t lBits = s_min;
			const float L = __uint_as_float(lBits);
			const uint32_t hiBits = __float_as_uint(L + 1.0f);
			const uint32_t ordFloor = s_ordMin;
			WF_STAMP(WP_MIN);
			// ---- partition: window (cost < fl(L+1)) -> sort buffer; the rest stays in the open list.
			// Window entries are packed as (cost - L : 23 | ~(order - floor) : 20 | cell : 21) in LDS; when a round does
			// not fit that encoding (or the LDS buffer) the window goes to the unpacked HBM buffers instead.
			const bool packable = pcells <= (1 << 21);
			bool fast = true;
			uint32_t w = 0, b = 0;
			for (int attempt = 0; attempt < 2; attempt++) {
				uint32_t restMin = 0xFFFFFFFFu, restOrd = 0xFFFFFFFFu;
				for (uint32_t i0 = 0; i0 < n; i0 += 8 * WF_T) {
					// up to 8 entries per thread are read before any is written back (in-place compaction of the LDS list)
					uint32_t ec[8], ecell[8], eord[8];
#pragma unroll
					for (int u = 0; u < 8; u++) {
						const uint32_t i = i0 + u * WF_T + tid;
						const bool in = i < n;
						if (inLds) {
							const uint64_t e = in ? lco[i] : ~0ull;
							ec[u] = (uint32_t)(e >> 32);
							eord[u] = (uint32_t)e;
							ecell[u] = in ? lcell[i] : 0u;
						} else {
							ec[u] = in ? S.fcost[cur][i] : 0xFFFFFFFFu;
							ecell[u] = in ? S.fcell[cur][i] : 0u;
							eord[u] = in ? S.ford[cur][i] : 0u;
						}
					}
					if (inLds)
						__syncthreads(); // all reads of this chunk done before survivors are compacted over it
					// one pair of LDS atomics per wave and chunk: ballots first, then the slots follow from lane prefixes
					unsigned long long bwM[8], brM[8];
					uint32_t wTot = 0, rTot = 0;
#pragma unroll
					for (int u = 0; u < 8; u++) {
						const bool in = i0 + u * WF_T + tid < n;
						bwM[u] = __ballot(in && ec[u] < hiBits);
						brM[u] = __ballot(in && !(ec[u] < hiBits));
						wTot += (uint32_t)__popcll(bwM[u]);
						rTot += (uint32_t)__popcll(brM[u]);
					}
					uint32_t wBase = 0, rBase = 0;
					if ((tid & 63) == 0) {
						if (wTot)
							wBase = atomicAdd(&s_w, wTot);
						if (rTot)
							rBase = atomicAdd(&s_b, rTot);
					}
					wBase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wBase);
					rBase = (uint32_t)__builtin_amdgcn_readfirstlane((int)rBase);
					const unsigned long long ltMask = (1ull << (tid & 63)) - 1ull;
#pragma unroll
					for (int u = 0; u < 8; u++) {
						const uint32_t i = i0 + u * WF_T + tid;
						const bool in = i < n;
						const uint32_t c = ec[u], cell = ecell[u], ord = eord[u];
						const bool inWindow = in && c < hiBits;
						const bool inRest = in && !inWindow;
						const uint32_t wslot = wBase + (uint32_t)__popcll(bwM[u] & ltMask);
						const uint32_t bslot = rBase + (uint32_t)__popcll(brM[u] & ltMask);
						wBase += (uint32_t)__popcll(bwM[u]);
						rBase += (uint32_t)__popcll(brM[u]);
						if (inWindow) {
							if (fast) {
								const uint32_t rel = ord - ordFloor;
								if (!packable || rel >= (1u << 20) || (c - lBits) >= (1u << 23))
									s_packFail = 1;
								if (wslot < WF_LCAP)
									skey[SK((int)wslot)] = ((uint64_t)(c - lBits) << 41) | ((uint64_t)(0xFFFFFu - (rel & 0xFFFFFu)) << 21) | (uint64_t)cell;
							} else if (wslot < S.gcap) {
								S.gkeys[wslot] = ((uint64_t)c << 32) | (uint64_t)(0xFFFFFFFFu - ord);
								S.gvals[wslot] = cell;
							}
						} else if (inRest) {
							if (inLds) {
								lco[bslot] = ((uint64_t)c << 32) | ord; // bslot <= i: never overtakes an unread entry
								lcell[bslot] = cell;
							} else {
								S.fcell[nxt][bslot] = cell; // bslot < n <= fcap
								S.fcost[nxt][bslot] = c;
								S.ford[nxt][bslot] = ord;
							}
							restMin = min(restMin, c);
							restOrd = min(restOrd, ord);
						}
					}
					if (inLds)
						__syncthreads();
				}
				if (restMin != 0xFFFFFFFFu) {
					atomicMin(&s_minNext, restMin);
					atomicMin(&s_ordMinNext, restOrd);
				}
				__syncthreads();
				w = s_w;
				b = s_b;
				const bool ok = !s_packFail && w <= WF_LCAP;
				__syncthreads();
				if (!fast || ok)
					break;
				// Redo into the HBM buffers.  An in-place LDS compaction cannot be replayed, so the redo is only taken
				// while the list is in HBM; with the list in LDS the window entries are recovered from the packed keys
				// (n <= WF_LIST <= WF_LCAP there, so they all fitted) -- only the encoding overflowed.
				fast = false;
				if (inLds) {
					overflow = true; // encoding overflow with an LDS-resident list: cannot happen for grids <= 2^21 padded cells
					break;
				}
				if (tid == 0) {
					s_w = 0;
					s_b = 0;
				}
				__syncthreads();
			}
			WF_STAMP(WP_PART);
			if (overflow || w > S.gcap || round + 1u >= (1u << 15) || w > (1u << 17)) {
				overflow = true;
				break;
			}
			const uint32_t P = w <= 1 ? 2 : (1u << (32 - __clz((int)(w - 1))));
			if (fast) {
				for (uint32_t i = w + tid; i < (P < 8 ? 8u : P); i += WF_T)
					skey[SK((int)i)] = ~0ull;
				__syncthreads();
				if (w > 1)
					bitonic_sort_packed(skey, (int)(P < 8 ? 8 : P));
			} else {
				for (uint32_t i = w + tid; i < P; i += WF_T)
					S.gkeys[i] = ~0ull;
				__syncthreads();
				bitonic_sort<false, true>(S.gkeys, S.gvals, (int)P);
			}
			WF_STAMP(WP_SORT);
			if (kProfile) {
				ph[WP_ROUNDS]++;
				ph[WP_SUMW] += w;
				ph[WP_SUMP] += P;
			}
			uint32_t newMin = 0xFFFFFFFFu, newOrd = 0xFFFFFFFFu;
			// Where do this round's pushes go?  They stay in LDS while survivors + every push are known to fit
			// (decided below, once the number of discovered cells is known or bounded).
			bool pushLds = false;
			// appends one discovered cell to the open list (slot relative to the survivors)
			auto push_entry = [&](uint32_t slot, uint32_t ncell, uint32_t pb, uint32_t ord) {
				if (pushLds) {
					lco[slot] = ((uint64_t)pb << 32) | ord;
					lcell[slot] = ncell;
				} else if (slot < S.fcap) {
					S.fcell[nxt][slot] = ncell;
					S.fcost[nxt][slot] = pb;
					S.ford[nxt][slot] = ord;
				}
			};
			// the list leaves LDS before the pushes when they might not fit: survivors are copied to HBM once
			auto spill_list = [&]() {
				for (uint32_t i = tid; i < b; i += WF_T) {
					const uint64_t e = lco[i];
					S.fcell[nxt][i] = lcell[i];
					S.fcost[nxt][i] = (uint32_t)(e >> 32);
					S.ford[nxt][i] = (uint32_t)e;
				}
			};
			// The claim table has WF_HCAP slots; the round may use it only when every candidate (counted with
			// duplicates, so an upper bound on distinct cells) fits with room to spare: insertion then always ends.
			uint32_t myCell[4] = { 0, 0, 0, 0 }, myCost[4] = { 0, 0, 0, 0 }, myMask[4] = { 0, 0, 0, 0 }, myOut[4] = { 0, 0, 0, 0 };
			bool hashed = false;
			if (fast && w <= 4u * WF_T) {
				Row3 up[4], mid[4], dn[4];
#pragma unroll
				for (int q = 0; q < 4; q++) {
					const uint32_t i = tid + q * WF_T;
					if (i < w) {
						const uint64_t k = skey[SK((int)i)];
						myCell[q] = (uint32_t)(k & 0x1FFFFFu);
						myCost[q] = lBits + (uint32_t)(k >> 41);
						const int pr = (int)(myCell[q] / (uint32_t)pc), pcc = (int)(myCell[q] - (uint32_t)pr * (uint32_t)pc);
						const uint8_t* rowBase = state + (int64_t)pr * pc;
						myOut[q] = (uint32_t)((pr - 1) * cols + (pcc - 1)); // index of the cell in the (unpadded) output field
						up[q] = load_state3(rowBase - pc, pcc - 1);
						mid[q] = load_state3(rowBase, pcc - 1);
						dn[q] = load_state3(rowBase + pc, pcc - 1);
					}
				}
				uint32_t cnt = 0;
#pragma unroll
				for (int q = 0; q < 4; q++) {
					const uint32_t i = tid + q * WF_T;
					if (i < w) {
						myMask[q] = candidate_mask(up[q], mid[q], dn[q]);
						cnt += __popc(myMask[q]);
					}
				}
				for (int off = 32; off > 0; off >>= 1)
					cnt += __shfl_xor((int)cnt, off, 64);
				if ((tid & 63) == 0 && cnt)
					atomicAdd(&s_cand, cnt);
				__syncthreads();
				hashed = s_cand <= (uint32_t)(WF_HCAP * 3 / 4);
			}
			if (hashed) {
				// ================= fast path: claims in an LDS hash table (the sort buffer is reused) =================
				for (int i = tid; i < WF_HCAP; i += WF_T) {
					hcell[i] = 0u;
					hkey[i] = 0xFFFFFFFFu;
				}
				__syncthreads();
#pragma unroll
				for (int q = 0; q < 4; q++) {
					const uint32_t i = tid + q * WF_T;
					const uint32_t mk = myMask[q];
					if (i < w && mk) {
#pragma unroll
						for (int j = 0; j < 8; j++) {
							if (!(mk & (1u << j)))
								continue;
							const uint32_t ncell = myCell[q] + (uint32_t)nbOff[j];
							const uint32_t key = i * 8u + (uint32_t)j;
							uint32_t h = (ncell * 2654435761u) >> (32 - 12); // WF_HCAP = 4096 = 2^12
							for (;;) {
								const uint32_t old = atomicCAS(&hcell[h], 0u, ncell + 1u);
								if (old == 0u || old == ncell + 1u) {
									atomicMin(&hkey[h], key);
									break;
								}
								h = (h + 1) & (WF_HCAP - 1);
							}
						}
					}
				}
				__syncthreads();
				{ // number of distinct discovered cells = occupied table slots: decides whether the pushes stay in LDS
					uint32_t cnt = 0;
					for (int i = tid; i < WF_HCAP; i += WF_T)
						cnt += hcell[i] != 0u;
					for (int off = 32; off > 0; off >>= 1)
						cnt += __shfl_xor((int)cnt, off, 64);
					if ((tid & 63) == 0 && cnt)
						atomicAdd(&s_distinct, cnt);
					__syncthreads();
					pushLds = inLds && b + s_distinct <= (uint32_t)WF_LIST;
					if (inLds && !pushLds)
						spill_list();
				}
				WF_STAMP(WP_OFFER);
				// push the winners: cost fixed at discovery (Q3), push order = roundBase + i*8 + j.  Each lane first finds
				// its wins (bit q*8+j), a wave scan then hands out consecutive open-list slots with ONE LDS atomic per wave.
				uint32_t winBits = 0;
#pragma unroll
				for (int q = 0; q < 4; q++) {
					const uint32_t i = tid + q * WF_T;
					const uint32_t mk = (i < w) ? myMask[q] : 0u;
#pragma unroll
					for (int j = 0; j < 8; j++) {
						if (!(mk & (1u << j)))
							continue;
						const uint32_t ncell = myCell[q] + (uint32_t)nbOff[j];
						uint32_t h = (ncell * 2654435761u) >> (32 - 12);
						while (hcell[h] != ncell + 1u)
							h = (h + 1) & (WF_HCAP - 1);
						if (hkey[h] == i * 8u + (uint32_t)j)
							winBits |= 1u << (q * 8 + j);
					}
				}
				const uint32_t nWin = (uint32_t)__popc(winBits);
				uint32_t incl = nWin;
#pragma unroll
				for (int d = 1; d < 64; d <<= 1) {
					const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
					if ((tid & 63) >= d)
						incl += t;
				}
				uint32_t wbase = 0;
				if ((tid & 63) == 63 && incl)
					wbase = atomicAdd(&s_new, incl);
				wbase = (uint32_t)__builtin_amdgcn_readlane((int)wbase, 63);
				uint32_t slot = b + wbase + incl - nWin;
#pragma unroll
				for (int q = 0; q < 4; q++) {
					if (!((winBits >> (q * 8)) & 0xFFu))
						continue;
					const uint32_t i = tid + q * WF_T;
					const float ci = __uint_as_float(myCost[q]);
#pragma unroll
					for (int j = 0; j < 8; j++) {
						if (!(winBits & (1u << (q * 8 + j))))
							continue;
						const uint32_t ncell = myCell[q] + (uint32_t)nbOff[j];
						const uint32_t key = i * 8u + (uint32_t)j;
						const float transitionCost = (kDr[j] == 0 || kDc[j] == 0) ? 1.0f : kDiag;
						const float pathCost = transitionCost + ci; // heuristics.cpp:135
						const uint32_t pb = __float_as_uint(pathCost);
						state[ncell] = (uint8_t)ST_SEEN;
						cost[(int)myOut[q] + kDr[j] * cols + kDc[j]] = pathCost;
						newMin = min(newMin, pb);
						newOrd = min(newOrd, roundBase + key);
						push_entry(slot++, ncell, pb, roundBase + key);
					}
				}
			} else {
				// ================= fallback: publish (round, rank), gather the other neighbours of n =================
				pushLds = inLds && b + 8u * w <= (uint32_t)WF_LIST;
				if (inLds && !pushLds)
					spill_list();
				if (tagGoal != g) { // first fallback round of this goal: stale (round, rank) words of earlier goals must go
					for (int64_t i = tid; i < pcells; i += WF_T)
						S.tag[i] = 0u;
					__syncthreads();
					tagGoal = g;
				}
				const uint32_t roundTag = (round + 1u) << 17;
				for (uint32_t i = tid; i < w; i += WF_T) {
					const uint32_t cell = fast ? (uint32_t)(skey[SK((int)i)] & 0x1FFFFFu) : S.gvals[i];
					S.tag[cell] = roundTag | i;
				}
				__syncthreads();
				WF_STAMP(WP_OFFER);
				for (uint32_t i = tid; i < w; i += WF_T) {
					uint32_t cell, cbits;
					if (fast) {
						const uint64_t k = skey[SK((int)i)];
						cell = (uint32_t)(k & 0x1FFFFFu);
						cbits = lBits + (uint32_t)(k >> 41);
					} else {
						cell = S.gvals[i];
						cbits = (uint32_t)(S.gkeys[i] >> 32);
					}
					const float ci = __uint_as_float(cbits);
					const int pr = (int)(cell / (uint32_t)pc), pcc = (int)(cell - (uint32_t)pr * (uint32_t)pc);
					const uint8_t* rowBase = state + (int64_t)pr * pc;
					const Row3 up = load_state3(rowBase - pc, pcc - 1), mid = load_state3(rowBase, pcc - 1), dn = load_state3(rowBase + pc, pcc - 1);
					const uint32_t mk = candidate_mask(up, mid, dn);
#pragma unroll
					for (int j = 0; j < 8; j++) {
						if (!(mk & (1u << j)))
							continue;
						// n = neighbour j.  Is another window cell the first to reach n?
						const uint32_t ncell = cell + (uint32_t)nbOff[j];
						const int nr = pr + kDr[j], nc = pcc + kDc[j];
						const uint32_t mine = i * 8u + (uint32_t)j;
						const Row3 tu = load_row3(S.tag + ncell - pc - 1), tm = load_row3(S.tag + ncell - 1), td = load_row3(S.tag + ncell + pc - 1);
						const uint8_t* nrow = state + (int64_t)nr * pc;
						const Row3 gu = load_state3(nrow - pc, nc - 1), gm = load_state3(nrow, nc - 1), gd = load_state3(nrow + pc, nc - 1);
						// p'' = n - d_jj reaches n through direction jj: p'' sits at offset -d_jj from n
						const uint32_t tg[8] = { tm.c, td.c, tu.c, tm.a, td.a, tu.a, td.b, tu.b };
						const bool oL = gm.a == ST_OCC, oR = gm.c == ST_OCC, oU = gu.b == ST_OCC, oD = gd.b == ST_OCC; // n's orthogonal neighbours
						bool win = true;
#pragma unroll
						for (int jj = 0; jj < 8; jj++) {
							if (jj == j || (tg[jj] & 0xFFFE0000u) != roundTag)
								continue; // itself, or not popped in this round
							// corner rule for p'' -> n: (n.row, p''.col) = (nr, nc - dc) and (p''.row, n.col) = (nr - dr, nc)
							bool allowed = true;
							if (kDr[jj] != 0 && kDc[jj] != 0) {
								const bool oc = kDc[jj] > 0 ? oL : oR;
								const bool orr = kDr[jj] > 0 ? oU : oD;
								allowed = !(oc && orr);
							}
							const uint32_t other = (tg[jj] & 0x1FFFFu) * 8u + (uint32_t)jj;
							if (allowed && other < mine)
								win = false;
						}
						if (!win)
							continue;
						const float transitionCost = (kDr[j] == 0 || kDc[j] == 0) ? 1.0f : kDiag;
						const float pathCost = transitionCost + ci; // heuristics.cpp:135
						const uint32_t pb = __float_as_uint(pathCost);
						state[ncell] = (uint8_t)ST_SEEN;
						cost[(int64_t)(nr - 1) * cols + (nc - 1)] = pathCost;
						newMin = min(newMin, pb);
						newOrd = min(newOrd, roundBase + mine);
						const uint32_t slot = b + atomicAdd(&s_new, 1u);
						push_entry(slot, ncell, pb, roundBase + mine);
					}
				}
			}
			if (newMin != 0xFFFFFFFFu) {
				atomicMin(&s_minNext, newMin);
				atomicMin(&s_ordMinNext, newOrd);
			}
			__syncthreads();
			if (kProfile && !hashed) {
				ph[WP_FBROUNDS]++;
				ph[WP_FBCYC] += clock64() - tl;
			}
			WF_STAMP(WP_PUSH);
			const uint32_t nn = b + s_new;
			const uint32_t nextMin = s_minNext, nextOrd = s_ordMinNext;
			__syncthreads();
			if (tid == 0) {
				s_min = nextMin;
				s_minNext = 0xFFFFFFFFu;
				s_ordMin = nextOrd;
				s_ordMinNext = 0xFFFFFFFFu;
				s_w = 0;
				s_b = 0;
				s_new = 0;
				s_packFail = 0;
				s_cand = 0;
				s_distinct = 0;
			}
			__syncthreads();
			if (nn > S.fcap) {
				overflow = true;
				break;
			}
			// where the list lives next round
			if (inLds && !pushLds) {
				inLds = false; // it was moved to S.f*[nxt] above
				cur = nxt;
			} else if (!inLds) {
				cur = nxt;
				if (nn <= (uint32_t)WF_LIST / 4) { // small again: bring it back into LDS
					for (uint32_t i = tid; i < nn; i += WF_T) {
						lco[i] = ((uint64_t)S.fcost[cur][i] << 32) | S.ford[cur][i];
						lcell[i] = S.fcell[cur][i];
					}
					__syncthreads();
					inLds = true;
				}
			}
			roundBase += w * 8u;
			round++;
			n = nn;
			WF_STAMP(WP_TAIL);
		}
		if (overflow && tid == 0)
			*errorFlag = 1; // open list / round count beyond the workspace encoding
		__syncthreads();
		if (kProfile && tid == 0)
			for (int i = 0; i < WP_COUNT; i++)
				prof[(size_t)g * WP_COUNT + i] = ph[i];
	}
#undef WF_STAMP
}

} // namespace

namespace pph {

static void wf_caps(int rows, int cols, uint32_t& fcap, uint32_t& gcap)
{
	uint64_t cells = (uint64_t)rows * cols;
	uint64_t f = 16ull * (uint64_t)(rows + cols) + 4096;
	if (f > cells + 8)
		f = cells + 8;
	fcap = (uint32_t)f;
	gcap = next_pow2(fcap);
}

int64_t wavefront_workspace_bytes(int rows, int cols)
{
	uint32_t fcap, gcap;
	wf_caps(rows, cols, fcap, gcap);
	const int64_t gbytes = (padded_cells(rows, cols) * 4 + 255) / 256 * 256;
	int64_t b = 2 * gbytes + 6ll * fcap * 4 + (int64_t)gcap * 12 + 16;
	return (b + 255) / 256 * 256;
}

int wavefront_resident_blocks()
{
	int perCu = 0, dev = 0;
	hipDeviceProp_t prop;
	if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
		return 256;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_wavefront<false>, WF_T, 0) != hipSuccess || perCu < 1)
		perCu = 1;
	return perCu * prop.multiProcessorCount;
}

hipError_t launch_wavefront(hipStream_t s, const MapView& m, int nGoals, const int32_t* goalCellsDev, float* costDev, void* workspaceDev,
	int64_t workspaceBytesPerSlot, int nSlots, int32_t* errorFlagDev, unsigned long long* profDev)
{
	if (nGoals <= 0)
		return hipSuccess;
	uint32_t fcap, gcap;
	wf_caps(m.rows, m.cols, fcap, gcap);
	int grid = nGoals < nSlots ? nGoals : nSlots;
	// errorFlagDev[0] = overflow flag, errorFlagDev[1] = next-goal counter
	hipError_t e = hipMemsetAsync(errorFlagDev + 1, 0, sizeof(int), s);
	if (e != hipSuccess)
		return e;
	if (profDev)
		hipLaunchKernelGGL(k_wavefront<true>, dim3(grid), dim3(WF_T), 0, s, m, nGoals, goalCellsDev, costDev, workspaceDev, workspaceBytesPerSlot, fcap, gcap, errorFlagDev,
			profDev, (int*)(errorFlagDev + 1));
	else
		hipLaunchKernelGGL(k_wavefront<false>, dim3(grid), dim3(WF_T), 0, s, m, nGoals, goalCellsDev, costDev, workspaceDev, workspaceBytesPerSlot, fcap, gcap, errorFlagDev,
			profDev, (int*)(errorFlagDev + 1));
	return hipGetLastError();
}

} // namespace pph
