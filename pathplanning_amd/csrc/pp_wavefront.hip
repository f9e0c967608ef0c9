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
//     a rank sort gives each window cell its rank i.
//   * popping cell i pushes its undiscovered neighbours in the fixed enumeration order
//     j = 0..7 (utils/grid.cpp:29-47).  A neighbour n is discovered by the smallest (i, j)
//     that reaches it; the winner writes cost = cost_i + edge (same f32 add as the
//     reference) and is appended to the open list at its place in the reference's push order.
//
// Layout and mechanics (measured history in DESIGN.md 4.3 and profiles/r01_wavefront_*):
//   * the open list is a SEQUENCE in push order (8-byte entries cost | row,col), kept in LDS while it has
//     <= 4096 entries, else ping-pong in HBM; stable compaction by ballots + a DPP scan, appends by a block
//     scan of win counts -- position in the list replaces an explicit push-order field;
//   * window entries are packed into ONE 64-bit key (cost - L : 23 | ~position : 19 | row,col : 22) and
//     ranked by a bucketed counting sort in LDS (7 LDS-only barriers);
//   * discovery claims of a round are resolved in an LDS hash table (cell -> min (i*8+j)) whenever the
//     window fits (w <= 2048); larger windows publish (round, rank) per cell and gather the seven other
//     neighbours of n (exact, atomic-free), windows beyond LDS sort in HBM;
//   * state = one byte per cell (free / occupied+border / discovered) in 8 x 8 tiles of one cache line: a
//     3 x 3 neighbourhood is three 8-byte loads; the cost is stored once, at discovery, into the output field
//     (8 x 8-tiled for the search kernel, row-major for the public entry points);
//   * barriers that only order LDS traffic do not drain outstanding global stores.
// One workgroup per goal, two per CU.  Algorithmic bytes: 9 B/cell.
#include "pp_internal.hpp"

#include <cstdlib>

using namespace ppd;

namespace {

constexpr int WF_T = 512;          // 8 waves per goal
constexpr int WF_LCAP = 4096;      // LDS sort capacity (packed u64 keys): 32 KiB
constexpr int WF_HCAP = 4096;      // LDS hash slots (reuses the sort buffer): windows up to 2048 cells
constexpr uint32_t kInfBits = 0x7F800000u;
constexpr uint32_t kOccBits = 0x7FC00000u; // NaN pattern marking occupied / border cells in the working grid

struct WfSlot {
	uint8_t* state;      // 8 x 8-tiled padded grid, one byte per cell: free / occupied (and border) / discovered
	uint32_t* tag;       // [(rows+2) * (cols+2)] row-major: (round+1) << 17 | rank, fallback claim resolution
	uint64_t* fent[2];   // open list in HBM (ping-pong), [fcap]: cost bits << 32 | padded cell, in push order
	uint64_t* gkeys;     // fallback sort buffers in HBM, [gcap] (gcap = pow2 >= fcap)
	uint32_t* gvals;
	uint32_t fcap, gcap;
};

// The state grid is padded by one border cell on every side and stored in 8 x 8 tiles of 64 bytes (one cache
// line): a ring of the wavefront crosses a tile during ~8 consecutive rounds, so the lines the 3 x 3 neighbourhood
// reads of a goal touch stay few (a row-major grid has a vertical front touch three new lines per cell, and with
// 64 goals sharing a 4 MiB L2 every such read went to HBM: 181 B fetched per cell, measured).
__host__ __device__ inline int state_tiles_per_row(int cols) { return (cols + 2 + 7) >> 3; }
__host__ __device__ inline int64_t state_bytes(int rows, int cols) { return (int64_t)((rows + 2 + 7) >> 3) * state_tiles_per_row(cols) * 64; }
__host__ __device__ inline int64_t padded_cells(int rows, int cols) { return (int64_t)(rows + 2) * (cols + 2); }
__host__ __device__ inline int64_t round256(int64_t b) { return (b + 255) / 256 * 256; }
/// byte address of padded cell (pr, pcc) in the tiled state grid
__device__ __forceinline__ uint32_t st_addr(int tpr, int pr, int pcc) { return ((uint32_t)((pr >> 3) * tpr + (pcc >> 3)) << 6) | (uint32_t)(((pr & 7) << 3) | (pcc & 7)); }
// cells travel as (padded row << 16 | padded col); the packed window key holds them in 2 * cb bits (cb = 11 for grids
// up to 2045 x 2045, else 13)
__device__ __forceinline__ uint32_t pack_cell(uint32_t cell, int cb) { return ((cell >> 16) << cb) | (cell & 0xFFFFu); }
__device__ __forceinline__ uint32_t unpack_cell(uint32_t k, int cb) { return ((k >> cb) << 16) | (k & ((1u << cb) - 1u)); }

__device__ __forceinline__ WfSlot slot_view(void* base, int64_t bytesPerSlot, int slot, int64_t stBytes, int64_t pcells, uint32_t fcap, uint32_t gcap)
{
	char* p = (char*)base + (int64_t)slot * bytesPerSlot;
	WfSlot s;
	s.state = (uint8_t*)p;
	p += round256(stBytes);
	s.tag = (uint32_t*)p;
	p += round256(pcells * 4 + 16);
	for (int k = 0; k < 2; k++) {
		s.fent[k] = (uint64_t*)p;
		p += (int64_t)fcap * 8;
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

// neighbour offsets in the reference's enumeration order (utils/grid.cpp:29-47)
constexpr int kDr[8] = { 0, -1, 1, 0, -1, 1, -1, 1 };
constexpr int kDc[8] = { -1, -1, -1, 1, 1, 1, 0, 0 };

/// kDr[j] / kDc[j] for a run-time j without a memory table: 2-bit fields (value + 1)
__device__ __forceinline__ int dir_dr(int j) { return (int)((0x8861u >> (2 * j)) & 3u) - 1; } // {0,-1,1,0,-1,1,-1,1} + 1 = {1,0,2,1,0,2,0,2}
__device__ __forceinline__ int dir_dc(int j) { return (int)((0x5A80u >> (2 * j)) & 3u) - 1; } // {-1,-1,-1,1,1,1,0,0} + 1 = {0,0,0,2,2,2,1,1}

struct Row3 {
	uint32_t a, b, c;
};
/// States of the 3 x 3 block around padded cell (pr, pcc) from the tiled grid.  A tile row is 8 aligned bytes, so each
/// of the three rows is ONE 8-byte load; only cells in the first / last column of a tile (1 in 4) also need the
/// adjacent tile's edge byte.  Issue and use are separate so that the loads can fly under other work.
struct NbhdRaw {
	unsigned long long w[3];
	uint32_t side[3];
	int x;
};
__device__ __forceinline__ NbhdRaw nbhd_issue(const uint8_t* __restrict__ state, int tpr, int pr, int pcc)
{
	NbhdRaw r;
	r.x = pcc & 7;
	uint32_t base[3];
#pragma unroll
	for (int k = 0; k < 3; k++) {
		const int rr = pr - 1 + k;
		base[k] = ((uint32_t)((rr >> 3) * tpr + (pcc >> 3)) << 6) | (uint32_t)((rr & 7) << 3);
		r.w[k] = *reinterpret_cast<const unsigned long long*>(state + base[k]);
		r.side[k] = 0;
	}
	if (r.x == 0 || r.x == 7) {
		const int off = r.x ? 64 : -57; // same row of the next tile (its byte 0) / of the previous tile (its byte 7)
#pragma unroll
		for (int k = 0; k < 3; k++)
			r.side[k] = state[(int)base[k] + off];
	}
	return r;
}
__device__ __forceinline__ Row3 nbhd_row(unsigned long long w, uint32_t side, int x)
{
	// no arrays of pointers / values here: anything indexable ends up in scratch memory (measured: 28 scratch stores and 16
	// loads per window cell in the round loop, ~280 GB of write-backs per 4096-goal launch)
	const uint32_t lo = (uint32_t)(w >> ((x ? x - 1 : 0) * 8)) & 0xFFu;
	const uint32_t hi = (uint32_t)(w >> ((x < 7 ? x + 1 : 7) * 8)) & 0xFFu;
	Row3 r;
	r.a = x == 0 ? side : lo;
	r.b = (uint32_t)(w >> (x * 8)) & 0xFFu;
	r.c = x == 7 ? side : hi;
	return r;
}
__device__ __forceinline__ void nbhd_finish(const NbhdRaw& r, Row3& up, Row3& mid, Row3& dn)
{
	up = nbhd_row(r.w[0], r.side[0], r.x);
	mid = nbhd_row(r.w[1], r.side[1], r.x);
	dn = nbhd_row(r.w[2], r.side[2], r.x);
}
__device__ __forceinline__ void load_state_nbhd(const uint8_t* __restrict__ state, int tpr, int pr, int pcc, Row3& up, Row3& mid, Row3& dn)
{
	const NbhdRaw r = nbhd_issue(state, tpr, pr, pcc);
	nbhd_finish(r, up, mid, dn);
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

// experiment switches (tools/build_variant.py): the cost field is written once per cell and never read by this kernel
#ifndef PP_WF_NT_COST
#define PP_WF_NT_COST 0
#endif
#ifndef PP_WF_DEFER_STORES
#define PP_WF_DEFER_STORES 1 // v11: the state / cost stores of a round's discoveries are issued densely at the start of the next round
#endif
#ifndef PP_WF_NT_FILL
#define PP_WF_NT_FILL 0
#endif
__device__ __forceinline__ void store_cost(float* p, float v)
{
#if PP_WF_NT_COST
	__builtin_nontemporal_store(v, p);
#else
	*p = v;
#endif
}

/// Candidate mask of a window cell from the 3x3 block of states around it (heuristics.cpp:127-136):
/// neighbour j is offered iff it is free and undiscovered and the corner rule allows the move.
__device__ __forceinline__ uint32_t candidate_mask(const Row3& up, const Row3& mid, const Row3& dn)
{
	const bool oL = mid.a == ST_OCC, oR = mid.c == ST_OCC, oU = up.b == ST_OCC, oD = dn.b == ST_OCC;
	// bit j = neighbour j in the order of kDr/kDc: free and undiscovered (not occupied, not yet in the open list nor explored)
	uint32_t mk = (mid.a == ST_FREE ? 1u : 0u) | (mid.c == ST_FREE ? 8u : 0u) | (up.b == ST_FREE ? 64u : 0u) | (dn.b == ST_FREE ? 128u : 0u);
	// diagonal: blocked only if BOTH (n.row, cell.col) and (cell.row, n.col) are occupied (heuristics.cpp:130-132)
	mk |= (up.a == ST_FREE && !(oU && oL)) ? 2u : 0u;
	mk |= (dn.a == ST_FREE && !(oD && oL)) ? 4u : 0u;
	mk |= (up.c == ST_FREE && !(oU && oR)) ? 16u : 0u;
	mk |= (dn.c == ST_FREE && !(oD && oR)) ? 32u : 0u;
	return mk;
}

/// +inf into the output field for every map cell whose state byte is not "discovered" (one pass over the tiled state grid when a goal
/// is done; see k_wavefront).  NOT inlined on purpose: inlined, its registers cost the round loop eight scratch stores per round.
__device__ __attribute__((noinline)) void fill_unreached(const uint8_t* __restrict__ state, int64_t stWords, int tpr, int rows, int cols, int tiledOut, float* __restrict__ cost, int tid)
{
	for (int64_t t = tid; t < stWords; t += WF_T) {
		const unsigned long long v = reinterpret_cast<const unsigned long long*>(state)[t];
		const int tile = (int)(t >> 3), trow = (int)(t & 7);
		const int tr = tile / tpr, tc = tile - tr * tpr;
		const int r = (tr << 3) + trow - 1;
		if (r < 0 || r >= rows)
			continue;
#pragma unroll
		for (int x = 0; x < 8; x++) {
			const int c = (tc << 3) + x - 1;
			if (c >= 0 && c < cols && (uint32_t)((v >> (8 * x)) & 0xFFull) != ST_SEEN)
				reinterpret_cast<uint32_t*>(cost)[tiledOut ? field_tiled_index(cols, r, c) : (size_t)r * cols + c] = kInfBits;
		}
	}
}

// kProfile: diagnostic build -- per goal {init, min, partition, sort, offer, push, tail} shader-clock sums + rounds, sum(w), sum(P)
enum { WP_INIT = 0, WP_MIN, WP_PART, WP_SORT, WP_OFFER, WP_PUSH, WP_TAIL, WP_ROUNDS, WP_SUMW, WP_SUMP, WP_FBROUNDS, WP_FBCYC, WP_O_WAIT, WP_O_LOAD, WP_O_COUNT, WP_O_INSERT, WP_P_LOOKUP, WP_P_SCAN, WP_P_STORE, WP_P_PAD, WP_COUNT };

constexpr int WF_LIST = 4096; // open-list entries kept in LDS (32 KiB); beyond that the list lives in HBM
constexpr int WF_NB = 2048;   // buckets of the rank sort
constexpr int WF_W = WF_T / 64;

/// workgroup barrier that orders LDS traffic only: global stores in flight are NOT waited for (a __syncthreads()
/// drains vmcnt, i.e. costs a full memory round trip whenever scattered stores are pending)
__device__ __forceinline__ void lds_barrier()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
	__builtin_amdgcn_s_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

/// inclusive prefix sum over the 64 lanes of a wave in DPP steps (no LDS): Hillis-Steele inside each row of 16,
/// then the row totals are carried across with row_bcast15 / row_bcast31
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v)
{
	uint32_t x = v;
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false); // row_shr:1
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false); // row_shr:2
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false); // row_shr:4
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false); // row_shr:8
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false); // row_bcast15 -> rows 1, 3
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false); // row_bcast31 -> rows 2, 3
	return x;
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v)
{
	const int id = (int)0xFFFFFFFF;
	uint32_t x = v;
	x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x111, 0xF, 0xF, false));
	x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x112, 0xF, 0xF, false));
	x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x114, 0xF, 0xF, false));
	x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x118, 0xF, 0xF, false));
	x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x142, 0xA, 0xF, false));
	x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x143, 0xC, 0xF, false));
	return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

/// Rank sort of the w <= 4096 packed window keys in skey (ascending), whole block.  The bitonic network needs
/// ~20 dependent LDS round trips; here the keys are bucketed on the leading bits of (cost - L) with an LDS
/// histogram (the atomic's return value is the arrival index inside the bucket), a block scan turns the
/// histogram into bucket offsets, and only members of the same bucket are compared with each other (equal costs
/// are common -- symmetric cells -- so buckets hold a handful of keys).  Seven barriers, all LDS-only.
/// hist must be all zero on entry and is all zero again on exit.
__device__ __forceinline__ void rank_sort(uint64_t* skey, uint32_t* hist, uint32_t* wsum, uint32_t w, uint32_t range, int shiftD)
{
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int bitsRange = range > 1u ? 32 - __clz((int)(range - 1u)) : 0;
	const int shift = bitsRange > 11 ? bitsRange - 11 : 0; // (cost - L) >> shift < WF_NB
	uint64_t k[8];
	uint32_t meta[8];
#pragma unroll
	for (int u = 0; u < 8; u++) {
		const uint32_t i = (uint32_t)(tid + u * WF_T);
		k[u] = i < w ? skey[SK((int)i)] : ~0ull;
	}
	lds_barrier();
#pragma unroll
	for (int u = 0; u < 8; u++) {
		const uint32_t i = (uint32_t)(tid + u * WF_T);
		if (i < w) {
			const uint32_t bkt = (uint32_t)(k[u] >> shiftD) >> shift; // shiftD: position of (cost - L) in the key
			const uint32_t arr = atomicAdd(&hist[bkt], 1u);
			meta[u] = bkt | (arr << 11);
		}
	}
	lds_barrier();
	// exclusive scan of the histogram: 4 consecutive buckets per thread
	uint4 h = reinterpret_cast<uint4*>(hist)[tid];
	const uint32_t s0 = h.x, s1 = s0 + h.y, s2 = s1 + h.z, s3 = s2 + h.w;
	const uint32_t incl = wave_incl_add(s3);
	if (lane == 63)
		wsum[wave] = incl;
	lds_barrier();
	uint32_t base = incl - s3;
#pragma unroll
	for (int v = 0; v < WF_W; v++)
		base += v < wave ? wsum[v] : 0u;
	reinterpret_cast<uint4*>(hist)[tid] = make_uint4(base, base + s0, base + s1, base + s2);
	lds_barrier();
	bool multi = false;
#pragma unroll
	for (int u = 0; u < 8; u++) {
		const uint32_t i = (uint32_t)(tid + u * WF_T);
		if (i < w) {
			const uint32_t bkt = meta[u] & (WF_NB - 1), arr = meta[u] >> 11;
			const uint32_t st = hist[bkt], en = bkt + 1 < (uint32_t)WF_NB ? hist[bkt + 1] : w;
			skey[SK((int)(st + arr))] = k[u];
			meta[u] = st | ((en - st) << 12);
			multi = multi || en - st > 1u;
		}
	}
	lds_barrier();
	reinterpret_cast<uint4*>(hist)[tid] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
	for (int u = 0; u < 8; u++) {
		const uint32_t i = (uint32_t)(tid + u * WF_T);
		if (i < w) {
			const uint32_t st = meta[u] & 0xFFFu, size = meta[u] >> 12;
			uint32_t rank = st;
			if (size > 1u)
				for (uint32_t p = st; p < st + size; p++)
					rank += skey[SK((int)p)] < k[u];
			meta[u] = rank;
		}
	}
	lds_barrier();
#pragma unroll
	for (int u = 0; u < 8; u++) {
		const uint32_t i = (uint32_t)(tid + u * WF_T);
		if (i < w && multi) // a lane whose buckets are all singletons already sits at its rank
			skey[SK((int)meta[u])] = k[u];
	}
	lds_barrier();
}

/// rank_sort for windows of at most 4 * WF_T cells that ALSO fetches every window cell's 3 x 3 state block: the loads
/// are issued before the first sorting step and consumed after the last one, so the memory round trip (~10 k cycles,
/// the largest single item of a round) runs under the seven LDS steps of the sort.  smask[rank] receives the
/// candidate mask of the cell that ends up at `rank`.  Starts with a full barrier: the state bytes stored by the
/// previous round must be visible to the loads.
__device__ __forceinline__ void rank_sort_masks(uint64_t* skey, uint32_t* hist, uint32_t* wsum, uint8_t* smask, uint32_t w, uint32_t range, int shiftD,
	const uint8_t* __restrict__ state, int tpr, int cb, uint32_t cellMask)
{
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int bitsRange = range > 1u ? 32 - __clz((int)(range - 1u)) : 0;
	const int shift = bitsRange > 11 ? bitsRange - 11 : 0; // (cost - L) >> shift < WF_NB
	uint64_t k[4];
	uint32_t meta[4];
	constexpr int kAhead = 2; // elements per thread whose loads fly under the sort (more would spill the loaded words)
	NbhdRaw raw[kAhead];
#pragma unroll
	for (int u = 0; u < 4; u++) {
		if ((uint32_t)(u * WF_T) >= w)
			break; // (uniform) nothing left for any thread
		const uint32_t i = (uint32_t)(tid + u * WF_T);
		k[u] = i < w ? skey[SK((int)i)] : ~0ull;
	}
	__syncthreads();
#pragma unroll
	for (int u = 0; u < kAhead; u++) {
		if ((uint32_t)(u * WF_T) >= w)
			break; // (uniform) nothing left for any thread
		const uint32_t i = (uint32_t)(tid + u * WF_T);
		if (i < w) {
			const uint32_t cell = unpack_cell((uint32_t)k[u] & cellMask, cb);
			raw[u] = nbhd_issue(state, tpr, (int)(cell >> 16), (int)(cell & 0xFFFFu));
		}
	}
#pragma unroll
	for (int u = 0; u < 4; u++) {
		if ((uint32_t)(u * WF_T) >= w)
			break; // (uniform) nothing left for any thread
		const uint32_t i = (uint32_t)(tid + u * WF_T);
		if (i < w) {
			const uint32_t bkt = (uint32_t)(k[u] >> shiftD) >> shift;
			const uint32_t arr = atomicAdd(&hist[bkt], 1u);
			meta[u] = bkt | (arr << 11);
		}
	}
	lds_barrier();
	uint4 h = reinterpret_cast<uint4*>(hist)[tid];
	const uint32_t s0 = h.x, s1 = s0 + h.y, s2 = s1 + h.z, s3 = s2 + h.w;
	const uint32_t incl = wave_incl_add(s3);
	if (lane == 63)
		wsum[wave] = incl;
	lds_barrier();
	uint32_t base = incl - s3;
#pragma unroll
	for (int v = 0; v < WF_W; v++)
		base += v < wave ? wsum[v] : 0u;
	reinterpret_cast<uint4*>(hist)[tid] = make_uint4(base, base + s0, base + s1, base + s2);
	lds_barrier();
	bool multi = false;
#pragma unroll
	for (int u = 0; u < 4; u++) {
		if ((uint32_t)(u * WF_T) >= w)
			break; // (uniform) nothing left for any thread
		const uint32_t i = (uint32_t)(tid + u * WF_T);
		if (i < w) {
			const uint32_t bkt = meta[u] & (WF_NB - 1), arr = meta[u] >> 11;
			const uint32_t st = hist[bkt], en = bkt + 1 < (uint32_t)WF_NB ? hist[bkt + 1] : w;
			skey[SK((int)(st + arr))] = k[u];
			meta[u] = st | ((en - st) << 12);
			multi = multi || en - st > 1u;
		}
	}
	lds_barrier();
	reinterpret_cast<uint4*>(hist)[tid] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
	for (int u = 0; u < 4; u++) {
		if ((uint32_t)(u * WF_T) >= w)
			break; // (uniform) nothing left for any thread
		const uint32_t i = (uint32_t)(tid + u * WF_T);
		if (i < w) {
			const uint32_t st = meta[u] & 0xFFFu, size = meta[u] >> 12;
			uint32_t rank = st;
			if (size > 1u)
				for (uint32_t p = st; p < st + size; p++)
					rank += skey[SK((int)p)] < k[u];
			meta[u] = rank;
		}
	}
	lds_barrier();
#pragma unroll
	for (int u = 0; u < 4; u++) {
		if ((uint32_t)(u * WF_T) >= w)
			break; // (uniform) nothing left for any thread
		const uint32_t i = (uint32_t)(tid + u * WF_T);
		if (i < w) {
			if (multi) // a lane whose buckets are all singletons already sits at its rank
				skey[SK((int)meta[u])] = k[u];
			Row3 up, mid, dn;
			if (u < kAhead) {
				nbhd_finish(raw[u], up, mid, dn); // first use of the loaded words: the wait for them sits here
			} else { // windows beyond kAhead * WF_T cells: the rest is fetched now
				const uint32_t cell = unpack_cell((uint32_t)k[u] & cellMask, cb);
				load_state_nbhd(state, tpr, (int)(cell >> 16), (int)(cell & 0xFFFFu), up, mid, dn);
			}
			smask[meta[u]] = (uint8_t)candidate_mask(up, mid, dn);
		}
	}
	lds_barrier();
}

#ifndef PP_WF_WAVES_PER_SIMD
#define PP_WF_WAVES_PER_SIMD 4 // 2 workgroups of 8 waves per CU: <= 128 VGPRs
#endif
template <bool kProfile>
__global__ void __launch_bounds__(WF_T, PP_WF_WAVES_PER_SIMD) k_wavefront(MapView m, int nGoals, const int32_t* __restrict__ goalCells, float* __restrict__ costOut,
	void* workspace, int64_t bytesPerSlot, uint32_t fcap, uint32_t gcap, int32_t* errorFlag, unsigned long long* __restrict__ prof, int* __restrict__ goalCounter,
	int tiledOut, const double* __restrict__ goalPoses, const double* __restrict__ orderStarts, int32_t* __restrict__ orderOut, int* __restrict__ doneCounter, float* __restrict__ orderKeys,
	pph::WavefrontPublish pub)
{
	unsigned long long ph[WP_COUNT];
	unsigned long long tl = 0;
#define WF_STAMP(i)                                \
	if (kProfile) {                                \
		const unsigned long long now_ = clock64(); \
		ph[i] += now_ - tl;                        \
		tl = now_;                                 \
	}
	__shared__ uint64_t skey[kSkewed(WF_LCAP)]; // window keys (skewed layout), then (as two uint32 arrays) the claim hash table
	__shared__ uint64_t lent[WF_LIST];           // open list in LDS, in push order: cost bits << 32 | padded cell
	__shared__ __attribute__((aligned(16))) uint32_t hist[WF_NB];
	__shared__ uint8_t smask[4 * WF_T];           // candidate mask per window rank (rank_sort_masks)
	__shared__ uint32_t s_wcnt[64];              // ordered compaction: counts per (chunk row u, wave)
	__shared__ uint64_t s_wtot[WF_W];
	__shared__ uint32_t s_wsum[WF_W];
	__shared__ uint32_t s_minNext[2], s_packFail, s_cand;
	__shared__ int s_goal, s_start;
	uint32_t* const hcell = reinterpret_cast<uint32_t*>(skey);          // [WF_HCAP] padded cell index + 1, 0 = empty
	uint32_t* const hkey = reinterpret_cast<uint32_t*>(skey) + WF_HCAP; // [WF_HCAP] min (i*8+j)

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const unsigned long long ltMask = (1ull << lane) - 1ull;
	const int cols = m.cols, rows = m.rows;
	const int pc = cols + 2;             // row stride of the (row-major) tag grid
	const int tpr = state_tiles_per_row(cols);
	const int64_t cells = (int64_t)rows * cols;
	const int64_t pcells = padded_cells(rows, cols);
	const int64_t stBytes = state_bytes(rows, cols);
	const int64_t fieldElems = tiledOut ? (int64_t)field_tiled_elems(rows, cols) : cells; // floats per goal in costOut
	// packed window key = (cost - L) | ~position : pb | (row, col) : 2 cb; pb = 12 while the list is in LDS, else 19
	const bool packable = rows + 2 <= 8191 && cols + 2 <= 8191;
	const int cb = (rows + 2 <= 2047 && cols + 2 <= 2047) ? 11 : 13;
	const uint32_t cellMask = (1u << (2 * cb)) - 1u;
	WfSlot S = slot_view(workspace, bytesPerSlot, blockIdx.x, stBytes, pcells, fcap, gcap);
	uint8_t* const state = S.state;
	// the two halves of the HBM list as plain pointers: `S.fent[cur]` with a run-time index keeps the whole struct in scratch
	// memory (one scratch load per use in the round loop)
	uint64_t* const fent0 = S.fent[0];
	uint64_t* const fent1 = S.fent[1];
	const float kDiag = sqrtf(2.0f); // std::sqrt(2.0f), heuristics.cpp:134
	const int nbOff[8] = { -1, -65536 - 1, 65536 - 1, 1, -65536 + 1, 65536 + 1, -65536, 65536 }; // (row << 16 | col) offsets of kDr/kDc
	// index of map cell (r, c) in this goal's output field
	auto out_index = [&](int r, int c) -> size_t { return tiledOut ? field_tiled_index(cols, r, c) : (size_t)r * cols + c; };
	auto tag_index = [&](uint32_t cell) -> uint32_t { return (cell >> 16) * (uint32_t)pc + (cell & 0xFFFFu); };

	for (int i = tid; i < WF_NB; i += WF_T)
		hist[i] = 0u;
	int tagGoal = -1; // goal whose fallback rounds the tag grid currently describes (it is cleared lazily)
	// goals are handed out dynamically: a workgroup that finishes early takes the next one (balanced tail)
	int pendingSlot = -1; // pipeline use: field slot of the goal this workgroup has just finished, not yet announced
	bool ringTurn = true; // (thread 0) pipeline use: does the urgent ring have the first look at the next hand-out?
	for (;;) {
		__syncthreads(); // (every wave's stores of the previous goal have completed: the barrier's release waits for them)
		if (tid == 0) {
			if (pendingSlot >= 0) {
				// the finished field is handed to the search grid (pp_pipeline.hpp): this XCD's dirty lines written back, then the
				// slot number appended to the ready ring under the stamp of its position (consumers take an entry only when its stamp
				// matches, so entries may land out of order)
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				const unsigned long long t = __hip_atomic_fetch_add(pub.readyTail, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(pub.ready + (t & pub.readyMask), ((t + 1ull) << 32) | (unsigned long long)(uint32_t)pendingSlot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			// ---- the next goal: an entry of the pipeline's urgent ring if there is one (whatever launch it came with), else the next of this
			// launch's own list.  Pipeline entries are (slot | generation << 20) and every slot has a claim word, `generation << 1` when the slot
			// was filled: whoever moves it to `generation << 1 | 1` builds the field.  A slot sits in its launch's list AND, if urgent, in the
			// ring, so it is built once; and an entry that has outlived its query -- an urgent slot may be built by an older launch, searched,
			// polled and refilled before its own launch reaches its list entry -- carries a generation that no longer matches and is skipped
			// (without the generation such an entry claims the refilled slot: a goal written AFTER this launch began, reached through the
			// list path; a version that did so, with the pose read through the caches, built ~1 field in 4096 for the slot's previous goal).
			// The ring has the first look on every other hand-out only: goals of the lists keep moving whatever share of the queries is urgent
			// (with the ring always first, a third of the queries urgent cost 12 % of the throughput and half of them 36 %: the later
			// submissions' urgent goals kept overtaking the earlier submissions' ordinary ones, whose slots then stayed taken).
			auto take_urgent = [&]() -> int {
				for (;;) {
					unsigned long long h = __hip_atomic_load(pub.urgentHead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					const unsigned long long e = __hip_atomic_load(pub.urgent + (h & pub.urgentMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if ((uint32_t)(e >> 32) != (uint32_t)(h + 1ull))
						return -1; // nothing there (or reserved and not yet written: its own launch will come to it)
					if (!__hip_atomic_compare_exchange_strong(pub.urgentHead, &h, h + 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
						continue; // another workgroup took it: look again
					const int slot = (int)((uint32_t)e & pph::kSlotMask);
					int expect = (int)(((uint32_t)e >> pph::kSlotBits) << 1);
					if (__hip_atomic_compare_exchange_strong(pub.claimed + slot, &expect, expect | 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
						return slot;
				}
			};
			int next = -1;
			if (pub.urgent && ringTurn)
				next = take_urgent();
			while (next < 0) {
				const int gi = atomicAdd(goalCounter, 1);
				if (gi >= (pub.nGoalsDev ? *pub.nGoalsDev : nGoals)) // (a count the tile form's launch in front of this one wrote: pp_wavefront_tiles.hip)
					break;
				if (!pub.slotList) {
					next = gi;
				} else {
					const uint32_t e = (uint32_t)pub.slotList[gi];
					const int slot = (int)(e & pph::kSlotMask);
					int expect = (int)((e >> pph::kSlotBits) << 1);
					if (!pub.claimed || __hip_atomic_compare_exchange_strong(pub.claimed + slot, &expect, expect | 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
						next = slot;
				}
			}
			// (A workgroup leaves when its launch's list has run out, whatever the ring holds: by then every slot of its own submission is
			// claimed -- the list names them all -- and what is left in the ring belongs to later submissions, whose launches follow.  Workgroups
			// that stayed for the ring kept their launch alive with ever fewer of them, and the stream's next launch waiting behind it.)
			ringTurn = !ringTurn;
			s_goal = next;
			// the goal's cell.  In the pipeline the pose may have been written while this launch was running (urgent ring): it is read with
			// agent-scope loads, which do not look at this XCD's possibly stale copy of the line -- cheaper than an acquire fence per goal,
			// which would drop the XCD's whole L2 contents under the search rows that share it.
			if (next >= 0) {
				int32_t st;
				if (goalPoses) { // (x, y, theta) triples: WorldPositionToGridCell(bounded), heuristics.cpp:115
					double px, py;
					if (pub.claimed || pub.agentPoseLoads) {
						px = __hip_atomic_load(goalPoses + 3 * (size_t)next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						py = __hip_atomic_load(goalPoses + 3 * (size_t)next + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					} else {
						px = goalPoses[3 * (size_t)next];
						py = goalPoses[3 * (size_t)next + 1];
					}
					int row, col;
					world_to_cell(m, px, py, row, col);
					st = inside_map(m, row, col) ? row * m.cols + col : -1;
				} else {
					st = goalCells[next];
				}
				s_start = st;
			}
		}
		pendingSlot = -1;
		__syncthreads();
		// g indexes the goal poses and the output fields: the launch's own numbering, or (pipeline) a field slot
		const int g = s_goal;
		if (g < 0)
			break;
		if (pub.ready)
			pendingSlot = g;
		if (kProfile) {
			for (int i = 0; i < WP_COUNT; i++)
				ph[i] = 0;
			tl = clock64();
		}
		float* cost = costOut + (int64_t)g * fieldElems;
		const int32_t start = s_start;
		// ---- every cell starts at +inf / unexplored (heuristics.cpp:108-113).  The field is NOT filled up front: every cell the wavefront
		// discovers is written exactly once, with its cost, and the cells it never reaches (occupied ones, enclosed pockets) get their +inf
		// in one pass over the state bytes when the goal is done -- 4 MB of stores per goal less at 1024^2, and the lines of the field
		// are dirtied once instead of twice.  Only a goal outside the map (nothing is discovered) fills the whole field.
		if (start < 0) {
			uint4* c4 = reinterpret_cast<uint4*>(cost); // fieldElems * 4 bytes is a multiple of 16 when tiled; row-major: tail below
			const int64_t n4 = fieldElems >> 2;
#if PP_WF_NT_FILL
			typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
			const u32x4 inf4 = { kInfBits, kInfBits, kInfBits, kInfBits };
			for (int64_t i = tid; i < n4; i += WF_T)
				__builtin_nontemporal_store(inf4, reinterpret_cast<u32x4*>(cost) + i);
#else
			for (int64_t i = tid; i < n4; i += WF_T)
				c4[i] = make_uint4(kInfBits, kInfBits, kInfBits, kInfBits);
#endif
			for (int64_t i = (n4 << 2) + tid; i < fieldElems; i += WF_T)
				reinterpret_cast<uint32_t*>(cost)[i] = kInfBits;
			continue; // goal outside the map (heuristics.cpp:115-117): the field stays +inf
		}
		// one tile row (8 bytes) per thread and step; everything outside the map is "occupied"
		for (int64_t t = tid; t < (stBytes >> 3); t += WF_T) {
			const int tile = (int)(t >> 3), trow = (int)(t & 7);
			const int tr = tile / tpr, tc = tile - tr * tpr;
			const int pr = (tr << 3) + trow, r = pr - 1;
			unsigned long long v = 0;
#pragma unroll
			for (int x = 0; x < 8; x++) {
				const int c = (tc << 3) + x - 1;
				uint32_t st = ST_OCC;
				if (r >= 0 && r < rows && c >= 0 && c < cols)
					st = m.occ8[(int64_t)r * cols + c] ? ST_OCC : ST_FREE;
				v |= (unsigned long long)st << (8 * x);
			}
			reinterpret_cast<unsigned long long*>(state)[t] = v;
		}
		if (tid == 0) {
			s_minNext[0] = 0xFFFFFFFFu;
			s_minNext[1] = 0xFFFFFFFFu;
			s_packFail = 0;
			s_cand = 0;
		}
		__syncthreads();
		bool inLds = packable; // where the open list lives
		int cur = 0;           // ping-pong index of the HBM list
		if (tid == 0) {
			const int sr = start / cols, sc = start - sr * cols;
			const uint32_t sp = ((uint32_t)(sr + 1) << 16) | (uint32_t)(sc + 1);
			state[st_addr(tpr, sr + 1, sc + 1)] = (uint8_t)ST_SEEN; // the reference pushes the goal cell even when it is occupied
			cost[out_index(sr, sc)] = 0.0f;
			if (inLds)
				lent[0] = (uint64_t)sp; // cost 0
			else
				fent0[0] = (uint64_t)sp;
		}
		__syncthreads();

		WF_STAMP(WP_INIT);
		uint32_t n = 1; // open-list size
		uint32_t round = 0;
		uint32_t deferFrom = 0xFFFFFFFFu; // first open-list entry whose state / cost stores are still owed (see the top of the round loop)
		uint32_t lBits = 0u; // smallest cost in the open list
		bool overflow = false;

		while (n > 0) {
#if PP_WF_DEFER_STORES
			// The cells the previous round discovered sit at the end of the (LDS) list, cost and cell in every entry; their "discovered"
			// byte and their cost are stored HERE, one entry per lane, instead of inside that round's push loop, where every lane walked
			// its own wins (a wave iterated as often as its busiest lane had wins, two scattered stores per pass).  Same bytes to the same
			// addresses, before anything reads them: the next reader of the state grid is this round's neighbourhood pass, behind a full
			// barrier; the field is not read by this kernel at all.
			if (deferFrom != 0xFFFFFFFFu) {
				for (uint32_t i = deferFrom + (uint32_t)tid; i < n; i += WF_T) {
					const uint64_t e = lent[i];
					const uint32_t ncell = (uint32_t)e;
					const int nr = (int)(ncell >> 16), nc = (int)(ncell & 0xFFFFu);
					state[st_addr(tpr, nr, nc)] = (uint8_t)ST_SEEN;
					store_cost(&cost[out_index(nr - 1, nc - 1)], __uint_as_float((uint32_t)(e >> 32)));
				}
				deferFrom = 0xFFFFFFFFu;
			}
#endif
			uint64_t* const fentCur = cur ? fent1 : fent0;
			uint64_t* const fentNxt = cur ? fent0 : fent1;
			const int par = (int)(round & 1u);
			const float L = __uint_as_float(lBits);
			const uint32_t hiBits = __float_as_uint(L + 1.0f);
			const int pb = inLds ? 12 : 19;
			const int shiftD = pb + 2 * cb; // (cost - L) has 64 - shiftD bits: >= 23 unless cb = 13 in HBM mode (19)
			const uint32_t posMask = (1u << pb) - 1u;
			WF_STAMP(WP_MIN);
			// ---- partition: window (cost < fl(L+1)) -> sort buffer; the rest stays in the open list IN ORDER.
			// The list is kept in push order, so "pushed later" == "further back": a window entry is packed as
			// (cost - L : 23 | ~position : 19 | row, col : 22) and sorting those keys yields the reference's pop order
			// (cost ascending, most recent push first).  Slots come from ballots + one 64-entry scan per chunk.
			bool fast = true;
			uint32_t w = 0, b = 0;
			uint32_t restMin = 0xFFFFFFFFu;
			if (__builtin_expect(inLds, 1)) {
				uint64_t e[8];
				unsigned long long bw[8], br[8];
#pragma unroll
				for (int u = 0; u < 8; u++) {
					const uint32_t i = (uint32_t)(u * WF_T + tid);
					if ((uint32_t)(u * WF_T) >= n) { // (uniform) the list ends before this row of entries
						e[u] = ~0ull;
						bw[u] = br[u] = 0ull;
						if (lane == 0)
							s_wcnt[u * WF_W + wave] = 0u;
						continue;
					}
					e[u] = i < n ? lent[i] : ~0ull;
					const bool inW = i < n && (uint32_t)(e[u] >> 32) < hiBits;
					bw[u] = __ballot(inW);
					br[u] = __ballot(i < n && !inW);
					if (lane == 0)
						s_wcnt[u * WF_W + wave] = ((uint32_t)__popcll(bw[u]) << 16) | (uint32_t)__popcll(br[u]);
				}
				lds_barrier(); // counts visible; every entry read before the list is compacted in place
				const uint32_t cnt = s_wcnt[lane];
				const uint32_t incl = wave_incl_add(cnt);
				const uint32_t excl = incl - cnt;
				const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
				w = total >> 16;
				b = total & 0xFFFFu;
#pragma unroll
				for (int u = 0; u < 8; u++) {
					if ((uint32_t)(u * WF_T) >= n)
						break; // (uniform)
					const uint32_t i = (uint32_t)(u * WF_T + tid);
					const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)excl, u * WF_W + wave);
					const uint32_t c = (uint32_t)(e[u] >> 32), cell = (uint32_t)e[u];
					const bool inW = i < n && c < hiBits;
					if (inW) {
						const uint32_t wslot = (off >> 16) + (uint32_t)__popcll(bw[u] & ltMask);
						skey[SK((int)wslot)] = ((uint64_t)(c - lBits) << shiftD) | ((uint64_t)(posMask - i) << (2 * cb)) | (uint64_t)pack_cell(cell, cb);
					} else if (i < n) {
						const uint32_t rslot = (off & 0xFFFFu) + (uint32_t)__popcll(br[u] & ltMask);
						lent[rslot] = e[u];
						restMin = min(restMin, c);
					}
				}
			} else {
				for (int attempt = 0; attempt < 2; attempt++) {
					uint32_t wRun = 0, bRun = 0;
					restMin = 0xFFFFFFFFu;
					for (uint32_t i0 = 0; i0 < n; i0 += 8 * WF_T) {
						uint64_t e[8];
						unsigned long long bw[8], br[8];
#pragma unroll
						for (int u = 0; u < 8; u++) {
							const uint32_t i = i0 + (uint32_t)(u * WF_T + tid);
							e[u] = i < n ? fentCur[i] : ~0ull;
							const bool inW = i < n && (uint32_t)(e[u] >> 32) < hiBits;
							bw[u] = __ballot(inW);
							br[u] = __ballot(i < n && !inW);
							if (lane == 0)
								s_wcnt[u * WF_W + wave] = ((uint32_t)__popcll(bw[u]) << 16) | (uint32_t)__popcll(br[u]);
						}
						lds_barrier();
						const uint32_t cnt = s_wcnt[lane];
						const uint32_t incl = wave_incl_add(cnt);
						const uint32_t excl = incl - cnt;
						const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
#pragma unroll
						for (int u = 0; u < 8; u++) {
							const uint32_t i = i0 + (uint32_t)(u * WF_T + tid);
							const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)excl, u * WF_W + wave);
							const uint32_t c = (uint32_t)(e[u] >> 32), cell = (uint32_t)e[u];
							const bool inW = i < n && c < hiBits;
							if (inW) {
								const uint32_t wslot = wRun + (off >> 16) + (uint32_t)__popcll(bw[u] & ltMask);
								if (fast) {
									if (!packable || i > posMask || (uint64_t)(c - lBits) >= (1ull << (64 - shiftD)))
										s_packFail = 1;
									if (wslot < (uint32_t)WF_LCAP)
										skey[SK((int)wslot)] = ((uint64_t)(c - lBits) << shiftD) | ((uint64_t)(posMask - (i & posMask)) << (2 * cb)) | (uint64_t)(pack_cell(cell, cb) & cellMask);
								} else if (wslot < S.gcap) {
									S.gkeys[wslot] = ((uint64_t)c << 32) | (uint64_t)(0xFFFFFFFFu - i);
									S.gvals[wslot] = cell;
								}
							} else if (i < n) {
								const uint32_t rslot = bRun + (off & 0xFFFFu) + (uint32_t)__popcll(br[u] & ltMask);
								fentNxt[rslot] = e[u]; // rslot < n <= fcap
								restMin = min(restMin, c);
							}
						}
						wRun += total >> 16;
						bRun += total & 0xFFFFu;
						lds_barrier(); // s_wcnt is rewritten by the next chunk
					}
					w = wRun;
					b = bRun;
					lds_barrier();
					const bool ok = !s_packFail && w <= (uint32_t)WF_LCAP;
					if (!fast || ok)
						break;
					fast = false; // redo into the unpacked HBM buffers (the source list S.fent[cur] is untouched)
				}
			}
			restMin = wave_min(restMin);
			if (lane == 0 && restMin != 0xFFFFFFFFu)
				atomicMin(&s_minNext[par], restMin);
			if (tid == 0)
				s_cand = 0;
			lds_barrier();
			WF_STAMP(WP_PART);
			if (w > S.gcap || round + 1u >= (1u << 15) || w > (1u << 17)) {
				overflow = true;
				break;
			}
			const bool prefetched = fast && w > 1 && w <= 4u * WF_T; // masks come out of the sort
			if (__builtin_expect(fast, 1)) {
				if (__builtin_expect(prefetched, 1))
					rank_sort_masks(skey, hist, s_wsum, smask, w, hiBits - lBits, shiftD, state, tpr, cb, cellMask);
				else if (w > 1)
					rank_sort(skey, hist, s_wsum, w, hiBits - lBits, shiftD);
			} else {
				const uint32_t P = w <= 1 ? 2 : (1u << (32 - __clz((int)(w - 1))));
				__syncthreads();
				for (uint32_t i = w + tid; i < P; i += WF_T)
					S.gkeys[i] = ~0ull;
				__syncthreads();
				bitonic_sort<false, true>(S.gkeys, S.gvals, (int)P);
			}
			WF_STAMP(WP_SORT);
			if (kProfile) {
				ph[WP_ROUNDS]++;
				ph[WP_SUMW] += w;
				ph[WP_SUMP] += inLds ? 0 : 1; // rounds whose open list lives in HBM
			}
			uint32_t newMin = 0xFFFFFFFFu;
			uint32_t newCount = 0;
			// Where do this round's pushes go?  They stay in LDS while survivors + pushes fit (decided once the
			// number of discovered cells is known or bounded).
			bool pushLds = false;
			auto push_entry = [&](uint32_t slot, uint32_t ncell, uint32_t pb) {
				if (pushLds)
					lent[slot] = ((uint64_t)pb << 32) | ncell;
				else if (slot < S.fcap)
					fentNxt[slot] = ((uint64_t)pb << 32) | ncell;
			};
			// the list leaves LDS before the pushes when they might not fit: survivors are copied to HBM once
			auto spill_list = [&]() {
				for (uint32_t i = tid; i < b; i += WF_T)
					fentNxt[i] = lent[i];
			};
			uint32_t myCell[4] = { 0, 0, 0, 0 }, myCost[4] = { 0, 0, 0, 0 }, myMask[4] = { 0, 0, 0, 0 };
			bool hashed = false;
			if (__builtin_expect(fast && w <= 4u * WF_T, 1)) {
#pragma unroll
				for (int q = 0; q < 4; q++) {
					const uint32_t i = tid + q * WF_T;
					if (i < w) {
						const uint64_t k = skey[SK((int)i)];
						myCell[q] = unpack_cell((uint32_t)k & cellMask, cb);
						myCost[q] = lBits + (uint32_t)(k >> shiftD);
					}
				}
				unsigned long long ts_ = 0;
				uint32_t cnt = 0;
				if (__builtin_expect(prefetched, 1)) {
					if (kProfile) {
						ts_ = clock64();
						ph[WP_O_WAIT] += ts_ - tl;
					}
#pragma unroll
					for (int q = 0; q < 4; q++) {
						const uint32_t i = tid + q * WF_T;
						if (i < w) {
							myMask[q] = smask[i];
							cnt += __popc(myMask[q]);
						}
					}
				} else {
					__syncthreads(); // the state bytes stored by the previous round are visible from here on
					if (kProfile) {
						ts_ = clock64();
						ph[WP_O_WAIT] += ts_ - tl;
					}
					// (rare: single-cell windows and windows beyond 4 x 512 cells never come here with more than one q busy, so
					// the loads are not batched -- holding 4 x 9 state words live would cost the round loop its registers)
#pragma unroll
					for (int q = 0; q < 4; q++) {
						const uint32_t i = tid + q * WF_T;
						if (i < w) {
							Row3 up, mid, dn;
							load_state_nbhd(state, tpr, (int)(myCell[q] >> 16), (int)(myCell[q] & 0xFFFFu), up, mid, dn);
							myMask[q] = candidate_mask(up, mid, dn);
							cnt += __popc(myMask[q]);
						}
					}
				}
				if (kProfile) {
					const unsigned long long now_ = clock64();
					ph[WP_O_LOAD] += now_ - ts_;
					ts_ = now_;
				}
				cnt = wave_incl_add(cnt);
				if (lane == 63 && cnt)
					atomicAdd(&s_cand, cnt);
				lds_barrier();
				if (kProfile)
					ph[WP_O_COUNT] += clock64() - ts_;
				// The claim table has WF_HCAP slots; the round may use it only when every candidate (counted with
				// duplicates, so an upper bound on distinct cells) fits with room to spare: insertion then always ends.
				hashed = s_cand <= (uint32_t)(WF_HCAP * 3 / 4);
			} else {
				__syncthreads();
			}
			if (__builtin_expect(hashed, 1)) {
				// ================= fast path: claims in an LDS hash table (the sort buffer is reused) =================
				for (int i = tid; i < WF_HCAP / 4; i += WF_T) {
					reinterpret_cast<uint4*>(hcell)[i] = make_uint4(0u, 0u, 0u, 0u);
					reinterpret_cast<uint4*>(hkey)[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
				}
				lds_barrier();
				// candidates of this lane as one bit set (bit q*8+j); lanes walk their own set bits, so the wave iterates
				// max-popcount times (about a dozen) instead of over all 32 (q, j) combinations
				const uint32_t candBits = myMask[0] | (myMask[1] << 8) | (myMask[2] << 16) | (myMask[3] << 24);
				for (uint32_t rem = candBits; rem;) {
					const int bit = __ffs((int)rem) - 1;
					rem &= rem - 1u;
					const int q = bit >> 3, j = bit & 7;
					const uint32_t pcell = q == 0 ? myCell[0] : q == 1 ? myCell[1] : q == 2 ? myCell[2] : myCell[3];
					const uint32_t ncell = pcell + (uint32_t)(dir_dr(j) * 65536 + dir_dc(j));
					const uint32_t key = ((uint32_t)tid + (uint32_t)q * WF_T) * 8u + (uint32_t)j;
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
				lds_barrier();
				if (kProfile)
					ph[WP_O_INSERT] += clock64() - tl;
				WF_STAMP(WP_OFFER);
				// the winners: a lane first finds its wins (bit q*8+j); a block scan then hands out open-list slots in
				// the reference's push order, i.e. ascending (i, j) = ascending (q, thread, j)
				uint32_t winBits = 0;
				for (uint32_t rem = candBits; rem;) {
					const int bit = __ffs((int)rem) - 1;
					rem &= rem - 1u;
					const int q = bit >> 3, j = bit & 7;
					const uint32_t pcell = q == 0 ? myCell[0] : q == 1 ? myCell[1] : q == 2 ? myCell[2] : myCell[3];
					const uint32_t ncell = pcell + (uint32_t)(dir_dr(j) * 65536 + dir_dc(j));
					const uint32_t key = ((uint32_t)tid + (uint32_t)q * WF_T) * 8u + (uint32_t)j;
					uint32_t h = (ncell * 2654435761u) >> (32 - 12);
					while (hcell[h] != ncell + 1u)
						h = (h + 1) & (WF_HCAP - 1);
					if (hkey[h] == key)
						winBits |= 1u << bit;
				}
				if (kProfile)
					ph[WP_P_LOOKUP] += clock64() - tl;
				// per-q counts packed in 16-bit fields (a field never exceeds 4096)
				const uint32_t c01 = (uint32_t)__popc(winBits & 0xFFu) | ((uint32_t)__popc(winBits & 0xFF00u) << 16);
				const uint32_t c23 = (uint32_t)__popc(winBits & 0xFF0000u) | ((uint32_t)__popc(winBits & 0xFF000000u) << 16);
				const uint32_t i01 = wave_incl_add(c01), i23 = wave_incl_add(c23);
				if (lane == 63)
					s_wtot[wave] = ((uint64_t)i23 << 32) | i01;
				lds_barrier();
				uint64_t all = 0, pre = 0;
#pragma unroll
				for (int v = 0; v < WF_W; v++) {
					const uint64_t t = s_wtot[v];
					all += t;
					pre += v < wave ? t : 0ull;
				}
				const uint64_t exclT = pre + (((uint64_t)(i23 - c23) << 32) | (uint64_t)(i01 - c01)); // per-q exclusive prefix of this lane
				uint32_t qBase[4];
				uint32_t run = 0;
#pragma unroll
				for (int q = 0; q < 4; q++) {
					qBase[q] = b + run + (uint32_t)((exclT >> (16 * q)) & 0xFFFFu);
					run += (uint32_t)((all >> (16 * q)) & 0xFFFFu);
				}
				newCount = run;
				if (kProfile)
					ph[WP_P_SCAN] += clock64() - tl;
				pushLds = inLds && b + newCount <= (uint32_t)WF_LIST;
				if (inLds && !pushLds)
					spill_list();
				// each lane walks its own wins (a handful per row q), lowest bit first = ascending j
#pragma unroll
				for (int q = 0; q < 4; q++) {
					uint32_t slot = qBase[q];
					uint32_t rem = (winBits >> (q * 8)) & 0xFFu;
#pragma unroll
					for (int it = 0; it < 8; it++) { // at most 8 wins per row; the trip count is the lane's own
						if (!rem)
							break;
						const int j = __ffs((int)rem) - 1;
						rem &= rem - 1u;
						const int dr = dir_dr(j), dc = dir_dc(j);
						const uint32_t ncell = myCell[q] + (uint32_t)(dr * 65536 + dc);
						const float transitionCost = (dr == 0 || dc == 0) ? 1.0f : kDiag;
						const float pathCost = transitionCost + __uint_as_float(myCost[q]); // heuristics.cpp:135
						const uint32_t pb = __float_as_uint(pathCost);
#if PP_WF_DEFER_STORES
						if (!pushLds) // (entries that stay in LDS are stored from there at the start of the next round)
#endif
						{
							const int nr = (int)(ncell >> 16), nc = (int)(ncell & 0xFFFFu);
							state[st_addr(tpr, nr, nc)] = (uint8_t)ST_SEEN;
							store_cost(&cost[out_index(nr - 1, nc - 1)], pathCost);
						}
						newMin = min(newMin, pb);
						push_entry(slot++, ncell, pb);
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
					tagGoal = g;
				}
				__syncthreads();
				const uint32_t roundTag = (round + 1u) << 17;
				for (uint32_t i = tid; i < w; i += WF_T) {
					const uint32_t cell = fast ? unpack_cell((uint32_t)skey[SK((int)i)] & cellMask, cb) : S.gvals[i];
					S.tag[tag_index(cell)] = roundTag | i;
				}
				__syncthreads();
				WF_STAMP(WP_OFFER);
				uint32_t runBase = b;
				for (uint32_t i0 = 0; i0 < w; i0 += WF_T) {
					const uint32_t i = i0 + tid;
					uint32_t cell = 0, cbits = 0, winMask = 0;
					int pr = 0, pcc = 0;
					if (i < w) {
						if (fast) {
							const uint64_t k = skey[SK((int)i)];
							cell = unpack_cell((uint32_t)k & cellMask, cb);
							cbits = lBits + (uint32_t)(k >> shiftD);
						} else {
							cell = S.gvals[i];
							cbits = (uint32_t)(S.gkeys[i] >> 32);
						}
						pr = (int)(cell >> 16);
						pcc = (int)(cell & 0xFFFFu);
						Row3 up, mid, dn;
						load_state_nbhd(state, tpr, pr, pcc, up, mid, dn);
						const uint32_t mk = candidate_mask(up, mid, dn);
#pragma unroll
						for (int j = 0; j < 8; j++) {
							if (!(mk & (1u << j)))
								continue;
							// n = neighbour j.  Is another window cell the first to reach n?
							const uint32_t ncell = cell + (uint32_t)nbOff[j];
							const int nr = pr + kDr[j], nc = pcc + kDc[j];
							const uint32_t mine = i * 8u + (uint32_t)j;
							const uint32_t tn = tag_index(ncell);
							const Row3 tu = load_row3(S.tag + tn - pc - 1), tm = load_row3(S.tag + tn - 1), td = load_row3(S.tag + tn + pc - 1);
							Row3 gu, gm, gd;
							load_state_nbhd(state, tpr, nr, nc, gu, gm, gd);
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
							if (win)
								winMask |= 1u << j;
						}
					}
					// ordered slots for this chunk of 512 window cells (ascending i, then j)
					const uint32_t cw = (uint32_t)__popc(winMask);
					const uint32_t inclW = wave_incl_add(cw);
					if (lane == 63)
						s_wsum[wave] = inclW;
					__syncthreads(); // also: every state/tag read of this chunk precedes the state stores below
					uint32_t slot = runBase + inclW - cw, chunkTotal = 0;
#pragma unroll
					for (int v = 0; v < WF_W; v++) {
						const uint32_t t = s_wsum[v];
						chunkTotal += t;
						slot += v < wave ? t : 0u;
					}
					if (winMask) {
						const float ci = __uint_as_float(cbits);
#pragma unroll
						for (int j = 0; j < 8; j++) {
							if (!(winMask & (1u << j)))
								continue;
							const uint32_t ncell = cell + (uint32_t)nbOff[j];
							const float transitionCost = (kDr[j] == 0 || kDc[j] == 0) ? 1.0f : kDiag;
							const float pathCost = transitionCost + ci; // heuristics.cpp:135
							const uint32_t pb = __float_as_uint(pathCost);
							state[st_addr(tpr, pr + kDr[j], pcc + kDc[j])] = (uint8_t)ST_SEEN;
							store_cost(&cost[out_index(pr - 1 + kDr[j], pcc - 1 + kDc[j])], pathCost);
							newMin = min(newMin, pb);
							push_entry(slot++, ncell, pb);
						}
					}
					runBase += chunkTotal;
					__syncthreads(); // s_wsum is rewritten by the next chunk; its candidate masks must see these state stores
				}
				newCount = runBase - b;
			}
#if PP_WF_DEFER_STORES
			if (hashed && pushLds && newCount > 0)
				deferFrom = b;
#endif
			if (kProfile && hashed)
				ph[WP_P_STORE] += clock64() - tl;
			newMin = wave_min(newMin);
			if (lane == 0 && newMin != 0xFFFFFFFFu)
				atomicMin(&s_minNext[par], newMin);
			if (kProfile && !hashed) {
				ph[WP_FBROUNDS]++;
				ph[WP_FBCYC] += clock64() - tl;
			}
			const uint32_t nn = b + newCount;
			// where the list lives next round
			bool nextLds = inLds && pushLds;
			if (!nextLds)
				__syncthreads(); // HBM list: stores of this round are read back by the next partition
			else
				lds_barrier();
			WF_STAMP(WP_PUSH);
			lBits = s_minNext[par];
			if (tid == 0) {
				s_minNext[par ^ 1] = 0xFFFFFFFFu; // accumulates during the next round; nobody reads it before that round ends
				s_packFail = 0;
			}
			if (nn > S.fcap) {
				overflow = true;
				break;
			}
			if (!nextLds) {
				cur ^= 1;
				if (packable && nn <= (uint32_t)WF_LIST / 2) { // small again: bring it back into LDS
					for (uint32_t i = tid; i < nn; i += WF_T)
						lent[i] = fentNxt[i]; // (the list just written)
					lds_barrier();
					nextLds = true;
				}
			}
			inLds = nextLds;
			round++;
			n = nn;
			WF_STAMP(WP_TAIL);
		}
		if (overflow && tid == 0)
			*errorFlag = 1; // open list / round count beyond the workspace encoding
		__syncthreads(); // (the last round's state stores are visible)
		// ---- +inf for every map cell the wavefront never discovered (see the note at the top of the goal)
		// (arguments recomputed from the kernel arguments: nothing extra stays live across the round loop for this call)
		fill_unreached(S.state, state_bytes(m.rows, m.cols) >> 3, state_tiles_per_row(m.cols), m.rows, m.cols, tiledOut, costOut + (int64_t)g * fieldElems, (int)threadIdx.x);
		__syncthreads();
		if (orderKeys && orderStarts && tid == 0) {
			// hand-out key of this query: the field value at its start pose, published with a device-scope store (the
			// workgroup that sorts the keys may sit on another XCD, whose L2 does not see this one's plain stores)
			int row, col;
			world_to_cell(m, orderStarts[3 * g], orderStarts[3 * g + 1], row, col);
			float c = __builtin_huge_valf();
			if (inside_map(m, row, col))
				c = cost[tiledOut ? field_tiled_index(cols, row, col) : (size_t)row * cols + col];
			__hip_atomic_store(orderKeys + g, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (kProfile && tid == 0)
			for (int i = 0; i < WP_COUNT; i++)
				prof[(size_t)g * WP_COUNT + i] = ph[i];
	}
#undef WF_STAMP
	if (pub.exitCounter && tid == 0) {
		// (this workgroup's last look at the goal counter precedes its exit count)
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		if (__hip_atomic_fetch_add(pub.exitCounter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
			__hip_atomic_store(goalCounter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(pub.exitCounter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (pub.resetOnExit)
				__hip_atomic_store(pub.resetOnExit, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	// ---- optional epilogue for the planner: the LAST workgroup to run out of goals orders the queries by decreasing
	// field value at their start pose (probable longest search first; +inf = unreachable first) for the search kernel's
	// hand-out -- no extra launch that would queue behind persistent grids.  nGoals <= WF_LCAP.
	if (orderOut) {
		__syncthreads();
		if (tid == 0) {
			__builtin_amdgcn_s_waitcnt(0); // this workgroup's key stores have reached the coherence point
			s_goal = __hip_atomic_fetch_add(doneCounter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1 ? 1 : 0;
		}
		__syncthreads();
		if (s_goal) {
			int P = 1;
			while (P < nGoals)
				P <<= 1;
			for (int i = tid; i < P; i += WF_T) {
				unsigned long long v = 0ull; // padding sorts last (descending order)
				if (i < nGoals) {
					const float c = __hip_atomic_load(orderKeys + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					// non-negative floats order like their bit patterns; +1 keeps real entries above the padding
					v = ((unsigned long long)(__float_as_uint(c) + 1u) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
				}
				skey[i] = v;
			}
			__syncthreads();
			for (int kk = 2; kk <= P; kk <<= 1)
				for (int j = kk >> 1, lj = 31 - __clz(kk >> 1); j > 0; j >>= 1, lj--) {
					for (int t = tid; t < P / 2; t += WF_T) {
						const int i1 = ((t >> lj) << (lj + 1)) | (t & (j - 1)), i2 = i1 + j;
						const bool desc = (i1 & kk) == 0;
						const unsigned long long a = skey[i1], b = skey[i2];
						if ((a < b) == desc) {
							skey[i1] = b;
							skey[i2] = a;
						}
					}
					__syncthreads();
				}
			for (int i = tid; i < nGoals; i += WF_T)
				orderOut[i] = (int32_t)(0xFFFFFFFFu - (uint32_t)skey[i]);
		}
	}
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
	const int64_t pcells = padded_cells(rows, cols);
	int64_t b = round256(state_bytes(rows, cols)) + round256(pcells * 4 + 16) + 2ll * fcap * 8 + (int64_t)gcap * 12 + 16;
	return (b + 255) / 256 * 256;
}

bool wavefront_tiles_enabled()
{
	static const bool on = [] {
		const char* e = getenv("PP_WF_TILES");
		return !(e && e[0] == '0');
	}();
	return on;
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

hipError_t warm_up_wavefront(hipStream_t s, const MapView& m, int32_t* ctlDev)
{
	// nGoals = 0: the workgroup reads the goal counter, finds nothing to do and leaves; no other pointer is dereferenced
	hipLaunchKernelGGL(k_wavefront<false>, dim3(1), dim3(WF_T), 0, s, m, 0, nullptr, nullptr, nullptr, (int64_t)0, 0u, 0u, ctlDev, nullptr, (int*)(ctlDev + 1), 0, nullptr, nullptr,
		nullptr, nullptr, nullptr, WavefrontPublish {});
	return hipGetLastError();
}

hipError_t launch_wavefront(hipStream_t s, const MapView& m, int nGoals, const int32_t* goalCellsDev, float* costDev, void* workspaceDev,
	int64_t workspaceBytesPerSlot, int nSlots, int32_t* errorFlagDev, unsigned long long* profDev, bool tiledOut, const double* goalPosesDev, bool countersZeroed,
	const double* orderStartsDev, int32_t* orderOutDev, int* doneCounterDev, float* orderKeysDev, const WavefrontPublish& pub)
{
	if (nGoals > WF_LCAP)
		orderOutDev = nullptr; // the epilogue sorts in the LDS sort buffer
	if (nGoals <= 0)
		return hipSuccess;
	uint32_t fcap, gcap;
	wf_caps(m.rows, m.cols, fcap, gcap);
	if (pub.tilesCtl && pub.tilesFallback && !profDev && pub.occBits && wavefront_tiles_enabled() && wavefront_tiles_supported(m.rows, m.cols)) {
		// The tile form (pp_wavefront_tiles.hip) builds the fields; behind it, on the same stream, the ordered kernel takes the goals it handed
		// over (their number is a device word: normally 0, and its few workgroups leave at once), then the hand-out order if the planner wants one.
		hipError_t e = launch_wavefront_tiles(s, m, nGoals, goalCellsDev, costDev, tiledOut, goalPosesDev, orderStartsDev, orderKeysDev, pub);
		if (e != hipSuccess)
			return e;
		WavefrontPublish fb = pub;
		fb.slotList = pub.tilesFallback; // plain goal / slot numbers: already claimed by the wave that handed them over
		fb.claimed = nullptr;
		fb.urgent = nullptr;
		fb.urgentHead = nullptr;
		fb.nGoalsDev = pub.tilesCtl + 2;
		fb.goalCounter = pub.tilesCtl + 3;
		fb.exitCounter = pub.tilesCtl + 4;
		fb.resetOnExit = pub.tilesCtl + 2;
		fb.agentPoseLoads = pub.claimed != nullptr;
		fb.tilesCtl = nullptr;
		const int fgrid = nSlots < 16 ? (nSlots < nGoals ? nSlots : nGoals) : (nGoals < 16 ? nGoals : 16);
		hipStream_t fs = s;
		if (pub.fallbackStream && pub.fallbackEvent) {
			e = hipEventRecord(pub.fallbackEvent, s);
			if (e == hipSuccess)
				e = hipStreamWaitEvent(pub.fallbackStream, pub.fallbackEvent, 0);
			if (e != hipSuccess)
				return e;
			fs = pub.fallbackStream;
		}
		fb.fallbackStream = nullptr;
		fb.fallbackEvent = nullptr;
		hipLaunchKernelGGL(k_wavefront<false>, dim3(fgrid), dim3(WF_T), 0, fs, m, nGoals, goalCellsDev, costDev, workspaceDev, workspaceBytesPerSlot, fcap, gcap, errorFlagDev,
			(unsigned long long*)nullptr, fb.goalCounter, tiledOut ? 1 : 0, goalPosesDev, orderStartsDev, (int32_t*)nullptr, (int*)nullptr, orderStartsDev ? orderKeysDev : nullptr, fb);
		e = hipGetLastError();
		if (e != hipSuccess)
			return e;
		if (orderOutDev && orderStartsDev)
			e = launch_order_by_key(s, nGoals, orderKeysDev, orderOutDev);
		return e;
	}
	int grid = nGoals < nSlots ? nGoals : nSlots;
	if (const char* e = getenv("PP_WF_GRID")) { // diagnostic: fewer resident workgroups (is a goal's latency independent of its neighbours on the CU?)
		const int g = atoi(e);
		if (g > 0 && g < grid)
			grid = g;
	}
	// errorFlagDev[0] = overflow flag, errorFlagDev[1] = next-goal counter
	if (!countersZeroed) {
		hipError_t e = hipMemsetAsync(pub.goalCounter ? (void*)pub.goalCounter : (void*)(errorFlagDev + 1), 0, sizeof(int), s);
		if (e != hipSuccess)
			return e;
	}
	if (profDev)
		hipLaunchKernelGGL(k_wavefront<true>, dim3(grid), dim3(WF_T), 0, s, m, nGoals, goalCellsDev, costDev, workspaceDev, workspaceBytesPerSlot, fcap, gcap, errorFlagDev,
			profDev, pub.goalCounter ? pub.goalCounter : (int*)(errorFlagDev + 1), tiledOut ? 1 : 0, goalPosesDev, orderStartsDev, orderOutDev, doneCounterDev, orderKeysDev, pub);
	else
		hipLaunchKernelGGL(k_wavefront<false>, dim3(grid), dim3(WF_T), 0, s, m, nGoals, goalCellsDev, costDev, workspaceDev, workspaceBytesPerSlot, fcap, gcap, errorFlagDev,
			profDev, pub.goalCounter ? pub.goalCounter : (int*)(errorFlagDev + 1), tiledOut ? 1 : 0, goalPosesDev, orderStartsDev, orderOutDev, doneCounterDev, orderKeysDev, pub);
	return hipGetLastError();
}

} // namespace pph
