// obst_wavefront, tile form -- ObstaclesHeuristic::Update (algo/heuristics.cpp:106-153) on gfx950 as the FIXED POINT it defines.
//
// The reference pops a sorted open list (LIFO among equal costs, utils/frontier.h:39-48,83-91) and never relaxes
// (heuristics.cpp:137-150): a cell keeps the cost it was given when the FIRST of its neighbours was popped.  Pops are
// non-decreasing in cost, so that neighbour is the cell's allowed, reached neighbour of smallest cost:
//     cost[n] = fl(cost[p] + edge(p, n)),   p = argmin { cost[m] : m neighbour of n, move m -> n allowed },        (*)
// edge = 1.0f or sqrtf(2.0f), "allowed" = n free and, for a diagonal move, not both orthogonal cells occupied (heuristics.cpp:127-132).
// The pop order among EQUAL costs decides the value only when a straight and a diagonal neighbour tie for that minimum (two
// straight or two diagonal ones give the same sum).  Such a tie needs two neighbours of n with bit-equal f32 costs whose paths
// differ in the parity of their straight moves (n's straight and diagonal neighbours lie on different colours of the chessboard):
// it does not occur on ordinary maps, it is DETECTED here whenever it does, and the goal is then rebuilt by the ordered kernel
// (pp_wavefront.hip).  Without such a tie (*) has exactly one solution (parents are strictly cheaper, every chain ends at the goal
// cell), which can be computed in any order.  tests/cpp/model_tile_field.cpp is the CPU statement of what follows, compared with the
// oracle's sequential restatement in tests/test_tile_field_model.py.
//
// Order used here: ONE WAVE PER GOAL, the grid in tiles of 64 x 64 cells, the tile being worked on resident in LDS.
//   * a tile is solved FROM SCRATCH from its one-cell halo (the neighbouring tiles' border cells as they stand, read back from the
//     output field) by bucket rounds: round k settles every undiscovered cell that has an allowed neighbour of cost in [k, k+1)
//     -- every edge costs >= 1 and f32 rounding is monotone, so those neighbours are final, and the new costs land in buckets
//     k+1 / k+2;
//   * lane = row of the tile.  Bucket membership, "closed" (occupied or discovered) and the four diagonal-allowed masks are 64-bit
//     words in registers; the candidate set of a round is a dozen shifts / ANDs / ORs with the two neighbouring rows' words
//     (DPP wave shifts), and each candidate PULLS its cost: eight LDS reads, two minima, one add.  No open list, no sort, no
//     atomics, no barrier; global memory is touched at tile load (occupancy bits, halo) and tile store (256-byte lines, each
//     written whole) only -- the 4-byte scattered stores of the ordered kernel (7x write amplification) are gone;
//   * tiles are taken in order of the smallest changed halo cost; a solved tile re-queues a neighbour only where one of its
//     changed border cells could be (or have been) the parent of a neighbour's border cell (cost[parent] <= cost[child] - 1).
//     When no tile is queued every cell satisfies (*) against its final neighbours.  A run that does not settle within a visit
//     budget is handed to the ordered kernel like a tie.
// Measured figures (tile visits per tile, rounds per visit, passes per round): DESIGN.md 4.3b.
// Algorithmic bytes as for the ordered kernel: 9 B/cell.
#include <cstdio>
#include "pp_internal.hpp"

#include <cstdlib>
#include <type_traits>

using namespace ppd;

namespace {

constexpr int TT = 64; // rows of a tile = lanes of a wave
// columns of a tile: TW = 64 (one 64-bit mask word per row) or 32 (half the LDS per wave: twice the waves per CU)
template <int TW>
struct TileShape {
	static constexpr int LS = TW + 2;         // LDS row stride in floats: the tile and its halo
	static constexpr int LN = (TT + 2) * LS;  // floats
	static constexpr int LNP = (LN + 3) & ~3;
	typedef typename std::conditional<TW == 64, uint64_t, uint32_t>::type Mask;
};
constexpr uint32_t kInfBits = 0x7F800000u;
#ifndef PP_WF_TILES_PRIO_DEFAULT
#define PP_WF_TILES_PRIO_DEFAULT 1 // as the search rows' (PP_ROWS_PRIO): measured 22.9-23.6 k plans/s against 19.3-22.4 k at 0 and 19.4 k at 2 (gpurun_out/r4_sweep_prio.txt)
#endif
#ifndef PP_WF_TILE_WIDTH_DEFAULT
#define PP_WF_TILE_WIDTH_DEFAULT 32 // measured in the pipeline (gpurun_out/r4_sweep_tw32*.txt): 26.3-26.9 k plans/s at 4096 search rows against 22.8 k with 64-column tiles -- eight waves
                                   // fit a pack's 93 KB instead of five; alone the two widths are equal (80-83 ms per 4096 goals)
#endif

__device__ __forceinline__ void wave_sync()
{
	// the lanes of a wave exchange values through LDS: its LDS operations execute in issue order, so only the compiler has to
	// be kept from moving them across this point
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}
/// lane i receives lane i-1's value, lane 0 keeps `first` (DPP wave_shr:1)
__device__ __forceinline__ uint32_t from_prev_lane(uint32_t v, uint32_t first) { return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xF, 0xF, false); }
/// lane i receives lane i+1's value, lane 63 keeps `last` (DPP wave_shl:1)
__device__ __forceinline__ uint32_t from_next_lane(uint32_t v, uint32_t last) { return (uint32_t)__builtin_amdgcn_update_dpp((int)last, (int)v, 0x130, 0xF, 0xF, false); }
__device__ __forceinline__ uint64_t from_prev_lane64(uint64_t v, uint64_t first)
{
	return (uint64_t)from_prev_lane((uint32_t)v, (uint32_t)first) | ((uint64_t)from_prev_lane((uint32_t)(v >> 32), (uint32_t)(first >> 32)) << 32);
}
__device__ __forceinline__ uint64_t from_next_lane64(uint64_t v, uint64_t last)
{
	return (uint64_t)from_next_lane((uint32_t)v, (uint32_t)last) | ((uint64_t)from_next_lane((uint32_t)(v >> 32), (uint32_t)(last >> 32)) << 32);
}
__device__ __forceinline__ uint32_t mask_from_prev(uint32_t v, uint32_t first) { return from_prev_lane(v, first); }
__device__ __forceinline__ uint32_t mask_from_next(uint32_t v, uint32_t last) { return from_next_lane(v, last); }
__device__ __forceinline__ uint64_t mask_from_prev(uint64_t v, uint64_t first) { return from_prev_lane64(v, first); }
__device__ __forceinline__ uint64_t mask_from_next(uint64_t v, uint64_t last) { return from_next_lane64(v, last); }
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
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
/// smallest of non-negative floats (their bit patterns order like the values; +inf = "none")
__device__ __forceinline__ float wave_min_nonneg(float v) { return __uint_as_float(wave_min_u32(__float_as_uint(v))); }

/// cell (i, j) of a 64 x TW tile that lane `e % 64` handles in step `e / 64` of a pass over the tile, chosen so that one step
/// covers whole lines of the output field: tiled field -> consecutive cells of a field tile; row-major -> a row of the tile
template <int TW>
__device__ __forceinline__ void tile_cell(int tiledOut, int e, int& i, int& j)
{
	if (tiledOut) {
		const int ft = e >> (2 * kFieldTileLog2), within = e & ((1 << (2 * kFieldTileLog2)) - 1);
		constexpr int perRow = TW >> kFieldTileLog2;
		i = ((ft / perRow) << kFieldTileLog2) + (within >> kFieldTileLog2);
		j = ((ft % perRow) << kFieldTileLog2) + (within & kFieldTileMask);
	} else {
		i = e / TW;
		j = e % TW;
	}
}

/// occupancy as bits: word (row + 1) * wpr + (col / 64 + 1), bit col % 64; one word of padding on every side, everything
/// outside the map occupied
__global__ void __launch_bounds__(256) k_occ_bits(const uint8_t* __restrict__ occ8, int rows, int cols, int wpr, int nWordRows, uint64_t* __restrict__ bits)
{
	const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); // one wave per word
	const int lane = threadIdx.x & 63;
	if (w >= (int64_t)nWordRows * wpr)
		return;
	const int pr = (int)(w / wpr), pw = (int)(w - (int64_t)pr * wpr);
	const int r = pr - 1, c = (pw - 1) * 64 + lane;
	bool o = true;
	if (r >= 0 && r < rows && pw >= 1 && c < cols)
		o = occ8[(int64_t)r * cols + c] != 0;
	const uint64_t b = __ballot(o);
	if (lane == 0)
		bits[w] = b;
}

struct TilesArgs {
	MapView m;
	int nGoals;
	const int32_t* goalCells;
	const double* goalPoses;
	float* costOut;
	int tiledOut;
	int* ctl;        // [0] goal counter, [1] exit counter, [2] number of goals handed to the ordered kernel (fbList)
	int32_t* fbList; // goals (field slots) this launch could not certify: a tie of (*) or a run that did not settle
	const double* orderStarts;
	float* orderKeys;
	unsigned long long* stats; // optional: {goals, tile visits, rounds, candidate passes, cells, handed over, cycles}
	int forceFallbackEvery;    // test hook: every n-th goal is handed to the ordered kernel whatever its outcome (0 = never)
	int ldsPerWave;            // bytes of LDS per wave of a workgroup
	int prio;                  // s_setprio level of the waves (0..3)
	pph::WavefrontPublish pub;
};

// A workgroup is a PACK of independent waves (no barrier, no shared data: each wave has its own slice of the workgroup's LDS and its own
// goals).  Waves are packed so that the workgroup's LDS exceeds half a CU's: at most one pack per CU, which always leaves room for a
// workgroup of the search grid (57 KB, 256 VGPRs per wave) next to it -- single-wave workgroups fill every CU's LDS eight at a time and
// keep the persistent search grid's workgroups from becoming resident (measured: 4096 search rows, 1000 of them busy).
// kGQ: the goal's tile queue lives in global memory (one region per wave of the launch, A.pub.tilesQueue) instead of LDS.  At 4096^2 the queue
// is 16 KB per wave against 9 KB for the tile itself -- four waves per CU with it in LDS, eight without; the queue is read once per tile visit
// (16 KB from the L2 against a visit's ~100 rounds) and written by lane 0 only, through agent-scope accesses (the wave's own L1 is not coherent
// with its stores).
template <int TW, bool kProf, bool kGQ>
__global__ void __launch_bounds__(512) k_wavefront_tiles(TilesArgs A)
{
	// kProf (diagnostic instantiation, pp_obstacle_heuristic_tiles_stats): shader-clock sums per phase of a tile visit
	unsigned long long ph[5] = { 0, 0, 0, 0, 0 }, tl = 0;
#define TILE_STAMP(i)                                        \
	if (kProf) {                                             \
		const unsigned long long now_ = __builtin_readcyclecounter(); \
		ph[i] += now_ - tl;                                  \
		tl = now_;                                           \
	}
	typedef typename TileShape<TW>::Mask Mask;
	constexpr int LS = TileShape<TW>::LS, LNP = TileShape<TW>::LNP;
	constexpr Mask kAll = (Mask)~(Mask)0;
	extern __shared__ __attribute__((aligned(16))) uint32_t smemAll[];
	const int lane = threadIdx.x & 63;
	uint32_t* const smemRaw = smemAll + (size_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * (size_t)(A.ldsPerWave >> 2);
	// issue priority next to the search grid's waves (which run at PP_ROWS_PRIO = 1): a wave here is a chain of short dependent steps that
	// issues an instruction every ~9 cycles; ahead of the search waves it costs them few slots, behind them it waits for theirs
	if (A.prio == 1)
		__builtin_amdgcn_s_setprio(1);
	else if (A.prio == 2)
		__builtin_amdgcn_s_setprio(2);
	else if (A.prio == 3)
		__builtin_amdgcn_s_setprio(3);
	const MapView& m = A.m;
	const int rows = m.rows, cols = m.cols;
	const int TR = (rows + TT - 1) / TT, TC = (cols + TW - 1) / TW, nTiles = TR * TC;
	float* const L = reinterpret_cast<float*>(smemRaw);                // [LN] the tile being solved, halo included
	const uint32_t* const Lu = smemRaw;
	// the goal's tile queue: 16 bits per tile -- bits 0..14 the bucket of the smallest changed halo cost of a QUEUED tile (0x7FFF = not queued),
	// bit 15 "solved at least once".  (Four bytes + one per tile until the 4096^2 map of config 5 needed 41 KB of LDS per wave for it: two waves per CU.)
	uint16_t* const tqL = reinterpret_cast<uint16_t*>(L + LNP); // [nTiles, padded to an even number]
	const int nTileWords = (nTiles + 1) >> 1;
	uint32_t* const tqG = kGQ ? A.pub.tilesQueue + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (size_t)nTileWords : nullptr;
	constexpr uint32_t kNotQueued = 0x7FFFu;
	auto tq_word = [&](int w) -> uint32_t { // two tiles
		if (kGQ)
			return __hip_atomic_load(tqG + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return reinterpret_cast<const uint32_t*>(tqL)[w];
	};
	auto tq_get = [&](int t) -> uint32_t {
		if (kGQ)
			return (uint32_t)__hip_atomic_load(reinterpret_cast<uint16_t*>(tqG) + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return (uint32_t)tqL[t];
	};
	auto tq_set = [&](int t, uint32_t v) { // (lanes write different tiles)
		if (kGQ)
			__hip_atomic_store(reinterpret_cast<uint16_t*>(tqG) + t, (uint16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		else
			tqL[t] = (uint16_t)v;
	};
	auto tq_sync = [&]() { // lane 0's queue writes before the other lanes' reads
		if (kGQ)
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		else
			wave_sync();
	};
	const float kInf = __builtin_huge_valf();
	const float kDiag = sqrtf(2.0f); // std::sqrt(2.0f), heuristics.cpp:134
	const int64_t cells = (int64_t)rows * cols;
	const int64_t fieldElems = A.tiledOut ? (int64_t)field_tiled_elems(rows, cols) : cells;
	const uint64_t* const occW = A.pub.occBits;
	const int wpr = (cols + 63) / 64 + 2; // (occ_bits_dims)
	auto out_index = [&](int r, int c) -> size_t { return A.tiledOut ? field_tiled_index(cols, r, c) : (size_t)r * cols + c; };

	int pendingSlot = -1;
	bool ringTurn = true;
	unsigned long long stVisits = 0, stRounds = 0, stPasses = 0, stCells = 0, stGoals = 0, stFb = 0;
	const unsigned long long t0 = A.stats ? __builtin_readcyclecounter() : 0ull;
	for (;;) {
		// ---- hand-out: as k_wavefront's (pp_wavefront.hip), lane 0 in the place of thread 0
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every lane's stores of the previous goal have left the wave
		int next = -1, st = -1;
		if (lane == 0) {
			if (pendingSlot >= 0) {
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				const unsigned long long t = __hip_atomic_fetch_add(A.pub.readyTail, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(A.pub.ready + (t & A.pub.readyMask), ((t + 1ull) << 32) | (unsigned long long)(uint32_t)pendingSlot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			auto take_urgent = [&]() -> int {
				for (;;) {
					unsigned long long h = __hip_atomic_load(A.pub.urgentHead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					const unsigned long long e = __hip_atomic_load(A.pub.urgent + (h & A.pub.urgentMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if ((uint32_t)(e >> 32) != (uint32_t)(h + 1ull))
						return -1;
					if (!__hip_atomic_compare_exchange_strong(A.pub.urgentHead, &h, h + 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
						continue;
					const int slot = (int)((uint32_t)e & pph::kSlotMask);
					int expect = (int)(((uint32_t)e >> pph::kSlotBits) << 1);
					if (__hip_atomic_compare_exchange_strong(A.pub.claimed + slot, &expect, expect | 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
						return slot;
				}
			};
			if (A.pub.urgent && ringTurn)
				next = take_urgent();
			while (next < 0) {
				const int gi = atomicAdd(A.ctl, 1);
				if (gi >= A.nGoals)
					break;
				if (!A.pub.slotList) {
					next = gi;
				} else {
					const uint32_t e = (uint32_t)A.pub.slotList[gi];
					const int slot = (int)(e & pph::kSlotMask);
					int expect = (int)((e >> pph::kSlotBits) << 1);
					if (!A.pub.claimed || __hip_atomic_compare_exchange_strong(A.pub.claimed + slot, &expect, expect | 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
						next = slot;
				}
			}
			ringTurn = !ringTurn;
			if (next >= 0) {
				if (A.goalPoses) { // WorldPositionToGridCell(bounded), heuristics.cpp:115
					double px, py;
					if (A.pub.claimed) { // (the pose may have been written while this launch was running: see k_wavefront)
						px = __hip_atomic_load(A.goalPoses + 3 * (size_t)next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						py = __hip_atomic_load(A.goalPoses + 3 * (size_t)next + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					} else {
						px = A.goalPoses[3 * (size_t)next];
						py = A.goalPoses[3 * (size_t)next + 1];
					}
					int row, col;
					world_to_cell(m, px, py, row, col);
					st = inside_map(m, row, col) ? row * m.cols + col : -1;
				} else {
					st = A.goalCells[next];
				}
			}
		}
		const int g = __builtin_amdgcn_readfirstlane(next);
		const int start = __builtin_amdgcn_readfirstlane(st);
		pendingSlot = -1;
		if (g < 0)
			break;
		if (kProf)
			stGoals++;
		float* const cost = A.costOut + (int64_t)g * fieldElems;
		const int goalR = start >= 0 ? start / cols : -1, goalC = start >= 0 ? start - goalR * cols : -1;

		if (kGQ) {
			for (int w = lane; w < nTileWords; w += 64)
				__hip_atomic_store(tqG + w, kNotQueued | (kNotQueued << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		} else {
			for (int t = lane; t < ((nTiles + 1) & ~1); t += 64)
				tqL[t] = (uint16_t)kNotQueued;
		}
		tq_sync();
		if (start >= 0 && lane == 0)
			tq_set((goalR / TT) * TC + goalC / TW, 0u);
		tq_sync();
		bool flagged = false; // a tie of (*) somewhere, or the visit budget spent
		int visits = 0;
		const int visitBudget = 6 * nTiles + 64;

		for (;;) {
			// ---- the queued tile whose changed halo is cheapest
			uint32_t bestBits = kNotQueued;
			int bestT = -1;
			if (kGQ)
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the previous visit's queue entries (written by lanes 0..8, read by all)
			for (int t2 = lane; t2 < (nTiles + 1) / 2; t2 += 64) { // two tiles per word
				const uint32_t w = tq_word(t2);
				const uint32_t lo = w & 0x7FFFu, hi = (w >> 16) & 0x7FFFu;
				if (lo < bestBits) {
					bestBits = lo;
					bestT = 2 * t2;
				}
				if (hi < bestBits) {
					bestBits = hi;
					bestT = 2 * t2 + 1;
				}
			}
			const uint32_t minBits = wave_min_u32(bestBits);
			if (minBits == kNotQueued)
				break; // nothing queued: the field is settled
			const uint64_t who = __ballot(bestBits == minBits);
			const int t = __builtin_amdgcn_readlane(bestT, (int)__builtin_ctzll(who));
			if (++visits > visitBudget) {
				flagged = true;
				break;
			}
			if (kProf)
				tl = __builtin_readcyclecounter();
			const int tr = t / TC, tc = t - tr * TC;
			const int r0 = tr * TT, c0 = tc * TW;
			auto solved = [&](int dtr, int dtc) -> bool {
				const int a = tr + dtr, b = tc + dtc;
				return a >= 0 && b >= 0 && a < TR && b < TC && (tq_get(a * TC + b) & 0x8000u);
			};
			const bool selfSolved = solved(0, 0);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the halo is read back from lines the previous visits stored
			// ---- occupancy: this row's bits, the bits left and right of them, the rows above and below the tile.  A tile's TW columns
			// lie inside one 64-bit word of the padded bit grid (TW divides 64, tiles are aligned)
			auto occ_bits = [&](int prow, int cFirst) -> Mask { // TW bits from column cFirst of padded row prow
				const uint64_t w = occW[(size_t)prow * wpr + (size_t)((cFirst >> 6) + 1)];
				return (Mask)(w >> (cFirst & 63));
			};
			auto occ_bit = [&](int prow, int c) -> uint32_t { // (c = -1 and c = cols.. fall into the padding words: occupied)
				return (uint32_t)(occW[(size_t)prow * wpr + (size_t)((c >> 6) + 1)] >> (c & 63)) & 1u;
			};
			const Mask occ = occ_bits(r0 + lane + 1, c0);
			const uint32_t occL = occ_bit(r0 + lane + 1, c0 - 1), occR = occ_bit(r0 + lane + 1, c0 + TW);
			const Mask tH = occ_bits(r0, c0), bH = occ_bits(r0 + TT + 1, c0);
			const uint32_t tHL = occ_bit(r0, c0 - 1), tHR = occ_bit(r0, c0 + TW);
			const uint32_t bHL = occ_bit(r0 + TT + 1, c0 - 1), bHR = occ_bit(r0 + TT + 1, c0 + TW);
			// ---- halo: the neighbouring tiles' border cells as they stand (tiles not solved yet count as +inf), and this tile's own
			// border as it was after its previous solve
			float hT = kInf, hB = kInf, hL = kInf, hR = kInf, hC = kInf;
			float oT = kInf, oB = kInf, oL = kInf, oR = kInf;
			{
				const int cj = c0 + lane, ri = r0 + lane;
				if (lane < TW && cj < cols) {
					if (solved(-1, 0))
						hT = cost[out_index(r0 - 1, cj)];
					if (r0 + TT < rows && solved(1, 0))
						hB = cost[out_index(r0 + TT, cj)];
				}
				if (ri < rows) {
					if (solved(0, -1))
						hL = cost[out_index(ri, c0 - 1)];
					if (c0 + TW < cols && solved(0, 1))
						hR = cost[out_index(ri, c0 + TW)];
				}
				if (lane < 4) {
					const int dr = lane < 2 ? -1 : 1, dc = (lane & 1) ? 1 : -1;
					const int rr = dr < 0 ? r0 - 1 : r0 + TT, cc = dc < 0 ? c0 - 1 : c0 + TW;
					if (rr >= 0 && rr < rows && cc >= 0 && cc < cols && solved(dr, dc))
						hC = cost[out_index(rr, cc)];
				}
				if (selfSolved) {
					if (lane < TW && cj < cols) {
						oT = cost[out_index(r0, cj)];
						if (r0 + TT - 1 < rows)
							oB = cost[out_index(r0 + TT - 1, cj)];
					}
					if (ri < rows) {
						oL = cost[out_index(ri, c0)];
						if (c0 + TW - 1 < cols)
							oR = cost[out_index(ri, c0 + TW - 1)];
					}
				}
			}
			// ---- LDS: +inf everywhere (undiscovered or occupied), then the halo, then the goal
			{
				float4* const L4 = reinterpret_cast<float4*>(L);
				const float4 inf4 = make_float4(kInf, kInf, kInf, kInf);
				for (int i = lane; i < LNP / 4; i += 64)
					L4[i] = inf4;
			}
			wave_sync();
			if (lane < TW) {
				L[lane + 1] = hT;
				L[(TT + 1) * LS + lane + 1] = hB;
			}
			L[(lane + 1) * LS] = hL;
			L[(lane + 1) * LS + TW + 1] = hR;
			if (lane < 4)
				L[(lane < 2 ? 0 : (TT + 1) * LS) + ((lane & 1) ? TW + 1 : 0)] = hC;
			Mask closed = occ, cur = 0, nx1 = 0, nx2 = 0;
			const bool goalHere = goalR >= r0 && goalR < r0 + TT && goalC >= c0 && goalC < c0 + TW;
			if (goalHere && lane == goalR - r0) { // the reference pushes the goal cell even when it is occupied (heuristics.cpp:119-121)
				L[(lane + 1) * LS + (goalC - c0) + 1] = 0.0f;
				closed |= (Mask)1 << (goalC - c0);
				cur = (Mask)1 << (goalC - c0);
			}
			wave_sync();
			// buckets of the halo cells: (int)cost, -1 = none
			const int hbT = hT < kInf ? (int)hT : -1, hbB = hB < kInf ? (int)hB : -1, hbL = hL < kInf ? (int)hL : -1, hbR = hR < kInf ? (int)hR : -1, hbC = hC < kInf ? (int)hC : -1;
			int kmin;
			{
				uint32_t a = 0xFFFFFFFFu;
				a = min(a, (uint32_t)hbT);
				a = min(a, (uint32_t)hbB);
				a = min(a, (uint32_t)hbL);
				a = min(a, (uint32_t)hbR);
				a = min(a, (uint32_t)hbC); // (-1 = 0xFFFFFFFF: never the minimum)
				a = wave_min_u32(a);
				kmin = goalHere ? 0 : (int)a;
			}
			if (kProf) // (counters exist in the instantiation that reports them: launches with A.stats; eight instructions per candidate pass otherwise)
				stVisits++;
			TILE_STAMP(0) // loads, LDS set-up
			if (kmin >= 0) { // (a tile queued by a border cell that no free halo cell of it sees any more has nothing to start from)
				// ---- static masks.  Diagonal move into cell (i, j) from (i-1, j-1): blocked iff (i, j-1) and (i-1, j) are both occupied
				const Mask occUp = mask_from_prev(occ, tH), occDn = mask_from_next(occ, bH);
				const Mask occWst = (Mask)(occ << 1) | (Mask)occL, occEst = (Mask)(occ >> 1) | ((Mask)occR << (TW - 1));
				const Mask bNW = occUp & occWst, bNE = occUp & occEst, bSW = occDn & occWst, bSE = occDn & occEst; // blocked diagonal moves
				const Mask aNW = ~bNW, aNE = ~bNE, aSW = ~bSW, aSE = ~bSE;
				int k = kmin;
				for (;;) {
					// members of bucket k: `cur` inside the tile, halo cells by their bucket number
					const Mask topM = (Mask)__ballot(hbT == k), botM = (Mask)__ballot(hbB == k);
					const uint32_t cm = (uint32_t)__ballot(hbC == k) & 0xFu; // lanes 0..3 = corners TL, TR, BL, BR
					const uint32_t lr = (hbL == k ? 1u : 0u) | (hbR == k ? 2u : 0u);
					const uint32_t lrU = from_prev_lane(lr, cm & 3u), lrD = from_next_lane(lr, (cm >> 2) & 3u);
					const Mask up = mask_from_prev(cur, topM), dn = mask_from_next(cur, botM);
					const Mask mW = (Mask)(cur << 1) | (Mask)(lr & 1u), mE = (Mask)(cur >> 1) | ((Mask)(lr >> 1) << (TW - 1));
					const Mask mNW = ((Mask)(up << 1) | (Mask)(lrU & 1u)) & aNW, mNE = ((Mask)(up >> 1) | ((Mask)(lrU >> 1) << (TW - 1))) & aNE;
					const Mask mSW = ((Mask)(dn << 1) | (Mask)(lrD & 1u)) & aSW, mSE = ((Mask)(dn >> 1) | ((Mask)(lrD >> 1) << (TW - 1))) & aSE;
					Mask cand = (mW | mE | up | dn | mNW | mNE | mSW | mSE) & (Mask)~closed;
					if (kProf)
						stRounds++;
					TILE_STAMP(1) // a round's masks
					const float kNext2 = (float)(k + 2);
					// A lane settles TWO of its row's candidates per pass: the sixteen LDS reads of both are in flight together, so a row that the
					// front crosses at a flat angle (two or three candidates) costs one latency chain, not two or three.  (A candidate next to the
					// other one may read it before or after its cost is written: either way a cost of bucket k+1 or later, never the minimum.)
					auto settle = [&](int j, uint32_t w, uint32_t e, uint32_t n, uint32_t s, uint32_t nw, uint32_t ne, uint32_t sw, uint32_t se) -> float {
						// costs are non-negative floats (+inf = none): their bit patterns order like the values, so the minima are integer
						// minima (v_min3_u32; a float minimum first quiets its operands: six more instructions per candidate).  A blocked diagonal
						// move becomes all ones -- above every cost, +inf included -- by OR-ing the sign-extended bit of the blocked mask (v_bfe_i32 + v_or:
						// the and / compare / select it replaces was a quarter of a pass's instructions, with a hazard nop behind every compare)
						const uint32_t minS = min(min(w, e), min(n, s));
						auto blocked = [&](Mask b) -> uint32_t {
							if constexpr (sizeof(Mask) == 4)
								return (uint32_t)__builtin_amdgcn_sbfe((int)b, (unsigned)j, 1u);
							else
								return (uint32_t)-(int32_t)((uint32_t)(b >> j) & 1u);
						};
						uint32_t minD = nw | blocked(bNW);
						minD = min(minD, ne | blocked(bNE));
						minD = min(minD, sw | blocked(bSW));
						minD = min(minD, se | blocked(bSE));
						flagged |= minS == minD; // a straight and a diagonal neighbour tie for the minimum: the pop order would decide
						return minS <= minD ? __uint_as_float(minS) + 1.0f : __uint_as_float(minD) + kDiag; // float pathCost = transitionCost + m_cost[cell], heuristics.cpp:134-135
					};
					while (__ballot(cand != 0)) {
						if (kProf)
							stPasses++;
						if (cand != 0) {
							const int j0 = (int)__builtin_ctzll((uint64_t)cand);
							const Mask bit0 = (Mask)1 << j0;
							cand &= (Mask)~bit0;
							const bool two = cand != 0;
							const int j1 = two ? (int)__builtin_ctzll((uint64_t)cand) : j0;
							const Mask bit1 = (Mask)1 << j1;
							cand &= (Mask)~bit1;
							const int a0 = (lane + 1) * LS + j0 + 1, a1 = (lane + 1) * LS + j1 + 1;
							const uint32_t w0 = Lu[a0 - 1], e0 = Lu[a0 + 1], n0 = Lu[a0 - LS], s0 = Lu[a0 + LS];
							const uint32_t nw0 = Lu[a0 - LS - 1], ne0 = Lu[a0 - LS + 1], sw0 = Lu[a0 + LS - 1], se0 = Lu[a0 + LS + 1];
							const uint32_t w1 = Lu[a1 - 1], e1 = Lu[a1 + 1], n1 = Lu[a1 - LS], s1 = Lu[a1 + LS];
							const uint32_t nw1 = Lu[a1 - LS - 1], ne1 = Lu[a1 - LS + 1], sw1 = Lu[a1 + LS - 1], se1 = Lu[a1 + LS + 1];
							const float v0 = settle(j0, w0, e0, n0, s0, nw0, ne0, sw0, se0);
							const float v1 = settle(j1, w1, e1, n1, s1, nw1, ne1, sw1, se1);
							L[a0] = v0;
							if (two)
								L[a1] = v1;
							closed |= bit0 | bit1;
							nx1 |= (v0 < kNext2 ? bit0 : (Mask)0) | (v1 < kNext2 ? bit1 : (Mask)0);
							nx2 |= (v0 < kNext2 ? (Mask)0 : bit0) | (v1 < kNext2 ? (Mask)0 : bit1);
							if (kProf)
								stCells += two ? 2 : 1;
						}
						wave_sync();
					}
					TILE_STAMP(2) // its candidate passes
					cur = nx1;
					nx1 = nx2;
					nx2 = 0;
					if (__ballot(closed != kAll) == 0ull)
						break; // every free cell of the tile has its cost
					if (__ballot((cur | nx1) != 0) != 0ull) {
						k++;
					} else { // nothing pending inside: the next bucket that holds a halo cell, if any
						uint32_t nb = 0xFFFFFFFFu;
						nb = min(nb, hbT > k ? (uint32_t)hbT : 0xFFFFFFFFu);
						nb = min(nb, hbB > k ? (uint32_t)hbB : 0xFFFFFFFFu);
						nb = min(nb, hbL > k ? (uint32_t)hbL : 0xFFFFFFFFu);
						nb = min(nb, hbR > k ? (uint32_t)hbR : 0xFFFFFFFFu);
						nb = min(nb, hbC > k ? (uint32_t)hbC : 0xFFFFFFFFu);
						nb = wave_min_u32(nb);
						if (nb == 0xFFFFFFFFu)
							break;
						k = (int)nb;
					}
				}
			}
			wave_sync();
			// ---- which neighbouring tiles must be solved (again): a border cell whose value changed re-queues the tile of a free halo cell
			// next to it if that cell has no cost yet or one that a parent of the changed cell's (old or new) cost could explain
			{
				const int lc = lane < TW ? lane : 0; // (top / bottom rows: lanes beyond the tile's columns idle)
				const float nT = lane < TW ? L[LS + lc + 1] : kInf, nB = lane < TW ? L[TT * LS + lc + 1] : kInf, nL = L[(lane + 1) * LS + 1], nR = L[(lane + 1) * LS + TW];
				auto sees = [&](float p, float q, uint32_t qOcc) -> bool { return !qOcc && (q == kInf || p < q - 0.99f); };
				const uint32_t occLU = from_prev_lane(occL, tHL), occLD = from_next_lane(occL, bHL);
				const uint32_t occRU = from_prev_lane(occR, tHR), occRD = from_next_lane(occR, bHR);
				float pN = kInf, pS = kInf, pW = kInf, pE = kInf, pNW = kInf, pNE = kInf, pSW = kInf, pSE = kInf;
				{ // top row, lane = column
					const bool ch = lane < TW && __float_as_uint(nT) != __float_as_uint(oT);
					const float p = fminf(nT, oT);
					const uint32_t o0 = lc == 0 ? tHL : (uint32_t)(tH >> (lc - 1)) & 1u, o1 = (uint32_t)(tH >> lc) & 1u, o2 = lc == TW - 1 ? tHR : (uint32_t)(tH >> (lc + 1)) & 1u;
					const bool c0_ = ch && sees(p, L[lc], o0), c1_ = ch && sees(p, L[lc + 1], o1), c2_ = ch && sees(p, L[lc + 2], o2);
					if (c1_ || (lc > 0 && c0_) || (lc < TW - 1 && c2_))
						pN = p;
					if (lc == 0 && c0_)
						pNW = p;
					if (lc == TW - 1 && c2_)
						pNE = p;
				}
				{ // bottom row
					const bool ch = lane < TW && __float_as_uint(nB) != __float_as_uint(oB);
					const float p = fminf(nB, oB);
					const float* const Lb = L + (TT + 1) * LS;
					const uint32_t o0 = lc == 0 ? bHL : (uint32_t)(bH >> (lc - 1)) & 1u, o1 = (uint32_t)(bH >> lc) & 1u, o2 = lc == TW - 1 ? bHR : (uint32_t)(bH >> (lc + 1)) & 1u;
					const bool c0_ = ch && sees(p, Lb[lc], o0), c1_ = ch && sees(p, Lb[lc + 1], o1), c2_ = ch && sees(p, Lb[lc + 2], o2);
					if (c1_ || (lc > 0 && c0_) || (lc < TW - 1 && c2_))
						pS = p;
					if (lc == 0 && c0_)
						pSW = p;
					if (lc == TW - 1 && c2_)
						pSE = p;
				}
				{ // left column, lane = row
					const bool ch = __float_as_uint(nL) != __float_as_uint(oL);
					const float p = fminf(nL, oL);
					const bool c0_ = ch && sees(p, L[lane * LS], occLU), c1_ = ch && sees(p, L[(lane + 1) * LS], occL), c2_ = ch && sees(p, L[(lane + 2) * LS], occLD);
					if (c1_ || (lane > 0 && c0_) || (lane < 63 && c2_))
						pW = p;
					if (lane == 0 && c0_)
						pNW = fminf(pNW, p);
					if (lane == 63 && c2_)
						pSW = fminf(pSW, p);
				}
				{ // right column
					const bool ch = __float_as_uint(nR) != __float_as_uint(oR);
					const float p = fminf(nR, oR);
					const bool c0_ = ch && sees(p, L[lane * LS + TW + 1], occRU), c1_ = ch && sees(p, L[(lane + 1) * LS + TW + 1], occR), c2_ = ch && sees(p, L[(lane + 2) * LS + TW + 1], occRD);
					if (c1_ || (lane > 0 && c0_) || (lane < 63 && c2_))
						pE = p;
					if (lane == 0 && c0_)
						pNE = fminf(pNE, p);
					if (lane == 63 && c2_)
						pSE = fminf(pSE, p);
				}
				const float qN = wave_min_nonneg(pN), qS = wave_min_nonneg(pS), qW = wave_min_nonneg(pW), qE = wave_min_nonneg(pE);
				const float qNW = wave_min_nonneg(pNW), qNE = wave_min_nonneg(pNE), qSW = wave_min_nonneg(pSW), qSE = wave_min_nonneg(pSE);
				// lanes 0..7: one neighbour each (N S W E NW NE SW SE), lane 8: this tile -- nine different queue entries, one round trip
				if (lane < 9) {
					const int dtr = lane < 2 ? (lane == 0 ? -1 : 1) : (lane < 4 ? 0 : (lane < 6 ? -1 : (lane < 8 ? 1 : 0)));
					const int dtc = lane < 2 ? 0 : (lane < 4 ? (lane == 2 ? -1 : 1) : (lane < 8 ? ((lane & 1) ? 1 : -1) : 0));
					const float p = lane == 0 ? qN : lane == 1 ? qS : lane == 2 ? qW : lane == 3 ? qE : lane == 4 ? qNW : lane == 5 ? qNE : lane == 6 ? qSW : lane == 7 ? qSE : kInf;
					const int a = tr + dtr, b = tc + dtc;
					if (lane == 8) {
						tq_set(t, 0x8000u | kNotQueued); // solved, not queued
					} else if (p < kInf && a >= 0 && b >= 0 && a < TR && b < TC) {
						const uint32_t bucket = p < 32766.0f ? (uint32_t)p : 32766u, old = tq_get(a * TC + b);
						if (bucket < (old & 0x7FFFu))
							tq_set(a * TC + b, (old & 0x8000u) | bucket);
					}
				}
			}
			TILE_STAMP(3) // neighbours re-queued
			// ---- the tile's costs to the output field: whole 256-byte lines (8 x 8-tiled field: one field tile per step; row-major: one row)
			for (int u = 0; u < TW; u++) {
				int i, j;
				tile_cell<TW>(A.tiledOut, u * TT + lane, i, j);
				const int r = r0 + i, c = c0 + j;
				if (r < rows && c < cols)
					cost[out_index(r, c)] = L[(i + 1) * LS + j + 1];
			}
			wave_sync();
			TILE_STAMP(4) // stores issued
		}
		flagged = __ballot(flagged) != 0ull;
		// ---- +inf for the tiles the wavefront never reached (heuristics.cpp:108-113)
		if (!flagged) {
			for (int t = 0; t < nTiles; t++) {
				if (tq_get(t) & 0x8000u)
					continue;
				const int tr = t / TC, tc = t - tr * TC;
				for (int u = 0; u < TW; u++) {
					int i, j;
					tile_cell<TW>(A.tiledOut, u * TT + lane, i, j);
					const int r = tr * TT + i, c = tc * TW + j;
					if (r < rows && c < cols)
						cost[out_index(r, c)] = kInf;
				}
			}
		}
		if (A.forceFallbackEvery > 0 && (g % A.forceFallbackEvery) == A.forceFallbackEvery - 1)
			flagged = true;
		if (flagged) {
			// not certified: the ordered kernel rebuilds this goal's field from scratch (launch_wavefront queues it behind this launch)
			if (kProf)
				stFb++;
			if (lane == 0)
				A.fbList[atomicAdd(A.ctl + 2, 1)] = g;
			continue;
		}
		if (A.pub.ready)
			pendingSlot = g;
		if (A.orderKeys) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			if (lane == 0) {
				int row, col;
				world_to_cell(m, A.orderStarts[3 * (size_t)g], A.orderStarts[3 * (size_t)g + 1], row, col);
				float c = kInf;
				if (inside_map(m, row, col))
					c = cost[out_index(row, col)];
				__hip_atomic_store(A.orderKeys + g, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
	}
	if (kProf && A.stats && lane == 0) {
		atomicAdd(A.stats + 0, stGoals);
		atomicAdd(A.stats + 1, stVisits);
		atomicAdd(A.stats + 2, stRounds);
		atomicAdd(A.stats + 3, stPasses);
		atomicAdd(A.stats + 5, stFb);
		atomicAdd(A.stats + 6, (unsigned long long)(__builtin_readcyclecounter() - t0));
		if (kProf)
			for (int i = 0; i < 5; i++)
				atomicAdd(A.stats + 8 + i, ph[i]);
	}
#undef TILE_STAMP
	if (kProf && A.stats)
		atomicAdd(A.stats + 4, stCells); // (counted per lane)
	// the last wave to leave sets the goal counter back for the stream's next launch (the number of handed-over goals stays: the
	// ordered kernel's launch behind this one reads it and sets it back in turn)
	if (lane == 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		if (__hip_atomic_fetch_add(A.ctl + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)(gridDim.x * (blockDim.x >> 6)) - 1) {
			__hip_atomic_store(A.ctl, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(A.ctl + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

/// the planner's hand-out order (pp_planner.hip): goal indices by decreasing field value at the start pose (+inf = unreachable first);
/// the ordered kernel does this in its last workgroup, the tile form in one small launch behind its two kernels.  n <= 4096.
__global__ void __launch_bounds__(512) k_order_by_key(int n, const float* __restrict__ keys, int32_t* __restrict__ orderOut)
{
	__shared__ unsigned long long skey[4096];
	const int tid = threadIdx.x;
	int P = 1;
	while (P < n)
		P <<= 1;
	for (int i = tid; i < P; i += 512) {
		unsigned long long v = 0ull;
		if (i < n)
			v = ((unsigned long long)(__float_as_uint(keys[i]) + 1u) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
		skey[i] = v;
	}
	__syncthreads();
	for (int kk = 2; kk <= P; kk <<= 1)
		for (int j = kk >> 1, lj = 31 - __clz(kk >> 1); j > 0; j >>= 1, lj--) {
			for (int t = tid; t < P / 2; t += 512) {
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
	for (int i = tid; i < n; i += 512)
		orderOut[i] = (int32_t)(0xFFFFFFFFu - (uint32_t)skey[i]);
}

int tile_width()
{
	static const int w = [] {
		const char* e = getenv("PP_WF_TILE_WIDTH"); // 64 or 32 columns per tile
		return e && atoi(e) == 64 ? 64 : (e && atoi(e) == 32 ? 32 : PP_WF_TILE_WIDTH_DEFAULT);
	}();
	return w;
}

size_t tiles_count(int rows, int cols)
{
	const int tw = tile_width();
	return (size_t)((rows + TT - 1) / TT) * (size_t)((cols + tw - 1) / tw);
}

size_t tiles_lds_bytes(int rows, int cols, bool queueInGlobal = false)
{
	const size_t nTiles = tiles_count(rows, cols);
	return ((size_t)(tile_width() == 64 ? TileShape<64>::LNP : TileShape<32>::LNP) * 4 + (queueInGlobal ? 0 : ((nTiles + 1) & ~(size_t)1) * 2) + 15) & ~(size_t)15;
}

/// waves per workgroup: as many as make the workgroup's LDS exceed half of a CU's 160 KB (see k_wavefront_tiles), at most 8
int waves_per_pack(size_t ldsPerWave)
{
	static const int forced = [] {
		const char* e = getenv("PP_WF_TILES_PACK"); // tuning: waves per workgroup (1..8)
		return e ? atoi(e) : 0;
	}();
	if (forced >= 1 && forced <= 8)
		return forced;
	int n = (int)((80 * 1024) / ldsPerWave) + 1;
	while (n > 1 && (size_t)n * ldsPerWave > 160 * 1024)
		n--;
	return n > 8 ? 8 : n;
}

void launch_tiles_kernel(hipStream_t s, int waves, TilesArgs A)
{
	// the tile queue in global memory: where the caller brought regions for it (a pipeline on a large map, wavefront_tiles_queue_words) and nobody asked for counters
	const bool gq = A.pub.tilesQueue && A.pub.tilesQueueWaves > 0 && !A.stats;
	const size_t lds = tiles_lds_bytes(A.m.rows, A.m.cols, gq);
	static const bool trace = getenv("PP_WF_TILES_TRACE") != nullptr; // tests: which instantiation ran
	if (trace && A.nGoals > 0) // (not the warm-up launch)
		fprintf(stderr, "[tiles] launch: %d waves, tile queue in %s\n", waves, gq ? "global memory" : "LDS");
	// packs protect the pipeline's persistent search grid; a launch outside a pipeline has the chip to itself: single waves, eight per CU
	const int pack = A.pub.ready ? waves_per_pack(lds) : 1;
	// (a pack of eight waves whose slices add up to less than half a CU's LDS asks for 82 KB all the same: one pack per CU is the point)
	static const size_t minPackLds = [] {
		const char* e = getenv("PP_WF_TILES_PACK_KB"); // tuning: LDS a pack asks for at least (a search grid with more waves per CU leaves less than 82 KB)
		const int v = e ? atoi(e) : 82;
		return (size_t)(v < 1 ? 1 : (v > 160 ? 160 : v)) * 1024;
	}();
	const size_t packLds = pack > 1 && (size_t)pack * lds < minPackLds ? minPackLds : (size_t)pack * lds;
	A.ldsPerWave = (int)lds;
	static const int prio = [] {
		const char* e = getenv("PP_WF_TILES_PRIO");
		const int v = e ? atoi(e) : PP_WF_TILES_PRIO_DEFAULT;
		return v < 0 ? 0 : (v > 3 ? 3 : v);
	}();
	A.prio = prio;
	if (gq && waves > A.pub.tilesQueueWaves / pack * pack)
		waves = A.pub.tilesQueueWaves / pack * pack; // (a region per wave of every workgroup)
	const int grid = (waves + pack - 1) / pack;
	static const bool attr = [] { // a pack's dynamic LDS goes beyond the 64 KB a launch gets without asking
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wavefront_tiles<64, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wavefront_tiles<32, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wavefront_tiles<64, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wavefront_tiles<32, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wavefront_tiles<64, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wavefront_tiles<32, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		return true;
	}();
	(void)attr;
	if (tile_width() == 64) {
		if (A.stats)
			hipLaunchKernelGGL((k_wavefront_tiles<64, true, false>), dim3(grid), dim3(64 * pack), packLds, s, A);
		else if (gq)
			hipLaunchKernelGGL((k_wavefront_tiles<64, false, true>), dim3(grid), dim3(64 * pack), packLds, s, A);
		else
			hipLaunchKernelGGL((k_wavefront_tiles<64, false, false>), dim3(grid), dim3(64 * pack), packLds, s, A);
	} else {
		if (A.stats)
			hipLaunchKernelGGL((k_wavefront_tiles<32, true, false>), dim3(grid), dim3(64 * pack), packLds, s, A);
		else if (gq)
			hipLaunchKernelGGL((k_wavefront_tiles<32, false, true>), dim3(grid), dim3(64 * pack), packLds, s, A);
		else
			hipLaunchKernelGGL((k_wavefront_tiles<32, false, false>), dim3(grid), dim3(64 * pack), packLds, s, A);
	}
}

} // namespace

namespace pph {

void occ_bits_dims(int rows, int cols, int& wpr, int& nWordRows)
{
	wpr = (cols + TT - 1) / TT + 2;
	nWordRows = (rows + TT - 1) / TT * TT + 2;
}

hipError_t launch_occ_bits(hipStream_t s, const uint8_t* occ8, int rows, int cols, uint64_t* bits)
{
	int wpr, nWordRows;
	occ_bits_dims(rows, cols, wpr, nWordRows);
	const int64_t words = (int64_t)wpr * nWordRows;
	hipLaunchKernelGGL(k_occ_bits, dim3((unsigned)((words + 3) / 4)), dim3(256), 0, s, occ8, rows, cols, wpr, nWordRows, bits);
	return hipGetLastError();
}

bool wavefront_tiles_supported(int rows, int cols)
{
	// (4096^2, config 5: 765 plans/s with the tile form against 647 with the ordered kernel -- one wave per goal for ~1 s there, four waves per CU because
	// of the 16 KB tile queue: profiles/r04_occupancy_and_config5.txt)
	return tiles_lds_bytes(rows, cols) <= 64 * 1024; // (dynamic LDS of a launch: packs of waves, <= 160 KB)
}

size_t wavefront_tiles_queue_words(int rows, int cols)
{
	// per wave of a launch; 0: the queue stays in LDS.  Opt-in (PP_WF_TILES_QUEUE=global): at 4096^2 -- 16 KB of queue next to 9 KB of tile, four
	// waves per CU with it in LDS, eight without -- the stage is bound by the goals in flight that 1536 slots of 64 MB allow, not by resident
	// waves: 732-748 plans/s with the queue in global memory, 745-765 with it in LDS (profiles/r04_packs_and_ab.txt).
	static const bool global = [] {
		const char* e = getenv("PP_WF_TILES_QUEUE");
		return e && e[0] == 'g';
	}();
	return global ? (tiles_count(rows, cols) + 1) / 2 : 0;
}

int wavefront_tiles_resident_blocks(int rows, int cols)
{
	// waves of the tile form that can be resident at once: packs per CU (occupancy API) x waves per pack x CUs
	int perCu = 0, dev = 0;
	hipDeviceProp_t prop;
	if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
		return 1024;
	const size_t lds = tiles_lds_bytes(rows, cols);
	const int pack = 1; // (an upper bound: single-wave workgroups, launch_tiles_kernel; packs hold fewer)
	const hipError_t e = tile_width() == 64 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_wavefront_tiles<64, false, false>, 64 * pack, lds * pack)
	                                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_wavefront_tiles<32, false, false>, 64 * pack, lds * pack);
	if (e != hipSuccess || perCu < 1)
		perCu = 1;
	return perCu * pack * prop.multiProcessorCount;
}

hipError_t warm_up_wavefront_tiles(hipStream_t s, const MapView& m, int* ctlDev)
{
	TilesArgs A {};
	A.m = m;
	A.ctl = ctlDev; // nGoals = 0: the wave reads the goal counter, finds nothing and leaves
	launch_tiles_kernel(s, 1, A);
	return hipGetLastError();
}

hipError_t launch_wavefront_tiles(hipStream_t s, const MapView& m, int nGoals, const int32_t* goalCellsDev, float* costDev, bool tiledOut, const double* goalPosesDev,
	const double* orderStartsDev, float* orderKeysDev, const WavefrontPublish& pub)
{
	TilesArgs A {};
	A.m = m;
	A.nGoals = nGoals;
	A.goalCells = goalCellsDev;
	A.goalPoses = goalPosesDev;
	A.costOut = costDev;
	A.tiledOut = tiledOut ? 1 : 0;
	A.ctl = pub.tilesCtl;
	A.fbList = pub.tilesFallback;
	A.orderStarts = orderStartsDev;
	A.orderKeys = orderStartsDev ? orderKeysDev : nullptr;
	A.stats = pub.tilesStats;
	A.pub = pub;
	static const int forceEvery = [] {
		const char* e = getenv("PP_WF_TILES_FORCE_FALLBACK"); // test hook (tests/test_gpu_parity.py): every n-th goal goes through the hand-over
		return e ? atoi(e) : 0;
	}();
	A.forceFallbackEvery = forceEvery;
	static int resident = 0;
	static int residentRows = 0, residentCols = 0;
	if (!resident || residentRows != m.rows || residentCols != m.cols) {
		resident = wavefront_tiles_resident_blocks(m.rows, m.cols);
		residentRows = m.rows;
		residentCols = m.cols;
	}
	int grid = nGoals < resident ? nGoals : resident;
	if (const char* e = getenv("PP_WF_TILES_GRID")) { // tuning: waves per launch
		const int g = atoi(e);
		if (g > 0 && g < grid)
			grid = g;
	}
	launch_tiles_kernel(s, grid, A);
	return hipGetLastError();
}

hipError_t launch_order_by_key(hipStream_t s, int n, const float* keysDev, int32_t* orderOutDev)
{
	hipLaunchKernelGGL(k_order_by_key, dim3(1), dim3(512), 0, s, n, keysDev, orderOutDev);
	return hipGetLastError();
}

} // namespace pph
