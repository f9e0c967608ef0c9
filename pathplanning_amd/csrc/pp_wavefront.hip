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
//     a bitonic sort on the 64-bit key (cost bits, ~pushOrder) gives each cell its rank i.
//   * popping cell i pushes its undiscovered neighbours in the fixed enumeration order
//     j = 0..7 (utils/grid.cpp:29-47).  A neighbour n is discovered by the smallest (i, j)
//     that reaches it.  Resolved WITHOUT atomics (scattered global atomics run at a fixed
//     chip-wide rate that 768 concurrent wavefronts saturate): every window cell publishes
//     (round, rank) in a per-cell word with a plain store; a cell p offering itself to n then
//     GATHERS the words of n's other seven neighbours and wins iff none of them is a window
//     cell with a smaller (rank, direction) and an allowed transition.  The winner writes
//     cost = cost_i + edge (same f32 add as the reference) and the push order
//     roundBase + i*8 + j, which is monotone in the reference's push time.
// One workgroup per goal; grids stay in HBM/L2 (cost f32 + rank word u32 per cell), the sort
// runs in LDS.  Algorithmic bytes: 9 B/cell.
#include "pp_internal.hpp"

using namespace ppd;

namespace {

constexpr int WF_T = 512;        // 8 waves: every access here is latency-bound, more lanes = shorter per-thread chains
constexpr int WF_WCAP = 4096;    // LDS window capacity (48 KiB of keys+cells)
constexpr uint32_t kInfBits = 0x7F800000u;

struct WfSlot {
	uint32_t* claim;     // [cells] (round+1) << 17 | rank of the cell in that round's window; 0 = never
	uint32_t* fcell[2];  // open list ping-pong, [fcap]
	uint32_t* fcost[2];
	uint32_t* ford[2];
	uint64_t* gkeys;     // fallback sort buffers, [gcap] (gcap = pow2 >= fcap)
	uint32_t* gvals;
	uint8_t* gmask;      // [gcap]
	uint32_t fcap, gcap;
};

__device__ __forceinline__ WfSlot slot_view(void* base, int64_t bytesPerSlot, int slot, int64_t cells, uint32_t fcap, uint32_t gcap)
{
	char* p = (char*)base + (int64_t)slot * bytesPerSlot;
	WfSlot s;
	s.claim = (uint32_t*)p;
	p += cells * 4;
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
	p += (int64_t)gcap * 4;
	s.gmask = (uint8_t*)p;
	p += (int64_t)gcap;
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

// Bitonic sort of P (power of two) key/value pairs, ascending by key, by the whole block.
// Wave w owns the contiguous chunk of E = P/4 elements [w*E, (w+1)*E).  A stage whose
// partner distance j satisfies 2*j <= E only moves data inside each wave's own chunk, so two
// such consecutive stages need no block barrier between them: a wave executes in lockstep
// and its LDS operations complete in program order.
__device__ __forceinline__ void cmpex(uint64_t* keys, uint32_t* vals, int t, int j, int lj, int k)
{
	// j = 1 << lj: pair t of stride j is (i1, i1 + j) with i1 = (t / j) * 2j + t % j.
	// All four reads are issued before the compare and the writes are unconditional (selects), so a
	// compare-exchange costs one LDS round trip instead of two dependent ones.
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

template <bool kLds>
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
				for (int t = wave * (E / 2) + lane; t < (wave + 1) * (E / 2); t += 64)
					cmpex(keys, vals, t, j, lj, k);
			} else {
				for (int t = tid; t < half; t += WF_T)
					cmpex(keys, vals, t, j, lj, k);
			}
			// the stage that follows (if any)
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

// kProfile: diagnostic build -- per goal {init, min, partition, sort, offer, push, tail} shader-clock sums + rounds, sum(w), sum(P)
enum { WP_INIT = 0, WP_MIN, WP_PART, WP_SORT, WP_OFFER, WP_PUSH, WP_TAIL, WP_ROUNDS, WP_SUMW, WP_SUMP, WP_COUNT };
template <bool kProfile>
__global__ void __launch_bounds__(WF_T) k_wavefront(MapView m, int nGoals, const int32_t* __restrict__ goalCells, float* __restrict__ costOut, void* workspace,
	int64_t bytesPerSlot, uint32_t fcap, uint32_t gcap, int32_t* errorFlag, unsigned long long* __restrict__ prof)
{
	unsigned long long ph[WP_COUNT];
	unsigned long long tl = 0;
#define WF_STAMP(i)                                 \
	if (kProfile) {                                 \
		const unsigned long long now_ = clock64();  \
		ph[i] += now_ - tl;                         \
		tl = now_;                                  \
	}
	__shared__ uint64_t skey[WF_WCAP];
	__shared__ uint32_t sval[WF_WCAP];
	__shared__ uint8_t smask[WF_WCAP];
	__shared__ uint32_t s_min, s_minNext, s_w, s_b, s_new;

	const int tid = threadIdx.x;
	const int64_t cells = (int64_t)m.rows * m.cols;
	const int cols = m.cols, rows = m.rows;
	WfSlot S = slot_view(workspace, bytesPerSlot, blockIdx.x, cells, fcap, gcap);
	const float kDiag = sqrtf(2.0f); // std::sqrt(2.0f), heuristics.cpp:134

	for (int g = blockIdx.x; g < nGoals; g += gridDim.x) {
		if (kProfile) {
			for (int i = 0; i < WP_COUNT; i++)
				ph[i] = 0;
			tl = clock64();
		}
		float* cost = costOut + (int64_t)g * cells;
		uint32_t* costBits = reinterpret_cast<uint32_t*>(cost);
		// heuristics.cpp:108-113: every cell starts at +inf / unexplored
		{
			uint4 inf4 = { kInfBits, kInfBits, kInfBits, kInfBits };
			uint4 ff4 = { 0u, 0u, 0u, 0u };
			const bool aligned = (((uintptr_t)costBits | (uintptr_t)S.claim) & 15) == 0;
			const int64_t n4 = aligned ? cells / 4 : 0;
			for (int64_t i = tid; i < n4; i += WF_T) {
				reinterpret_cast<uint4*>(costBits)[i] = inf4;
				reinterpret_cast<uint4*>(S.claim)[i] = ff4;
			}
			for (int64_t i = n4 * 4 + tid; i < cells; i += WF_T) {
				costBits[i] = kInfBits;
				S.claim[i] = 0u;
			}
		}
		const int32_t start = goalCells[g];
		if (tid == 0) {
			s_min = 0u; // cost bits of the start cell
			s_minNext = 0xFFFFFFFFu;
			s_w = 0;
			s_b = 0;
			s_new = 0;
		}
		__syncthreads();
		if (start < 0)
			continue; // goal outside the map: heuristics.cpp:115-117
		if (tid == 0) {
			cost[start] = 0.0f;
			S.fcell[0][0] = (uint32_t)start;
			S.fcost[0][0] = 0u;
			S.ford[0][0] = 0u;
		}
		__syncthreads();

		WF_STAMP(WP_INIT);
		uint32_t n = 1;          // open-list size
		uint32_t roundBase = 1;  // next push-order value
		uint32_t round = 0;
		int cur = 0;
		bool overflow = false;

		while (n > 0) {
			const int nxt = cur ^ 1;
			// ---- L = smallest cost in the open list (tracked while the list was written in the previous round)
			const float L = __uint_as_float(s_min);
			const uint32_t hiBits = __float_as_uint(L + 1.0f);
			WF_STAMP(WP_MIN);
			// ---- P2: window (cost < fl(L+1)) -> sort buffer; the rest -> next open list
			uint32_t restMin = 0xFFFFFFFFu;
			for (uint32_t i = tid; i < n; i += WF_T) {
				const uint32_t c = S.fcost[cur][i], cell = S.fcell[cur][i], ord = S.ford[cur][i];
				if (c < hiBits) {
					const uint32_t slot = atomicAdd(&s_w, 1u);
					const uint64_t key = ((uint64_t)c << 32) | (uint64_t)(0xFFFFFFFFu - ord);
					if (slot < WF_WCAP) {
						skey[slot] = key;
						sval[slot] = cell;
					} else if (slot < S.gcap) {
						S.gkeys[slot] = key;
						S.gvals[slot] = cell;
					}
				} else {
					const uint32_t slot = atomicAdd(&s_b, 1u);
					S.fcell[nxt][slot] = cell; // slot < n <= fcap
					S.fcost[nxt][slot] = c;
					S.ford[nxt][slot] = ord;
					restMin = min(restMin, c);
				}
			}
			if (restMin != 0xFFFFFFFFu)
				atomicMin(&s_minNext, restMin);
			__syncthreads();
			const uint32_t w = s_w, b = s_b;
			WF_STAMP(WP_PART);
			if (w > S.gcap) {
				overflow = true;
				break;
			}
			const bool lds = w <= WF_WCAP;
			uint64_t* keys = lds ? skey : S.gkeys;
			uint32_t* vals = lds ? sval : S.gvals;
			uint8_t* mask = lds ? smask : S.gmask;
			const uint32_t P = w <= 1 ? 2 : (1u << (32 - __clz((int)(w - 1))));
			if (lds) {
				for (uint32_t i = w + tid; i < P; i += WF_T)
					skey[i] = ~0ull;
				__syncthreads();
				if (w > 1)
					bitonic_sort<true>(skey, sval, (int)P);
			} else {
				for (uint32_t i = tid; i < WF_WCAP; i += WF_T) {
					S.gkeys[i] = skey[i];
					S.gvals[i] = sval[i];
				}
				for (uint32_t i = w + tid; i < P; i += WF_T)
					S.gkeys[i] = ~0ull;
				__syncthreads();
				bitonic_sort<false>(S.gkeys, S.gvals, (int)P);
			}
			WF_STAMP(WP_SORT);
			if (kProfile) {
				ph[WP_ROUNDS]++;
				ph[WP_SUMW] += w;
				ph[WP_SUMP] += P;
			}
			// ---- P3a: publish (round, rank) of every window cell
			if (round + 1u >= (1u << 15) || w > (1u << 17)) {
				overflow = true;
				break;
			}
			const uint32_t roundTag = (round + 1u) << 17;
			for (uint32_t i = tid; i < w; i += WF_T)
				S.claim[vals[i]] = roundTag | i;
			__syncthreads();
			WF_STAMP(WP_OFFER);
			// ---- P3b: every window cell, in pop order i, offers itself to its undiscovered neighbours and
			// pushes the ones it discovers first: cost fixed at discovery (Q3), push order = roundBase + i*8 + j.
			// Reads are issued in batches before use (one memory round trip per batch).
			uint32_t newMin = 0xFFFFFFFFu;
			for (uint32_t i = tid; i < w; i += WF_T) {
				const uint32_t cell = vals[i];
				const int r = (int)(cell / (uint32_t)cols), c = (int)(cell - (uint32_t)r * (uint32_t)cols);
				const float ci = __uint_as_float((uint32_t)(keys[i] >> 32));
				int64_t nidx[8];
				uint8_t occ[8];
				uint32_t cb[8];
				bool inb[8];
#pragma unroll
				for (int j = 0; j < 8; j++) {
					const int nr = r + kDr[j], nc = c + kDc[j];
					inb[j] = nr >= 0 && nr < rows && nc >= 0 && nc < cols;
					nidx[j] = inb[j] ? (int64_t)nr * cols + nc : (int64_t)cell;
				}
#pragma unroll
				for (int j = 0; j < 8; j++) {
					occ[j] = m.occ8[nidx[j]];
					cb[j] = costBits[nidx[j]];
				}
#pragma unroll
				for (int j = 0; j < 8; j++) {
					bool ok = inb[j] && !occ[j]; // IsOccupied(n), heuristics.cpp:128-129
					// diagonal: blocked only if BOTH orthogonal cells (n.row, cell.col) and (cell.row, n.col) are occupied (:130-132);
					// they are neighbours 6/7 (row -/+ 1) and 0/3 (col -/+ 1) of this cell, in bounds whenever the diagonal is
					if (j == 1)
						ok = ok && !(occ[6] && occ[0]);
					if (j == 2)
						ok = ok && !(occ[7] && occ[0]);
					if (j == 4)
						ok = ok && !(occ[6] && occ[3]);
					if (j == 5)
						ok = ok && !(occ[7] && occ[3]);
					ok = ok && cb[j] == kInfBits; // not yet in the open list nor explored
					if (!ok)
						continue;
					// n = neighbour j of this cell.  Is another window cell the first to reach n?
					const int nr = r + kDr[j], nc = c + kDc[j];
					const uint32_t mine = i * 8u + (uint32_t)j;
					uint32_t tag[8];
					uint8_t no[4]; // occupancy of n's orthogonal neighbours: (nr, nc-1), (nr, nc+1), (nr-1, nc), (nr+1, nc)
#pragma unroll
					for (int jj = 0; jj < 8; jj++) {
						// p'' = n - d_jj reaches n through direction jj
						const int pr = nr - kDr[jj], pc = nc - kDc[jj];
						const bool pin = pr >= 0 && pr < rows && pc >= 0 && pc < cols && jj != j;
						tag[jj] = pin ? S.claim[(int64_t)pr * cols + pc] : 0u;
					}
					no[0] = (nc - 1 >= 0) ? m.occ8[(int64_t)nr * cols + (nc - 1)] : (uint8_t)1;
					no[1] = (nc + 1 < cols) ? m.occ8[(int64_t)nr * cols + (nc + 1)] : (uint8_t)1;
					no[2] = (nr - 1 >= 0) ? m.occ8[(int64_t)(nr - 1) * cols + nc] : (uint8_t)1;
					no[3] = (nr + 1 < rows) ? m.occ8[(int64_t)(nr + 1) * cols + nc] : (uint8_t)1;
					bool win = true;
#pragma unroll
					for (int jj = 0; jj < 8; jj++) {
						if ((tag[jj] & 0xFFFE0000u) != roundTag)
							continue; // not popped in this round
						// corner rule for p'' -> n: both (n.row, p''.col) and (p''.row, n.col) occupied blocks a diagonal move
						bool allowed = true;
						if (kDr[jj] != 0 && kDc[jj] != 0) {
							const uint8_t oc = kDc[jj] > 0 ? no[0] : no[1]; // (nr, nc - dc)
							const uint8_t orr = kDr[jj] > 0 ? no[2] : no[3]; // (nr - dr, nc)
							allowed = !(oc && orr);
						}
						const uint32_t other = (tag[jj] & 0x1FFFFu) * 8u + (uint32_t)jj;
						if (allowed && other < mine)
							win = false;
					}
					if (!win)
						continue;
					const float transitionCost = (kDr[j] == 0 || kDc[j] == 0) ? 1.0f : kDiag;
					const float pathCost = transitionCost + ci; // heuristics.cpp:135
					cost[nidx[j]] = pathCost;
					newMin = min(newMin, __float_as_uint(pathCost));
					const uint32_t slot = b + atomicAdd(&s_new, 1u);
					if (slot < S.fcap) {
						S.fcell[nxt][slot] = (uint32_t)nidx[j];
						S.fcost[nxt][slot] = __float_as_uint(pathCost);
						S.ford[nxt][slot] = roundBase + mine;
					}
				}
			}
			if (newMin != 0xFFFFFFFFu)
				atomicMin(&s_minNext, newMin);
			__syncthreads();
			WF_STAMP(WP_PUSH);
			const uint32_t nn = b + s_new;
			const uint32_t nextMin = s_minNext;
			__syncthreads();
			if (tid == 0) {
				s_min = nextMin;
				s_minNext = 0xFFFFFFFFu;
				s_w = 0;
				s_b = 0;
				s_new = 0;
			}
			__syncthreads();
			if (nn > S.fcap) {
				overflow = true;
				break;
			}
			roundBase += w * 8u;
			round++;
			n = nn;
			cur = nxt;
			WF_STAMP(WP_TAIL);
		}
		if (kProfile && tid == 0)
			for (int i = 0; i < WP_COUNT; i++)
				prof[(size_t)g * WP_COUNT + i] = ph[i];
		if (overflow && tid == 0)
			*errorFlag = 1; // open list larger than the workspace
		__syncthreads();
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
	int64_t cells = (int64_t)rows * cols;
	int64_t b = cells * 4 + 6ll * fcap * 4 + (int64_t)gcap * 13 + 16;
	return (b + 255) / 256 * 256;
}

hipError_t launch_wavefront(hipStream_t s, const MapView& m, int nGoals, const int32_t* goalCellsDev, float* costDev, void* workspaceDev,
	int64_t workspaceBytesPerSlot, int nSlots, int32_t* errorFlagDev, unsigned long long* profDev)
{
	if (nGoals <= 0)
		return hipSuccess;
	uint32_t fcap, gcap;
	wf_caps(m.rows, m.cols, fcap, gcap);
	int grid = nGoals < nSlots ? nGoals : nSlots;
	if (profDev)
		hipLaunchKernelGGL(k_wavefront<true>, dim3(grid), dim3(WF_T), 0, s, m, nGoals, goalCellsDev, costDev, workspaceDev, workspaceBytesPerSlot, fcap, gcap, errorFlagDev,
			profDev);
	else
		hipLaunchKernelGGL(k_wavefront<false>, dim3(grid), dim3(WF_T), 0, s, m, nGoals, goalCellsDev, costDev, workspaceDev, workspaceBytesPerSlot, fcap, gcap, errorFlagDev,
			profDev);
	return hipGetLastError();
}

} // namespace pph
