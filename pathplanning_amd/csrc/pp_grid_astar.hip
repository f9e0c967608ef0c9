// grid_astar -- AStarN2 / BidirectionalAStarN2 (algo/a_star_n2.{h,cpp}, algo/a_star.h:326-427,
// algo/bidirectional_a_star.h:10-39,130-196) as a batch on gfx950: SURVEY 8a row a12 / 8f rank 4, BASELINE config 1.
//
// The reference's propagator (a_star_n2.cpp:12-28) walks GridCellPosition::GetNeighbors (utils/grid.cpp:16-50) in its
// fixed order, drops occupied cells and diagonal moves whose two orthogonal cells are BOTH occupied, and asks a user
// function for the transition cost; the heuristic is a user function too.  On the device both are the functions of the
// reference's own script (interfaces/python/scripts/example_a_star_grid.py:46-52): the Euclidean distance between the two
// cells in double precision.  The engine keeps the reference's observable behaviour:
//   * pop order = (totalCost ascending, most recent push first) -- utils/frontier.h:39-48,83-91 (Appendix A Q1);
//   * a state is expanded once; a child whose state is open replaces the open node only when its totalCost is strictly
//     lower (a_star.h:391-402,417-427, Q2); the root counts as explored from the start (a_star.h:361);
//   * bidirectional: forward and reverse searches step alternately, every expanded node is looked up among the states
//     the other side has explored, and the loop stops once fTop.pathCost + rTop.pathCost >= best + offset; the path is
//     forward part + reversed reverse part, so the meeting cell appears twice (Q16).  The two heuristics are the
//     AverageHeuristic pair; the goals of the heuristics it wraps are inputs (AverageHeuristic never forwards SetGoal).
//
// One wave per query, queries handed out dynamically to persistent waves.  Not the reference's data structures:
//   * per cell ONE 16-byte record {pathCost, epoch|flags, parent}: a shortcut overwrites it (the reference leaves the old
//     node behind as a dead leaf, which no output can see); the epoch makes clearing between queries unnecessary;
//   * the open list is a dense UNSORTED array of live entries {totalCost, push sequence, cell} in blocks of 256, the first
//     blocks in LDS and the rest in HBM; every FULL block has its best (totalCost, sequence) in LDS, the tail block has none.
//     A pop lets the tail block's entries and the full blocks' summaries compete in one DPP reduction, fetches the winning
//     block if it is not the tail (4 x 16 bytes per lane, one round trip), moves the list's last entry into the freed slot and,
//     if that slot is in a full block, recomputes the block's summary from registers: a few wave-wide steps, where a sorted
//     vector or a binary heap is a chain of dependent accesses for one lane -- what a GPU is worst at.  A push appends (slots
//     from a ballot prefix, in the reference's neighbour order) and maintains nothing; a shortcut finds the cell's entry by a
//     scan and overwrites it, so no dead entries exist and `empty` / `top` mean what they say.
#include "pp_internal.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

using namespace ppd;

namespace {

constexpr uint32_t kOpen = 1u, kExplored = 2u;
constexpr int GT = 64; // one wave per query

struct CellRec {
	double g;        // pathCost of the state's current node
	uint32_t tag;    // epoch << 2 | kOpen | kExplored; a record of another epoch is "never seen"
	uint32_t parent; // parent cell (row << 16 | col), 0xFFFFFFFF for the root
};
struct OpenEntry {
	double f;     // totalCost
	uint32_t seq; // push sequence number of this search
	uint32_t cell; // row << 16 | col (no division to get the coordinates back)
};
static_assert(sizeof(CellRec) == 16 && sizeof(OpenEntry) == 16, "16-byte records");

struct GridQuery {
	int32_t init[2], goal[2], innerF[2], innerR[2];
};
struct GridOut {
	pp_grid_result r;
	int32_t overflow, pad;
};

struct GridArgs {
	int rows, cols;
	const uint8_t* occ8;
	int nQueries, bidirectional;
	int ldsEntries;      // open-list entries per search kept in LDS (multiple of kBlk)
	uint32_t hbmEntries; // ... and behind them in HBM
	int nBlocks;         // blocks of kBlk entries per search (ldsEntries + hbmEntries, rounded up)
	int maxPath, maxExpanded;
	int64_t slotBytes; // workspace of one resident wave
};

__device__ __forceinline__ uint32_t lin(uint32_t cell, int cols) { return (cell >> 16) * (uint32_t)cols + (cell & 0xFFFFu); }

/// example_a_star_grid.py:46-52 / math.sqrt of an exact integer: one correctly rounded square root
__device__ __forceinline__ double euclid(int r0, int c0, int r1, int c1)
{
	const double dr = (double)(r0 - r1), dc = (double)(c0 - c1);
	return sqrt(dr * dr + dc * dc);
}

/// Orders this wave's memory operations for its own lanes.  Lanes of one wave execute the same instruction stream, and the
/// memory pipeline performs a wave's LDS and vector-memory instructions in issue order, so synchronisation at WAVEFRONT scope
/// needs no instruction at all (the AMDGPU memory model's code sequence for a wavefront-scope fence is empty): only the
/// compiler must keep the order.  No s_waitcnt: draining the stores would add a memory round trip to every expansion.
__device__ __forceinline__ void wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

constexpr int kBlk = 256, kBlkShift = 8, kPer = kBlk / GT; // entries per block of the open list; a lane holds kPer of a block

struct Search {
	CellRec* cells;
	OpenEntry* lds;
	OpenEntry* hbm;
	double* sumF;     // per block of kBlk entries: the block's best entry (totalCost, push sequence), in LDS
	uint32_t* sumSeq;
	uint32_t n, seq;
	// heuristic: mode 0: euclid(s, A); mode 1 (AverageHeuristic): cst + (euclid(s, A) - euclid(s, B)) / 2
	int mode;
	double cst;
	int ar, ac, br, bc;
	int32_t* expandedOut; // (row, col) pairs in expansion order, or null
	int nExpanded;
	uint32_t solution; // cell, 0xFFFFFFFF = none
	uint32_t peekIdx, peekCell; // the entry that pops next when it was looked up and nothing has changed since; else peekIdx = 0xFFFFFFFF
};

__device__ __forceinline__ double heuristic(const Search& s, int r, int c)
{
	const double a = euclid(r, c, s.ar, s.ac);
	if (s.mode == 0)
		return a;
	return s.cst + (a - euclid(r, c, s.br, s.bc)) / 2.0; // bidirectional_a_star.h:23-26
}

__device__ __forceinline__ OpenEntry* entry_ptr(const Search& s, uint32_t i, int ldsEntries)
{
	return i < (uint32_t)ldsEntries ? s.lds + i : s.hbm + (i - (uint32_t)ldsEntries);
}

/// pops before: lower totalCost, then the later push (utils/frontier.h:39-48: insert behind every element >=, pop the back)
__device__ __forceinline__ bool pops_before(double f, uint32_t seq, double g, uint32_t gseq) { return f < g || (f == g && seq > gseq); }

/// One DPP step of a reduction over doubles / uint32: the partner's value, or `ident` where the pattern has no source lane
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_f64(double x, double ident)
{
	const int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(x), kCtrl, kRowMask, 0xF, false);
	const int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(x), kCtrl, kRowMask, 0xF, false);
	return __hiloint2double(hi, lo);
}
template <int kCtrl, int kRowMask>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t x) // identity 0 (max)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, kCtrl, kRowMask, 0xF, false);
}
__device__ __forceinline__ double read_lane_f64(double x, int l)
{
	return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}

/// Best of the lanes' (f, seq) candidates in DPP steps (no LDS round trips: a __shfl is a ds_bpermute, ~100 cycles a step for
/// the three words): the minimum f first (row_shr 1/2/4/8, then row_bcast15/31 across the rows of 16), then the latest push
/// among the lanes that hold it, then the winner's payload by readlane.  kWide = false: only lanes 0..15 take part.
/// Returns false when no lane had a candidate; otherwise the winner's triple comes back on every lane.
template <bool kWide>
__device__ __forceinline__ bool wave_best(double& f, uint32_t& seq, uint32_t& payload, bool has)
{
	unsigned long long hm = __ballot(has);
	if (!kWide)
		hm &= 0xFFFFull;
	if (!hm)
		return false;
	const double kInf = __builtin_huge_val();
	double x = has ? f : kInf;
	x = fmin(x, dpp_f64<0x111, 0xF>(x, kInf)); // row_shr:1
	x = fmin(x, dpp_f64<0x112, 0xF>(x, kInf));
	x = fmin(x, dpp_f64<0x114, 0xF>(x, kInf));
	x = fmin(x, dpp_f64<0x118, 0xF>(x, kInf));
	if (kWide) {
		x = fmin(x, dpp_f64<0x142, 0xA>(x, kInf)); // row_bcast15 -> rows 1, 3
		x = fmin(x, dpp_f64<0x143, 0xC>(x, kInf)); // row_bcast31 -> rows 2, 3
	}
	const double best = read_lane_f64(x, kWide ? 63 : 15);
	const bool tie = has && f == best;
	uint32_t q = tie ? seq : 0u;
	q = max(q, dpp_u32<0x111, 0xF>(q));
	q = max(q, dpp_u32<0x112, 0xF>(q));
	q = max(q, dpp_u32<0x114, 0xF>(q));
	q = max(q, dpp_u32<0x118, 0xF>(q));
	if (kWide) {
		q = max(q, dpp_u32<0x142, 0xA>(q));
		q = max(q, dpp_u32<0x143, 0xC>(q));
	}
	const uint32_t latest = (uint32_t)__builtin_amdgcn_readlane((int)q, kWide ? 63 : 15);
	unsigned long long wm = __ballot(tie && seq == latest);
	if (!kWide)
		wm &= 0xFFFFull;
	const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)wm) - 1);
	f = best;
	seq = latest;
	payload = (uint32_t)__builtin_amdgcn_readlane((int)payload, src);
	return true;
}

/// the block's entries, kPer per lane (entry u of lane l = index block * kBlk + u * 64 + l)
struct BlockRegs {
	OpenEntry e0, e1, e2, e3;
};
__device__ __forceinline__ void load_block(const Search& s, uint32_t block, int ldsEntries, int lane, BlockRegs& B)
{
	const uint32_t base = (block << kBlkShift) + (uint32_t)lane;
	// (no `cond ? *ptr : none`: the compiler turns that into a select of two ADDRESSES and parks `none` in scratch memory)
	B.e0.f = B.e1.f = B.e2.f = B.e3.f = 0.0;
	B.e0.seq = B.e1.seq = B.e2.seq = B.e3.seq = 0u;
	B.e0.cell = B.e1.cell = B.e2.cell = B.e3.cell = 0u;
	if (base < s.n)
		B.e0 = *entry_ptr(s, base, ldsEntries);
	if (base + 64u < s.n)
		B.e1 = *entry_ptr(s, base + 64u, ldsEntries);
	if (base + 128u < s.n)
		B.e2 = *entry_ptr(s, base + 128u, ldsEntries);
	if (base + 192u < s.n)
		B.e3 = *entry_ptr(s, base + 192u, ldsEntries);
}
/// this lane's best among its entries of `block` (indices < n count); slot = u * 64 + lane of the winner
__device__ __forceinline__ bool lane_best(const BlockRegs& B, uint32_t block, uint32_t n, int lane, double& bf, uint32_t& bseq, uint32_t& slot)
{
	const uint32_t base = (block << kBlkShift) + (uint32_t)lane;
	bool has = false;
	if (base < n) {
		bf = B.e0.f;
		bseq = B.e0.seq;
		slot = (uint32_t)lane;
		has = true;
	}
	if (base + 64u < n && (!has || pops_before(B.e1.f, B.e1.seq, bf, bseq))) {
		bf = B.e1.f;
		bseq = B.e1.seq;
		slot = 64u + (uint32_t)lane;
		has = true;
	}
	if (base + 128u < n && (!has || pops_before(B.e2.f, B.e2.seq, bf, bseq))) {
		bf = B.e2.f;
		bseq = B.e2.seq;
		slot = 128u + (uint32_t)lane;
		has = true;
	}
	if (base + 192u < n && (!has || pops_before(B.e3.f, B.e3.seq, bf, bseq))) {
		bf = B.e3.f;
		bseq = B.e3.seq;
		slot = 192u + (uint32_t)lane;
		has = true;
	}
	return has;
}
/// recomputes the summary of a full `block` from its entries in registers
__device__ __forceinline__ void store_summary(Search& s, uint32_t block, uint32_t n, int lane, const BlockRegs& B)
{
	double bf = 0.0;
	uint32_t bseq = 0u, slot = 0u;
	const bool has = lane_best(B, block, n, lane, bf, bseq, slot);
	wave_best<true>(bf, bseq, slot, has);
	if (lane == 0) {
		s.sumF[block] = bf;
		s.sumSeq[block] = bseq;
	}
}

constexpr uint32_t kInTail = 0x80000000u;

/// The entry the reference's Frontier would pop next; s.n > 0.  Blocks below the tail block (n / 256) are full and have a
/// summary in LDS; the tail block -- where pushes land and which removals drain -- has none: its entries are read and compete
/// directly, so neither a push nor a removal has to maintain anything for it.  On return: index of the entry and its cell;
/// B holds the entries of the entry's block.
__device__ __forceinline__ void find_best(const Search& s, int ldsEntries, int lane, BlockRegs& B, uint32_t& idx, uint32_t& cell)
{
	const uint32_t tail = s.n >> kBlkShift;
	load_block(s, tail, ldsEntries, lane, B); // (nothing when n is a multiple of the block size)
	double bf = 0.0;
	uint32_t bseq = 0u, where = 0u;
	bool has = lane_best(B, tail, s.n, lane, bf, bseq, where);
	where |= kInTail;
	for (uint32_t b = (uint32_t)lane; b < tail; b += GT) {
		const double f = s.sumF[b];
		const uint32_t q = s.sumSeq[b];
		if (!has || pops_before(f, q, bf, bseq)) {
			bf = f;
			bseq = q;
			where = b;
			has = true;
		}
	}
	wave_best<true>(bf, bseq, where, has);
	uint32_t block = tail;
	if (!(where & kInTail)) { // the best entry sits in a full block: fetch that block
		block = where;
		load_block(s, block, ldsEntries, lane, B);
	}
	const uint32_t base = (block << kBlkShift) + (uint32_t)lane;
	int mu = -1;
	uint32_t mine = 0u;
	if (base < s.n && B.e0.seq == bseq) {
		mu = 0;
		mine = B.e0.cell;
	}
	if (base + 64u < s.n && B.e1.seq == bseq) {
		mu = 1;
		mine = B.e1.cell;
	}
	if (base + 128u < s.n && B.e2.seq == bseq) {
		mu = 2;
		mine = B.e2.cell;
	}
	if (base + 192u < s.n && B.e3.seq == bseq) {
		mu = 3;
		mine = B.e3.cell;
	}
	const unsigned long long m = __ballot(mu >= 0);
	const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
	cell = (uint32_t)__builtin_amdgcn_readlane((int)mine, src);
	idx = (block << kBlkShift) + (uint32_t)__builtin_amdgcn_readlane(mu, src) * 64u + (uint32_t)src;
}

/// cell of the entry that pops next (BidirectionalAStar's stop rule reads the top nodes' path costs); the following pop reuses it
__device__ __forceinline__ uint32_t peek_best_cell(Search& s, int ldsEntries, int lane)
{
	if (s.peekIdx == 0xFFFFFFFFu) {
		BlockRegs B;
		find_best(s, ldsEntries, lane, B, s.peekIdx, s.peekCell);
	}
	return s.peekCell;
}

/// Frontier::Pop: removes the best entry -- the list's last entry takes its slot -- and, if that slot is in a full block, brings
/// the block's summary up to date.  The state's open flag is left to the caller: every popped node is either the solution or
/// expanded at once, which rewrites the tag.
__device__ __forceinline__ uint32_t pop_best(Search& s, int ldsEntries, int lane)
{
	const uint32_t last = s.n - 1u;
	const OpenEntry moved = *entry_ptr(s, last, ldsEntries); // (every lane reads the same entry; issued with the block loads)
	BlockRegs B;
	uint32_t idx = s.peekIdx, cell = s.peekCell;
	const bool known = idx != 0xFFFFFFFFu; // (uniform) looked up by the stop rule: B is not loaded
	if (!known)
		find_best(s, ldsEntries, lane, B, idx, cell);
	const uint32_t block = idx >> kBlkShift;
	s.n = last;
	s.peekIdx = 0xFFFFFFFFu;
	if (idx != last) {
		const uint32_t slot = idx & (kBlk - 1);
		const bool full = block < (s.n >> kBlkShift); // still a full block (the new tail block needs no summary)
		if ((slot & 63u) == (uint32_t)lane) {
			*entry_ptr(s, idx, ldsEntries) = moved;
			const uint32_t u = slot >> 6;
			if (u == 0)
				B.e0 = moved;
			else if (u == 1)
				B.e1 = moved;
			else if (u == 2)
				B.e2 = moved;
			else
				B.e3 = moved;
		}
		if (full) {
			if (known) {
				wave_sync();
				load_block(s, block, ldsEntries, lane, B);
			}
			store_summary(s, block, s.n, lane, B);
		}
	}
	wave_sync();
	return cell;
}

/// a_star.h:377-409 with a_star_n2.cpp:12-28.  Returns false when the open list ran out of room.
__device__ __forceinline__ bool expand(Search& s, const GridArgs& A, uint32_t epoch, uint32_t cell, int lane)
{
	const int cols = A.cols, rows = A.rows;
	const int r = (int)(cell >> 16), c = (int)(cell & 0xFFFFu);
	const uint32_t cellLin = lin(cell, cols);
	const double g = s.cells[cellLin].g;
	if (lane == 0) {
		s.cells[cellLin].tag = (epoch << 2) | kExplored;
		if (s.expandedOut && s.nExpanded < A.maxExpanded) {
			s.expandedOut[2 * s.nExpanded] = r;
			s.expandedOut[2 * s.nExpanded + 1] = c;
		}
	}
	s.nExpanded++;
	// neighbour j on lane j, in the order of utils/grid.cpp:29-47
	const int dr = (int)((0x8861u >> (2 * (lane & 7))) & 3u) - 1; // {0,-1,1,0,-1,1,-1,1}
	const int dc = (int)((0x5A80u >> (2 * (lane & 7))) & 3u) - 1; // {-1,-1,-1,1,1,1,0,0}
	const int nr = r + dr, nc = c + dc;
	bool push = false, replace = false;
	double pathCost = 0.0, totalCost = 0.0;
	uint32_t ncell = 0u, ncellLin = 0u;
	if (lane < 8 && nr >= 0 && nr < rows && nc >= 0 && nc < cols) {
		ncell = ((uint32_t)nr << 16) | (uint32_t)nc;
		ncellLin = (uint32_t)nr * (uint32_t)cols + (uint32_t)nc;
		// every load of the expansion is issued before the first use: one memory round trip, not three dependent ones
		const uint8_t on = A.occ8[ncellLin];
		const uint8_t oa = A.occ8[(uint32_t)nr * (uint32_t)cols + (uint32_t)c], ob = A.occ8[(uint32_t)r * (uint32_t)cols + (uint32_t)nc]; // in bounds like n itself
		const CellRec rec = s.cells[ncellLin];
		bool ok = on == 0;
		if (dr != 0 && dc != 0) // diagonal: blocked only if both orthogonal cells are occupied (a_star_n2.cpp:21-23)
			ok = ok && !(oa != 0 && ob != 0);
		if (ok) {
			const bool seen = (rec.tag >> 2) == epoch;
			const bool inF = seen && (rec.tag & kOpen), inE = seen && (rec.tag & kExplored);
			pathCost = g + ((dr != 0 && dc != 0) ? sqrt(2.0) : 1.0); // euclid(cell, n): sqrt(1.0) or sqrt(2.0), folded
			const double h = heuristic(s, nr, nc);
			totalCost = pathCost + h;
			if (!inF && !inE)
				push = true;
			else if (inF && rec.g + h > totalCost) // the open node's totalCost is its pathCost + the same h (a_star.h:417-427)
				replace = true;
		}
	}
	const unsigned long long pm = __ballot(push), rm = __ballot(replace);
	const unsigned long long below = (1ull << lane) - 1ull;
	const uint32_t nPush = (uint32_t)__popcll(pm);
	if (s.n + nPush > (uint32_t)A.ldsEntries + A.hbmEntries)
		return false;
	const uint32_t mySeq = s.seq + (uint32_t)__popcll((pm | rm) & below); // NewNode + PushOpen in neighbour order, both kinds
	const uint32_t mySlot = s.n + (uint32_t)__popcll(pm & below);
	if (push) {
		OpenEntry e;
		e.f = totalCost;
		e.seq = mySeq;
		e.cell = ncell;
		*entry_ptr(s, mySlot, A.ldsEntries) = e;
	}
	if (push || replace) {
		CellRec rec;
		rec.g = pathCost;
		rec.tag = (epoch << 2) | kOpen;
		rec.parent = cell;
		s.cells[ncellLin] = rec;
	}
	// shortcuts (rare): the cell's entry is found by a scan and overwritten; its block's summary can only improve
	for (unsigned long long m = rm; m; m &= m - 1ull) {
		const int src = __ffsll((long long)m) - 1;
		const uint32_t target = (uint32_t)__shfl((int)ncell, src, 64);
		const double nf = __shfl(totalCost, src, 64);
		const uint32_t ns = (uint32_t)__shfl((int)mySeq, src, 64);
		for (uint32_t base = 0; base < s.n; base += GT) {
			const uint32_t i = base + (uint32_t)lane;
			if (i < s.n) {
				OpenEntry* p = entry_ptr(s, i, A.ldsEntries);
				if (p->cell == target) {
					p->f = nf;
					p->seq = ns;
					const uint32_t b = i >> kBlkShift;
					if (b < (s.n >> kBlkShift) && pops_before(nf, ns, s.sumF[b], s.sumSeq[b])) { // full blocks only: the tail has no summary
						s.sumF[b] = nf;
						s.sumSeq[b] = ns;
					}
				}
			}
		}
		wave_sync();
	}
	const uint32_t before = s.n;
	if (pm | rm)
		s.peekIdx = 0xFFFFFFFFu;
	s.n += nPush;
	s.seq += (uint32_t)__popcll(pm | rm);
	wave_sync();
	if ((before >> kBlkShift) != (s.n >> kBlkShift)) { // the appends completed a block: from now on it is known by its summary
		BlockRegs B;
		load_block(s, before >> kBlkShift, A.ldsEntries, lane, B);
		store_summary(s, before >> kBlkShift, s.n, lane, B);
		wave_sync();
	}
	return true;
}

__device__ __forceinline__ void init_search(Search& s, uint32_t epoch, uint32_t root, int cols, int lane)
{
	s.n = 1;
	s.seq = 1;
	s.nExpanded = 0;
	s.solution = 0xFFFFFFFFu;
	s.peekIdx = 0xFFFFFFFFu;
	s.peekCell = 0u;
	if (lane == 0) { // a_star.h:350-364: the root is in the frontier AND in the explored map
		CellRec rec;
		rec.g = 0.0;
		rec.tag = (epoch << 2) | kOpen | kExplored;
		rec.parent = 0xFFFFFFFFu;
		s.cells[lin(root, cols)] = rec;
		OpenEntry e;
		e.f = 0.0;
		e.seq = 0u;
		e.cell = root;
		s.lds[0] = e;
	}
}

/// length of the parent chain root .. leaf
__device__ inline int chain_length(const CellRec* cells, uint32_t leaf, int cols)
{
	int n = 0;
	for (uint32_t k = leaf; k != 0xFFFFFFFFu; k = cells[lin(k, cols)].parent)
		n++;
	return n;
}

__global__ void __launch_bounds__(GT) k_grid_astar(GridArgs A, const GridQuery* __restrict__ queries, GridOut* __restrict__ outs, int32_t* __restrict__ paths,
	int32_t* __restrict__ expanded, int32_t* __restrict__ expandedR, char* __restrict__ workspace, int* __restrict__ counter)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	__shared__ int s_q;
	const int lane = threadIdx.x;
	const size_t cells = (size_t)A.rows * A.cols;
	char* ws = workspace + (int64_t)blockIdx.x * A.slotBytes;
	Search f, r;
	f.cells = (CellRec*)ws;
	ws += cells * sizeof(CellRec);
	f.hbm = (OpenEntry*)ws;
	ws += (size_t)A.hbmEntries * sizeof(OpenEntry);
	// LDS per search: [ldsEntries] entries, [nBlocks] block summaries (totalCost), [nBlocks] (push sequence)
	const size_t ldsPerSearch = (size_t)A.ldsEntries * sizeof(OpenEntry) + (size_t)A.nBlocks * 12;
	f.lds = (OpenEntry*)smem;
	f.sumF = (double*)(smem + (size_t)A.ldsEntries * sizeof(OpenEntry));
	f.sumSeq = (uint32_t*)(smem + (size_t)A.ldsEntries * sizeof(OpenEntry) + (size_t)A.nBlocks * 8);
	r = f;
	if (A.bidirectional) {
		r.cells = (CellRec*)ws;
		ws += cells * sizeof(CellRec);
		r.hbm = (OpenEntry*)ws;
		r.lds = (OpenEntry*)(smem + ldsPerSearch);
		r.sumF = (double*)(smem + ldsPerSearch + (size_t)A.ldsEntries * sizeof(OpenEntry));
		r.sumSeq = (uint32_t*)(smem + ldsPerSearch + (size_t)A.ldsEntries * sizeof(OpenEntry) + (size_t)A.nBlocks * 8);
	}
	uint32_t epoch = 0;
	for (;;) {
		if (lane == 0)
			s_q = atomicAdd(counter, 1);
		__syncthreads();
		const int q = s_q;
		__syncthreads();
		if (q >= A.nQueries)
			break;
		epoch++; // the workspace starts zeroed: epoch 0 never matches
		const GridQuery Q = queries[q];
		const uint32_t initCell = ((uint32_t)Q.init[0] << 16) | (uint32_t)Q.init[1];
		const uint32_t goalCell = ((uint32_t)Q.goal[0] << 16) | (uint32_t)Q.goal[1];
		int status = -1, overflow = 0;
		double cost = __builtin_huge_val();
		f.expandedOut = expanded ? expanded + (size_t)q * A.maxExpanded * 2 : nullptr;
		r.expandedOut = expandedR ? expandedR + (size_t)q * A.maxExpanded * 2 : nullptr;
		if (!A.bidirectional) {
			f.mode = 0;
			f.cst = 0.0;
			f.ar = Q.goal[0];
			f.ac = Q.goal[1];
			f.br = f.bc = 0;
			init_search(f, epoch, initCell, A.cols, lane);
			r.nExpanded = 0;
			wave_sync();
			while (f.n > 0) { // a_star.h:326-346
				const uint32_t cell = pop_best(f, A.ldsEntries, lane);
				if (cell == goalCell) {
					f.solution = cell;
					status = 0;
					cost = f.cells[lin(cell, A.cols)].g;
					break;
				}
				if (!expand(f, A, epoch, cell, lane)) {
					overflow = 1;
					break;
				}
			}
		} else {
			// AverageHeuristic pair (bidirectional_a_star.h:10-39,58-63): Hf / Hr = the wrapped heuristics with the goals they hold
			f.mode = r.mode = 1;
			f.ar = Q.innerF[0];
			f.ac = Q.innerF[1];
			f.br = Q.innerR[0];
			f.bc = Q.innerR[1];
			r.ar = Q.innerR[0];
			r.ac = Q.innerR[1];
			r.br = Q.innerF[0];
			r.bc = Q.innerF[1];
			f.cst = euclid(Q.goal[0], Q.goal[1], Q.innerR[0], Q.innerR[1]) / 2.0; // Update: toInit(goal of that direction) / 2
			r.cst = euclid(Q.init[0], Q.init[1], Q.innerF[0], Q.innerF[1]) / 2.0;
			init_search(f, epoch, initCell, A.cols, lane);
			init_search(r, epoch, goalCell, A.cols, lane);
			wave_sync();
			const double offset = heuristic(f, Q.goal[0], Q.goal[1]) + heuristic(r, Q.goal[0], Q.goal[1]); // bidirectional_a_star.h:147
			double best = __builtin_huge_val();
			while (f.n > 0 && r.n > 0) {
				// forward step, then reverse step (bidirectional_a_star.h:150-157, 180-196)
				{
					const uint32_t cell = pop_best(f, A.ldsEntries, lane);
					if (!expand(f, A, epoch, cell, lane)) {
						overflow = 1;
						break;
					}
					const CellRec other = r.cells[lin(cell, A.cols)];
					if ((other.tag >> 2) == epoch && (other.tag & kExplored)) {
						const double through = f.cells[lin(cell, A.cols)].g + other.g;
						if (through < best) {
							best = through;
							f.solution = r.solution = cell;
						}
					}
				}
				{
					const uint32_t cell = pop_best(r, A.ldsEntries, lane);
					if (!expand(r, A, epoch, cell, lane)) {
						overflow = 1;
						break;
					}
					const CellRec other = f.cells[lin(cell, A.cols)];
					if ((other.tag >> 2) == epoch && (other.tag & kExplored)) {
						const double through = r.cells[lin(cell, A.cols)].g + other.g;
						if (through < best) {
							best = through;
							f.solution = r.solution = cell;
						}
					}
				}
				if (f.solution != 0xFFFFFFFFu) {
					bool done = f.n == 0 || r.n == 0;
					if (!done) {
						const uint32_t ft = peek_best_cell(f, A.ldsEntries, lane), rt = peek_best_cell(r, A.ldsEntries, lane);
						done = f.cells[lin(ft, A.cols)].g + r.cells[lin(rt, A.cols)].g >= best + offset;
					}
					if (done) {
						status = 0;
						cost = f.cells[lin(f.solution, A.cols)].g + r.cells[lin(r.solution, A.cols)].g;
						break;
					}
				}
			}
		}
		// ---- results: GetPath (a_star.h:254-269; bidirectional_a_star.h:66-72)
		int nPath = 0;
		if (status == 0 && lane == 0) {
			int32_t* out = paths + (size_t)q * A.maxPath * 2;
			const int lf = chain_length(f.cells, f.solution, A.cols);
			int k = lf - 1;
			for (uint32_t cidx = f.solution; cidx != 0xFFFFFFFFu; cidx = f.cells[lin(cidx, A.cols)].parent, k--)
				if (k < A.maxPath) {
					out[2 * k] = (int32_t)(cidx >> 16);
					out[2 * k + 1] = (int32_t)(cidx & 0xFFFFu);
				}
			nPath = lf;
			if (A.bidirectional) { // the reverse search's chain, walked from the meeting cell, is already in path order
				k = lf;
				for (uint32_t cidx = r.solution; cidx != 0xFFFFFFFFu; cidx = r.cells[lin(cidx, A.cols)].parent, k++)
					if (k < A.maxPath) {
						out[2 * k] = (int32_t)(cidx >> 16);
						out[2 * k + 1] = (int32_t)(cidx & 0xFFFFu);
					}
				nPath = k;
			}
		}
		if (lane == 0) {
			GridOut o;
			o.r.status = overflow ? -2 : status;
			o.r.n_path = nPath;
			o.r.n_expanded = f.nExpanded;
			o.r.n_expanded_reverse = A.bidirectional ? r.nExpanded : 0;
			o.r.cost = cost;
			o.overflow = overflow;
			o.pad = 0;
			outs[q] = o;
		}
		wave_sync();
	}
}

struct Buf {
	void* p = nullptr;
	~Buf()
	{
		if (p)
			(void)hipFree(p);
	}
	hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};

} // namespace

int pp_grid_astar_batch(pp_map* map, int32_t n_queries, const int32_t* init_cells, const int32_t* goal_cells, int32_t bidirectional,
	const int32_t* inner_goals, int32_t max_path, int32_t max_expanded, pp_grid_result* results, int32_t* paths, int32_t* expanded,
	int32_t* expanded_reverse)
{
	using pph::set_error;
	if (!map || !map->ctx || n_queries < 0 || (n_queries > 0 && (!init_cells || !goal_cells || !results)) || max_path < 0 || max_expanded < 0
		|| (max_path > 0 && !paths) || (expanded_reverse && !expanded)) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	if (!map->occ8) {
		set_error("occupancy grid not uploaded (pp_map_upload_occupancy)");
		return PP_ERR_INVALID;
	}
	if (n_queries == 0)
		return PP_OK;
	const int rows = map->desc.rows, cols = map->desc.cols;
	const size_t cells = (size_t)rows * cols;
	if (cells >= (1ull << 30) || rows > 65535 || cols > 65535) {
		set_error("grid too large for packed 16-bit cell coordinates");
		return PP_ERR_INVALID;
	}
	std::vector<GridQuery> hq((size_t)n_queries);
	for (int i = 0; i < n_queries; i++) {
		GridQuery& q = hq[(size_t)i];
		q.init[0] = init_cells[2 * i];
		q.init[1] = init_cells[2 * i + 1];
		q.goal[0] = goal_cells[2 * i];
		q.goal[1] = goal_cells[2 * i + 1];
		// goals held by the wrapped heuristics of the AverageHeuristic pair; default: what a caller means (forward -> goal, reverse -> init)
		q.innerF[0] = inner_goals ? inner_goals[4 * i] : q.goal[0];
		q.innerF[1] = inner_goals ? inner_goals[4 * i + 1] : q.goal[1];
		q.innerR[0] = inner_goals ? inner_goals[4 * i + 2] : q.init[0];
		q.innerR[1] = inner_goals ? inner_goals[4 * i + 3] : q.init[1];
		if (q.init[0] < 0 || q.init[0] >= rows || q.init[1] < 0 || q.init[1] >= cols || q.goal[0] < 0 || q.goal[0] >= rows || q.goal[1] < 0 || q.goal[1] >= cols) {
			set_error("init / goal cell outside the map (the reference indexes its grids with them unchecked)");
			return PP_ERR_INVALID;
		}
	}
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipStream_t s = map->ctx->stream;
	GridArgs A;
	std::memset(&A, 0, sizeof(A));
	A.rows = rows;
	A.cols = cols;
	A.occ8 = map->occ8;
	A.nQueries = n_queries;
	A.bidirectional = bidirectional ? 1 : 0;
	A.maxPath = max_path;
	A.maxExpanded = max_expanded;
	// open list: a front is a few cells per unit of perimeter; the first part lives in LDS
	int lds = 256;
	while (lds < 2 * (rows + cols) && lds < 1024)
		lds <<= 1;
	A.ldsEntries = lds;
	const uint64_t hbmWant = std::min<uint64_t>(cells, 64ull * (uint64_t)(rows + cols) + 4096ull);
	A.hbmEntries = (uint32_t)hbmWant;
	const int dirs = A.bidirectional ? 2 : 1;
	A.nBlocks = (int)(((uint64_t)A.ldsEntries + A.hbmEntries + kBlk - 1) / kBlk);
	A.nBlocks = (A.nBlocks + 3) & ~3; // keeps the 16-byte alignment of what follows in LDS
	A.slotBytes = (int64_t)((cells * sizeof(CellRec) + (size_t)A.hbmEntries * sizeof(OpenEntry)) * dirs + 255) / 256 * 256;
	const size_t ldsBytes = ((size_t)A.ldsEntries * sizeof(OpenEntry) + (size_t)A.nBlocks * 12) * dirs;
	// resident waves: what the LDS lets a CU hold, bounded by a quarter of the free memory
	hipDeviceProp_t prop;
	PP_HIP_TRY(hipGetDeviceProperties(&prop, map->ctx->device));
	int perCu = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_grid_astar, GT, ldsBytes) != hipSuccess || perCu < 1)
		perCu = 1;
	size_t freeB = 0, totalB = 0;
	PP_HIP_TRY(hipMemGetInfo(&freeB, &totalB));
	int64_t slots = (int64_t)perCu * prop.multiProcessorCount;
	slots = std::min<int64_t>(slots, (int64_t)(freeB / 4) / A.slotBytes);
	slots = std::min<int64_t>(slots, n_queries);
	if (slots < 1) {
		set_error("not enough device memory for one grid search workspace");
		return PP_ERR_CAPACITY;
	}
	Buf ws, dq, dout, dpaths, dexp, dexpR, dcnt;
	PP_HIP_TRY(ws.alloc((size_t)slots * (size_t)A.slotBytes));
	PP_HIP_TRY(dq.alloc(hq.size() * sizeof(GridQuery)));
	PP_HIP_TRY(dout.alloc(hq.size() * sizeof(GridOut)));
	PP_HIP_TRY(dpaths.alloc((size_t)n_queries * (size_t)max_path * 8));
	if (expanded)
		PP_HIP_TRY(dexp.alloc((size_t)n_queries * (size_t)max_expanded * 8));
	if (expanded_reverse && A.bidirectional)
		PP_HIP_TRY(dexpR.alloc((size_t)n_queries * (size_t)max_expanded * 8));
	PP_HIP_TRY(dcnt.alloc(16));
	PP_HIP_TRY(hipMemsetAsync(ws.p, 0, (size_t)slots * (size_t)A.slotBytes, s)); // epoch 0 everywhere
	PP_HIP_TRY(hipMemsetAsync(dcnt.p, 0, 16, s));
	PP_HIP_TRY(hipMemcpyAsync(dq.p, hq.data(), hq.size() * sizeof(GridQuery), hipMemcpyHostToDevice, s));
	hipLaunchKernelGGL(k_grid_astar, dim3((unsigned)slots), dim3(GT), ldsBytes, s, A, (const GridQuery*)dq.p, (GridOut*)dout.p, (int32_t*)dpaths.p,
		expanded ? (int32_t*)dexp.p : nullptr, dexpR.p ? (int32_t*)dexpR.p : nullptr, (char*)ws.p, (int*)dcnt.p);
	PP_HIP_TRY(hipGetLastError());
	std::vector<GridOut> ho(hq.size());
	PP_HIP_TRY(hipMemcpyAsync(ho.data(), dout.p, ho.size() * sizeof(GridOut), hipMemcpyDeviceToHost, s));
	if (max_path > 0)
		PP_HIP_TRY(hipMemcpyAsync(paths, dpaths.p, (size_t)n_queries * (size_t)max_path * 8, hipMemcpyDeviceToHost, s));
	if (expanded)
		PP_HIP_TRY(hipMemcpyAsync(expanded, dexp.p, (size_t)n_queries * (size_t)max_expanded * 8, hipMemcpyDeviceToHost, s));
	if (dexpR.p)
		PP_HIP_TRY(hipMemcpyAsync(expanded_reverse, dexpR.p, (size_t)n_queries * (size_t)max_expanded * 8, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipStreamSynchronize(s));
	bool overflow = false;
	for (size_t i = 0; i < ho.size(); i++) {
		results[i] = ho[i].r;
		overflow = overflow || ho[i].overflow;
	}
	if (overflow) {
		set_error("grid A*: an open list exceeded its workspace (status -2 on the queries concerned)");
		return PP_ERR_CAPACITY;
	}
	return PP_OK;
}
