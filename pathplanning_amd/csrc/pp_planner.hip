// hybrid_astar_batch -- HybridAStar::SearchPath graph search (algo/hybrid_a_star.cpp:237-257,
// 59-173; algo/a_star.h:326-427; utils/frontier.h) for a batch of independent queries.
//
// Two kernels share the search state layout below:
//   k_hybrid_search       one wave per query (latency: the plugin's single-query path, profiling, and the
//                         continuation of queries the rows kernel hands over);
//   k_hybrid_search_rows  four queries per wave on a persistent grid (throughput; pp_planner_rows.hpp).
// k_hybrid_search -- one wave (64 lanes) per query, all per-query state in HBM:
//   lanes      = motion primitives of the node being expanded (rollout + IsPathValid +
//                Voronoi cost + heuristic per child), then the 48 Reeds-Shepp words of the
//                analytic expansion;
//   open list  = 64-ary heap, pop is one coalesced 1 KiB load + wave arg-min per level;
//   membership = dense key map over (x cell, y cell, heading bin): unseen / in the open list
//                (node index) / explored -- replaces Frontier::Find + the explored hash set;
//   RNG        = the query's own mt19937_64 (utils/random.h), twist done by the wave.
// Results are bit-identical to the sequential reference as long as libm and OCML agree on
// the discrete outcomes (cells, validity); continuous poses agree to ~1e-15.
#include "pp_search_device.hpp"
#include "pp_row_primitives.hpp"

#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

using namespace ppd;

namespace {

struct Node { // 96 bytes
	double x, y, t;
	double pathCost, totalCost;
	double length;     // arc length after truncation / RS path length
	double h;          // combined heuristic of this pose (reused as the RS gate input when it is expanded)
	double sinT, cosT; // sin / cos of t (reused as the initial-heading terms of the children's arcs)
	int32_t parent;
	uint32_t key;      // packed discrete pose
	int16_t action;    // -1 root, 0..P-1 constant-steer primitive, 1000 + word for Reeds-Shepp
	uint8_t dead;      // removed from the open list by ProcessPossibleShortcut
	uint8_t pad0;
	float dist0;       // obstacle distance at this pose, < 0 when the pose is not a valid state: the first sample of every
	                   // child arc's validity march (IsPathValid samples the start pose first) without a dependent load
	uint8_t pad[8];
};
static_assert(sizeof(Node) == 96, "Node layout");

/// One node of a solution path, written by the search kernel when the query ends (goal first, root last): the node
/// records themselves live in per-slot buffers that the next query of the slot overwrites.
struct PathRec {
	double x, y, t, length;
	int32_t action; // Node::action
	int32_t node;   // node index (matches RsLogEntry::node)
};
static_assert(sizeof(PathRec) == 40, "PathRec layout");

/// A query handed from the four-queries-per-wave kernel to the one-query-per-wave kernel after its first
/// `suspendAfter` expansions: everything that is not already in the slot's node / heap / key-map / engine buffers.
struct SuspendRec {
	int32_t q, slot;
	int32_t heapSize, nNodes, nExpanded, nRngDraws, nRsAttempts, nRsLog, mtIdx;
	uint32_t seq;
	long long stateChecks, pathChecks; // totals so far (arcs + Reeds-Shepp)
	long long bandLo;                  // bottom of the f-band window (slot counts are saved next to the bands)
	int32_t nOutside, pad;             // open-list entries in bands + heap
};

struct RsLogEntry {
	int32_t node, word;
	double t, u, v;
};
constexpr int kRsLogCap = 64;

struct KeySpace {
	int x0, y0, t0; // smallest representable discrete coordinate
	int nx, ny, nt;
	PPD_INLINE bool pack(int ix, int iy, int it, uint32_t& key) const
	{
		const int kx = ix - x0, ky = iy - y0, kt = it - t0;
		if (kx < 0 || kx >= nx || ky < 0 || ky >= ny || kt < 0 || kt >= nt)
			return false;
		key = (uint32_t)((kx * ny + ky) * nt + kt);
		return true;
	}
	__host__ __device__ void unpack(uint32_t key, int& ix, int& iy, int& it) const
	{
		it = (int)(key % (uint32_t)nt) + t0;
		const uint32_t r = key / (uint32_t)nt;
		iy = (int)(r % (uint32_t)ny) + y0;
		ix = (int)(r / (uint32_t)ny) + x0;
	}
	__host__ __device__ size_t size() const { return (size_t)nx * ny * nt; }
};

/// A child whose DiscretizePose quotient lay within 1e-9 cells of a lattice line (pp_device.hpp: near_integer), logged by the
/// one-query-per-wave kernel for pp_planner_certify_lattice: the node it was generated from, the primitive, the (possibly truncated)
/// arc length and the cell the DEVICE put it in.  kind 1: constant-steer child; 2: Reeds-Shepp child (not recomputable on the host)
struct GuardRec {
	int32_t parent;
	int16_t prim, kind;
	double length;
	int32_t ix, iy, it, pad;
};
static_assert(sizeof(GuardRec) == 32, "GuardRec layout");
constexpr int kGuardLogCap = 64; // records kept per query (the count goes on)

struct SearchArgs {
	MapView m;
	pph::RolloutParams rp;
	pph::PrimTable prims;
	HeurView heur;
	KeySpace ks;
	double rmin;
	float rsRev, rsFwd, rsSw; // Reeds-Shepp cost weights as floats (reeds_shepp.cpp:654)
	int maxNodes;
	int maxPath; // PathRec entries per query
	int suspendAfter;  // rows kernel, first pass: expansions after which a query is set aside for the second pass (0 = never)
	int suspendAfter2; // second pass (rows kernel over the set-aside queries): expansions after which the one-query kernel takes over
	int extraSlots;    // buffer slots beyond the rows' own, taken by rows whose query was set aside
	int searchRows;    // rows the planner's buffers were sized for (spare slots start here)
	int listCap;       // capacity of each SuspendRec list: one record per spare slot + one per row
	int rowsWaves;     // rows kernel: waves of this launch (the grid is rounded up to whole workgroups)
	int directCount;   // rows kernel: the first `directCount` queries of the hand-out order are not its own (they run one per wave)
	size_t cells;
	int64_t fieldElems; // floats per query in costFields (8 x 8-tiled obstacle-heuristic field)
	GuardRec* guardLog; // [queries][kGuardLogCap], nullptr: no log (throughput planners)
	int* guardCount;    // [queries]
};

struct DevResult {
	pp_query_result r;
	int32_t solutionNode;
	int32_t nRsLog;
};

#ifndef PP_SEARCH_FIELD_PREFETCH
#define PP_SEARCH_FIELD_PREFETCH 0 // (experiment, measured neutral to harmful) one-query kernel: touch the children's heuristic-field lines
                                   // as soon as the parent pose is known.  The child phase's wait went 10.0 k -> 9.8 k cycles, the added code cost
                                   // 1.4 k: the wait is the slowest of five independent gathers (key map, field, table, distance, path cost)
#endif
#ifndef PP_SEARCH_DIST_WINDOW
#define PP_SEARCH_DIST_WINDOW 0 // (experiment, measured harmful) one-query kernel: LDS window of the distance grid around the expanded
                                // node (pp_device.hpp: DistWindow), filled by LDS-DMA at the pop.  Same results; the marches' distance reads
                                // hit L2 / MALL and were already hidden under the heuristic-field gathers (HBM misses), so the window only
                                // adds its 18 load instructions: 25.7 k -> 29 k cycles per expansion (tools/diag_search.py, PP_SEARCH_ROWS=0)
#endif

constexpr uint32_t kExplored = 0xFFFFFFFFu;
constexpr uint32_t kNoKey = 0xFFFFFFFFu;


// kProfile: diagnostic build only -- accumulates shader-clock cycles per phase into `prof`
// (8 words per query); the product path launches kProfile = false, where no stamp executes.
// One wave per block: LDS operations of a wave complete in program order, so lanes only need the
// compiler not to reorder around the hand-off.  __syncthreads() would also drain every outstanding
// HBM store (s_waitcnt vmcnt(0)), ~1-2k cycles each time.
__device__ __forceinline__ void wave_lds_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}
/// waits until this wave's global stores are visible to its other lanes' loads
__device__ __forceinline__ void wave_vmem_sync()
{
#if PP_WAVE_SYNC_DRAIN
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	__builtin_amdgcn_s_waitcnt(0);
#else
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#endif
	__builtin_amdgcn_wave_barrier();
}

/// GetPath (a_star.h:254-288): walks the parent chain from the solution node; records are written goal first.
/// Returns the number of nodes on the path (it may exceed `cap`: only the first `cap` records are stored).
/// hostPoses (pipeline: pp_pipeline_get_paths): the first hostCap poses also go, as (x, y, theta) triples in the same order, to the query's
/// slot of a ring in pinned host memory -- the plan leaves the GPU with its completion record, no copy and no kernel at fetch time.
__device__ inline int write_path(const Node* nodes, int solutionNode, PathRec* out, int cap, double* hostPoses = nullptr, int hostCap = 0)
{
	int depth = 0;
	for (int k = solutionNode; k >= 0;) {
		const Node nd = nodes[k];
		if (depth < hostCap) {
			hostPoses[3 * depth] = nd.x;
			hostPoses[3 * depth + 1] = nd.y;
			hostPoses[3 * depth + 2] = nd.t;
		}
		if (depth < cap) {
			PathRec pr;
			pr.x = nd.x;
			pr.y = nd.y;
			pr.t = nd.t;
			pr.length = nd.length;
			pr.action = nd.action;
			pr.node = k;
			out[depth] = pr;
		}
		depth++;
		k = nd.parent;
	}
	return depth;
}

__device__ __forceinline__ void guard_log(const SearchArgs& A, int q, int parent, int prim, int kind, double length, int ix, int iy, int it)
{
	if (!A.guardLog)
		return;
	const int pos = atomicAdd(A.guardCount + q, 1);
	if (pos < kGuardLogCap) {
		GuardRec r;
		r.parent = parent;
		r.prim = (int16_t)prim;
		r.kind = (int16_t)kind;
		r.length = length;
		r.ix = ix;
		r.iy = iy;
		r.it = it;
		r.pad = 0;
		A.guardLog[(size_t)q * kGuardLogCap + pos] = r;
	}
}

enum { PH_POP = 0, PH_LOAD, PH_HEUR, PH_CHILD, PH_DUP, PH_INSERT, PH_WRITE, PH_RS, PH_COUNT };
constexpr int kSlots = 65; // staging: one slot per lane + one for the Reeds-Shepp child
constexpr int kRsSlot = 64;

#ifndef PP_SEARCH_WAVES_PER_SIMD
#define PP_SEARCH_WAVES_PER_SIMD 2 // 256 VGPRs: no spills on the per-expansion critical path (128 + 4 batches in flight measured ~10 % faster but needs ~240 GB)
#endif
template <bool kProfile>
__global__ void __launch_bounds__(64, PP_SEARCH_WAVES_PER_SIMD) k_hybrid_search(SearchArgs A, int nQueries, const double* __restrict__ starts, const double* __restrict__ goals,
	const uint64_t* __restrict__ seeds, const float* __restrict__ costFields, Node* __restrict__ nodesBase, HeapEntry* __restrict__ heapBase,
	uint32_t* __restrict__ keymapBase, uint32_t* __restrict__ expandedBase, RsLogEntry* __restrict__ rsLogBase, PathRec* __restrict__ pathBase,
	DevResult* __restrict__ results, unsigned long long* __restrict__ prof, const SuspendRec* __restrict__ resume, const int* __restrict__ nResume,
	const unsigned long long* __restrict__ mtBase, HeapEntry* __restrict__ bandBase, double bandInvW, uint8_t* __restrict__ bandMetaBase,
	const int32_t* __restrict__ queryList, int slotBase)
{
	// Three uses: (a) one block per query of the batch, buffers indexed by the query (resume == nullptr, queryList == nullptr);
	// (b) continuation of the queries the rows kernel suspended: one block per SuspendRec, buffers indexed by its slot;
	// (c) the probable longest queries of a batch the rows kernel works on: block i takes query queryList[i] and buffer slot
	// slotBase + i (nQueries = length of the list).
	if (resume ? ((int)blockIdx.x >= *nResume || (int)blockIdx.x >= A.listCap) : (int)blockIdx.x >= nQueries)
		return;
	const SuspendRec rec = resume ? resume[blockIdx.x] : SuspendRec {};
	const int q = resume ? rec.q : (queryList ? queryList[blockIdx.x] : (int)blockIdx.x);
	const size_t slot = resume ? (size_t)rec.slot : (queryList ? (size_t)slotBase + blockIdx.x : (size_t)q);
	const int lane = threadIdx.x;
#if PP_SEARCH_SETPRIO
	__builtin_amdgcn_s_setprio(3); // see k_hybrid_search_rows
#endif
	unsigned long long phase[PH_COUNT] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	unsigned long long tlast = 0;
	if (kProfile)
		tlast = clock64();
#define PP_STAMP(i)                                \
	if (kProfile) {                                \
		const unsigned long long now_ = clock64(); \
		phase[i] += now_ - tlast;                  \
		tlast = now_;                              \
	}

	__shared__ unsigned long long mt[Mt64::N];
	// staging of the children of the node being expanded; kept until the next expansion so that a
	// child popped right away is read back from LDS instead of HBM
	__shared__ double c_x[kSlots], c_y[kSlots], c_t[kSlots], c_cost[kSlots], c_total[kSlots], c_len[kSlots], c_h[kSlots], c_sin[kSlots], c_cos[kSlots];
	__shared__ uint32_t c_key[kSlots], c_state[kSlots];
	__shared__ float c_d0[kSlots];
	__shared__ uint8_t c_valid[kSlots];
	__shared__ int16_t c_action[kSlots];
	__shared__ int s_rsChecks;
	__shared__ double s_rsPre[24]; // rs::Path::make_prefix of the Reeds-Shepp attempt (23 doubles)
#if PP_SEARCH_DIST_WINDOW
	__shared__ __attribute__((aligned(16))) float s_win[kDistWinElems]; // obstacle distances around the expanded node (pp_device.hpp: DistWindow)
#endif
	__shared__ HeapEntry s_spill[16]; // entries that left the front buffer during this expansion
	__shared__ uint8_t s_bandCnt[kBands]; // f-bands of the open list (pp_search_device.hpp): entries per ring slot

	const MapView& m = A.m;
	const int P = A.prims.n;
	const int maxNodes = A.maxNodes;
	Node* nodes = nodesBase + slot * maxNodes;
	HeapEntry* heap = heapBase + slot * maxNodes;
	HeapEntry* bands = bandBase + slot * (size_t)(kBands * kBandCap);
	uint32_t* keymap = keymapBase + slot * A.ks.size();
	uint32_t* expanded = expandedBase + (size_t)q * maxNodes;
	RsLogEntry* rsLog = rsLogBase + (size_t)q * kRsLogCap;
	const float* field = costFields + (size_t)q * A.fieldElems;

	// goal / start poses go through the Pose2d constructor on the caller's side (theta wrapped)
	const Pose start = { starts[3 * q], starts[3 * q + 1], wrap_theta(starts[3 * q + 2]) };
	const Pose goal = { goals[3 * q], goals[3 * q + 1], wrap_theta(goals[3 * q + 2]) };

	// ---- InitializeSearch, a_star.h:350-364 (a resumed query finds its key map, nodes and heap in the slot)
	if (!resume) {
		const size_t n = A.ks.size();
		const size_t n4 = n / 4;
		if ((((uintptr_t)keymap) & 15) == 0) {
			uint4 z = { 0, 0, 0, 0 };
			for (size_t i = lane; i < n4; i += 64)
				reinterpret_cast<uint4*>(keymap)[i] = z;
			for (size_t i = n4 * 4 + lane; i < n; i += 64)
				keymap[i] = 0;
		} else {
			for (size_t i = lane; i < n; i += 64)
				keymap[i] = 0;
		}
	}
	int myNode = -1; // node index of the child staged in this lane's slot (-1: none / not pushed)
	int rsNode = -1; // same for the Reeds-Shepp slot (wave-uniform)
	// Prefetch of the probable NEXT pop: while a node is expanded, lane k < 24 loads 32-bit word k of the record at the
	// head of the open list.  If that node is popped next (and is not a staged child) its fields come out of these
	// registers with v_readlane instead of a dependent HBM round trip.
	int pfNode = -1;
	uint32_t pfWord = 0u;
	bool pfDead = false; // the prefetched node was replaced (ProcessPossibleShortcut) after it was fetched
	if (!resume) {
		if (lane == 0)
			Mt64::seed(mt, seeds[q]);
	} else {
		for (int i = lane; i < Mt64::N; i += 64)
			mt[i] = mtBase[slot * Mt64::N + i]; // the engine state the rows kernel left in the slot
	}
	FrontLane front;
	front_clear(front);
	int frontCount = 0;
	int heapSize = resume ? rec.heapSize : 0;
	HeapEntry heapTop;
	heapTop.ckey = ~0ull;
	heapTop.nseq = ~0u;
	heapTop.node = 0;
	if (resume && heapSize > 0)
		heapTop = heap[0]; // the whole open list was flushed into the heap when the query was suspended
	// band window: fresh queries start it a little below the root's band; a resumed query brings its window and the
	// slot counts (saved next to the bands when it was set aside)
	uint8_t* const bandMeta = reinterpret_cast<uint8_t*>(bandMetaBase) + slot * (size_t)kBands;
	for (int i = lane; i < kBands; i += 64)
		s_bandCnt[i] = resume ? bandMeta[i] : (uint8_t)0;
	long long bandLo = resume ? rec.bandLo : 0;
	bool bandLoSet = resume != nullptr;
	int nOutside = resume ? rec.nOutside : 0; // open-list entries outside the front buffer (bands + heap + spill buffer)
	// lower bound of everything outside (a resumed query starts with the lowest possible bound: nothing enters the empty
	// front buffer before the first refill)
	unsigned long long lowK = resume ? 0ull : ~0ull;
	unsigned int lowS = resume ? 0u : ~0u;
	int nNodes = resume ? rec.nNodes : 1;
	unsigned int seq = resume ? rec.seq : 1;
	bool startOnBoundary = false;
	if (!resume) {
		double rs_, rc_;
		sincos(start.t, &rs_, &rc_);
		int ix, iy, it;
		startOnBoundary = discretize_pose(start, A.rp.lat, A.rp.headingAlias, ix, iy, it);
		uint32_t key = kNoKey;
		const bool ok = A.ks.pack(ix, iy, it, key);
		if (lane == 0) {
			Node root;
			root.x = start.x;
			root.y = start.y;
			root.t = start.t;
			root.pathCost = 0.0;
			root.totalCost = 0.0;
			root.length = 0.0;
			root.h = combined_heuristic_sc(A.heur, m, field, goal, start, rs_, rc_);
			root.sinT = rs_;
			root.cosT = rc_;
			root.parent = -1;
			root.key = ok ? key : kNoKey;
			root.action = -1;
			root.dead = 0;
			{
				float d0;
				root.dist0 = is_state_valid(m, start.x, start.y, start.t, d0) ? d0 : -1.0f;
			}
			nodes[0] = root;
			if (ok)
				keymap[key] = kExplored; // the root is inserted in the explored set at init (a_star.h:361)
		}
		HeapEntry e;
		e.ckey = cost_key(0.0);
		e.nseq = 0xFFFFFFFFu;
		e.node = 0;
		HeapEntry sp;
		front_insert(front, frontCount, e, lane, sp);
	}
	int mtIdx = resume ? rec.mtIdx : Mt64::N; // engine freshly seeded: first draw twists
	__syncthreads();

	int nExpanded = resume ? rec.nExpanded : 0, nRngDraws = resume ? rec.nRngDraws : 0, nRsAttempts = resume ? rec.nRsAttempts : 0,
		nRsLog = resume ? rec.nRsLog : 0;
	// this lane's arcs; the totals of the suspended part ride in lane 0
	long long laneStateChecks = resume && lane == 0 ? rec.stateChecks : 0, lanePathChecks = resume && lane == 0 ? rec.pathChecks : 0;
	long long rsStateChecks = 0, rsPathChecks = 0;     // wave-uniform (Reeds-Shepp children)
	if (lane == 0)
		lanePathChecks += (long long)startOnBoundary << kGuardShift; // (guard band, pp_device.hpp: the count shares this counter's upper bits)
	int status = -1, solutionNode = -1;
	double solutionCost = __builtin_huge_val();

	// Entries that leave the front buffer are staged in LDS and flushed to the HBM heap in one go:
	// the flush loads all their heap parents in parallel (one memory round trip); only when some
	// entry really has to move up does lane 0 fall back to one-by-one sift-ups.
	int nSpill = 0;
	auto flush_spills = [&]() {
		if (nSpill == 0)
			return;
		__syncthreads();
		// every entry goes to the ring slot of its f-band when that slot is free or already serves the band and has
		// room; else to the heap.  One lane routes them: band counters live in LDS, so nothing here waits for HBM
		// except the occasional heap sift.
		if (lane == 0) {
			int hs = heapSize;
			for (int i = 0; i < nSpill; i++) {
				const HeapEntry e = s_spill[i];
				const long long B = band_of_key(e.ckey, bandInvW);
				const int sl = (int)(B & (kBands - 1));
				const int cnt = s_bandCnt[sl];
				if (B >= bandLo && B < bandLo + kBands && cnt < kBandCap) {
					bands[sl * kBandCap + cnt] = e;
					s_bandCnt[sl] = (uint8_t)(cnt + 1);
					s_spill[i].node = 0xFFFFFFFFu; // marks "not in the heap" for the loop below
				} else {
					heap_push(heap, hs, e);
				}
			}
		}
		__syncthreads();
		for (int i = 0; i < nSpill; i++) {
			const HeapEntry e = s_spill[i];
			if (e.node == 0xFFFFFFFFu)
				continue;
			if (heapSize == 0 || heap_before(e, heapTop))
				heapTop = e;
			heapSize++;
		}
		nSpill = 0;
		__syncthreads();
	};
	auto spill = [&](const HeapEntry& e) {
		if (!bandLoSet) { // the window starts one cost unit below the first entry that leaves the front buffer
			bandLo = (band_of_key(e.ckey, bandInvW) - 64) & ~3ll;
			bandLoSet = true;
		}
		if (lane == 0)
			s_spill[nSpill] = e;
		nSpill++;
		nOutside++;
		if (key_before(e.ckey, e.nseq, lowK, lowS)) {
			lowK = e.ckey;
			lowS = e.nseq;
		}
		if (nSpill == 16)
			flush_spills();
	};
	// An entry joins the front buffer exactly when "front <= everything outside" demands or allows it: it beats the
	// buffer's last entry (then it must; if the buffer is full that last entry leaves), or the buffer has room and the
	// entry beats the lower bound of the outside part.  (Checked against oracle traces by a CPU model of this policy.)
	auto push_open = [&](const HeapEntry& e) {
		bool toFront = true;
		if (frontCount < PP_FRONT_CAP)
			toFront = nOutside == 0 || key_before(e.ckey, e.nseq, lowK, lowS) ||
				(frontCount > 0 && key_before(e.ckey, e.nseq, lane_read64(front.ckey, frontCount - 1), lane_read(front.nseq, frontCount - 1)));
		if (toFront) {
			HeapEntry sp;
			if (front_insert(front, frontCount, e, lane, sp))
				spill(sp);
		} else {
			spill(e);
		}
	};
	// The front buffer ran empty: load the lowest band (all of it: one coalesced load), sort it in the wave, then pull in
	// whatever the heap holds below the buffer's last entry.
	auto refill = [&]() {
		flush_spills();
		// lowest non-empty band: the slots are scanned in ring order from the window's bottom, 64 per step (entries
		// cluster right above the current cost, so the first step nearly always hits)
		long long bAbs = 0x7FFFFFFFFFFFFFFFll;
		for (int step = 0; step < kBands / 64; step++) {
			const long long b = bandLo + step * 64 + lane;
			const unsigned long long hitm = __ballot(s_bandCnt[(int)(b & (kBands - 1))] > 0);
			if (hitm) {
				bAbs = bandLo + step * 64 + (__ffsll((long long)hitm) - 1);
				break;
			}
		}
		long long loadedTop = bandLo - 1; // highest band that has certainly been emptied
		if (bAbs != 0x7FFFFFFFFFFFFFFFll) {
			// the 64 lanes take the aligned group of four consecutive bands that contains the lowest one (4 x 16 entries,
			// contiguous in memory): lane l -> band (group << 2 | l >> 4), entry l & 15
			const long long bn = ((bAbs >> 2) << 2) | (long long)(lane >> 4);
			const int sl = (int)(bn & (kBands - 1));
			const int cntSl = s_bandCnt[sl];
			const bool mineBand = cntSl > 0 && bn >= bandLo && bn < bandLo + kBands;
			const bool have = mineBand && (lane & 15) < cntSl;
#if PP_WAVE_SYNC_DRAIN
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
			__builtin_amdgcn_s_waitcnt(0); // lane 0's band stores
#else
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#endif
			HeapEntry e;
			e.ckey = ~0ull;
			e.nseq = ~0u;
			e.node = 0;
			if (have)
				e = bands[sl * kBandCap + (lane & 15)];
			const int n = __popcll(__ballot(have));
			wave_sort_entries(e.ckey, e.nseq, e.node, lane);
			front.ckey = e.ckey;
			front.nseq = e.nseq;
			front.node = e.node;
			frontCount = n;
			nOutside -= n;
			__syncthreads();
			if (mineBand && (lane & 15) == 0)
				s_bandCnt[sl] = 0;
			__syncthreads();
			bandLo = bAbs & ~3ll;           // every band below the lowest one was empty: the window moves up (multiple of 4)
			loadedTop = ((bAbs >> 2) << 2) | 3; // the group's bands inside the window are empty now
		}
		// (the bound of the outside part is rebuilt from here: spill() lowers it for every entry the loop below pushes out of a full buffer
		// -- those go back into bands at or below `loadedTop` when the spill buffer fills up, where the band-boundary term further down does
		// not see them.  Without this a later, worse entry could enter the buffer ahead of them: found on a 44 597-expansion query of the
		// full-size batch, whose expansion order left the oracle's at expansion 41 196; tests/test_gpu_fullsize.py)
		lowK = ~0ull;
		lowS = ~0u;
		// heap entries that come before the buffer's last entry (or, with an empty buffer, the heap's best) move in
		while (heapSize > 0 && (frontCount == 0 || key_before(heapTop.ckey, heapTop.nseq, lane_read64(front.ckey, frontCount - 1), lane_read(front.nseq, frontCount - 1)))) {
			__syncthreads();
			const HeapEntry he = heap_pop_wave(heap, heapSize, lane, heapTop);
			__syncthreads();
			nOutside--;
			HeapEntry sp;
			if (front_insert(front, frontCount, he, lane, sp))
				spill(sp);
		}
		// lower bound of what is outside now: the heap's best, the start of the first band that was not loaded, the spill
		// buffer
		if (heapSize > 0 && key_before(heapTop.ckey, heapTop.nseq, lowK, lowS)) {
			lowK = heapTop.ckey;
			lowS = heapTop.nseq;
		}
		{
			const unsigned long long bk = cost_key((double)(loadedTop + 1) / bandInvW);
			if (bk < lowK || (bk == lowK && 0u < lowS)) {
				lowK = bk;
				lowS = 0u; // below every entry of that band
			}
		}
		for (int i = 0; i < nSpill; i++) {
			const HeapEntry e = s_spill[i];
			if (key_before(e.ckey, e.nseq, lowK, lowS)) {
				lowK = e.ckey;
				lowS = e.nseq;
			}
		}
	};

	// ---- SearchPath main loop, a_star.h:337-345
	while (frontCount > 0 || nOutside > 0) {
		if (frontCount == 0)
			refill();
		const HeapEntry top = front_pop(front, frontCount, lane); // the front holds the global best entries
		PP_STAMP(PH_POP);
		const int ni = (int)top.node;
		// ---- the popped node: from the staging of the previous expansion when it is one of its children
		double px, py, pt, pPathCost, pH, pSin, pCos;
		uint32_t pKey;
		float pDist0;
		bool pDead = false;
		{
			const unsigned long long hit = __ballot(myNode == ni);
			int slot = hit ? (__ffsll((long long)hit) - 1) : (rsNode == ni ? kRsSlot : -1);
			if (slot >= 0) {
				px = c_x[slot];
				py = c_y[slot];
				pt = c_t[slot];
				pPathCost = c_cost[slot];
				pH = c_h[slot];
				pSin = c_sin[slot];
				pCos = c_cos[slot];
				pKey = c_key[slot];
				pDist0 = c_d0[slot];
			} else if (ni == pfNode) {
				auto dbl = [&](int wi) { return __hiloint2double((int)lane_read(pfWord, wi + 1), (int)lane_read(pfWord, wi)); };
				px = dbl(0);
				py = dbl(2);
				pt = dbl(4);
				pPathCost = dbl(6);
				pH = dbl(12);
				pSin = dbl(14);
				pCos = dbl(16);
				pKey = lane_read(pfWord, 19);
				pDead = pfDead || ((lane_read(pfWord, 20) >> 16) & 0xFFu) != 0u;
				pDist0 = __uint_as_float(lane_read(pfWord, 21));
			} else {
				const Node nd = nodes[ni];
				px = nd.x;
				py = nd.y;
				pt = nd.t;
				pPathCost = nd.pathCost;
				pH = nd.h;
				pSin = nd.sinT;
				pCos = nd.cosT;
				pKey = nd.key;
				pDead = nd.dead != 0;
				pDist0 = nd.dist0;
			}
		}
#if PP_SEARCH_FIELD_PREFETCH
		// The obstacle-heuristic value of a child is a gather from this query's own 4 MB field: an HBM miss every time (17 GB of
		// fields per planner), and the child phase waits for it.  Its cell is known to within float rounding as soon as the parent
		// pose is: touch that cache line now (fast float sin / cos: the value is discarded, only the line matters); the exact look-up
		// two phases later finds it in L2 or on its way.
		float prefetched = 0.0f;
		if (lane < P) {
			const double kap = A.prims.kappa[lane];
			const double d = A.prims.backward[lane] ? -A.rp.arcLength : A.rp.arcLength;
			double ax = px + d * pCos, ay = py + d * pSin;
			if (fabs(kap) > 1e-9) {
				const float tf = (float)(pt + d * kap);
				const double ik = A.prims.invKappa[lane];
				ax = px + ik * ((double)__sinf(tf) - pSin);
				ay = py + ik * (pCos - (double)__cosf(tf));
			}
			int row, col;
			world_to_cell(m, ax, ay, row, col);
			if (inside_map(m, row, col))
				prefetched = field[field_tiled_index(m.cols, row, col)];
		}
#endif
		wave_lds_sync(); // staging is about to be overwritten
		if (pDead)
			continue; // entry of a node replaced by ProcessPossibleShortcut
		const Pose ppose = { px, py, pt };
		if (identical_poses(ppose, goal)) { // IsSolution, hybrid_a_star.h:193-196
			status = 0;
			solutionNode = ni;
			solutionCost = pPathCost;
			break;
		}
#if PP_SEARCH_DIST_WINDOW
		// the children's marches read the distance grid within 1.5 m of this pose: fetch that window into LDS now (LDS-DMA: no
		// registers, one round trip, lands while the heuristic and the child end points are computed)
		DistWindow dwin;
		{
			int prow, pcol;
			world_to_cell(m, ppose.x, ppose.y, prow, pcol);
			prow = min(max(prow, 0), m.rows - 1); // (a popped node is a valid state, i.e. inside the map)
			pcol = min(max(pcol, 0), m.cols - 1);
			const uint32_t winLds = (uint32_t)(uintptr_t)s_win; // low half of the generic address = the LDS byte address
			dwin.win = (LdsFloatPtr)(uintptr_t)winLds;
			dwin.r0 = prow - kDistWinHalf;
			dwin.c0 = pcol - kDistWinHalf;
			// element e = k * 64 + lane of the window is cell (e / kDistWin, e % kDistWin); stepping e by 64 = one row + 31 columns
			int wr = lane / kDistWin, wc = lane - wr * kDistWin;
#pragma unroll 1
			for (int k = 0; k < kDistWinElems / 64; k++) {
				// cells beyond the map edge are never asked for (is_state_valid tests the map first): any in-range address serves
				const int gr = min(max(dwin.r0 + wr, 0), m.rows - 1), gc = min(max(dwin.c0 + wc, 0), m.cols - 1);
				// (the LDS address goes through an integer: the folded generic -> LDS cast of a constant address trips the gfx950 backend,
				// "Illegal instruction: V_CMP_NE_U32 0, src_shared_base")
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(m.dist + (size_t)gr * m.cols + gc),
					(__attribute__((address_space(3))) void*)(uintptr_t)(winLds + (uint32_t)(k * 64 * sizeof(float))), 4, 0, 0);
				wr += 1;
				wc += 64 - kDistWin;
				if (wc >= kDistWin) {
					wc -= kDistWin;
					wr += 1;
				}
			}
		}
#endif
		// ---- Expand, a_star.h:377-409
		if (lane == 0) {
			if (pKey != kNoKey)
				keymap[pKey] = kExplored; // children in the parent's own cell are caught by a key compare below
			expanded[nExpanded] = pKey; // the expansion log holds the packed discrete pose of each expanded node
		}
		rsNode = -1;
		nExpanded++;
		int pix, piy, pit;
		discretize_pose(ppose, A.rp.lat, A.rp.headingAlias, pix, piy, pit);
		PP_STAMP(PH_LOAD);
		// RS gate input (hybrid_a_star.cpp:81): the heuristic of this pose was computed when the node was created
		const double hCost = pH;
		PP_STAMP(PH_HEUR);

		bool capacity = false;
		// ---- constant-steer children, reference order p = 2*deltaIndex + direction (hybrid_a_star.cpp:65-77)
		for (int base = 0; base < P; base += 64) {
			const int p = base + lane;
			bool ok = false;
			uint32_t key = kNoKey, st = 0u;
			Pose child = ppose;
			double cs = pSin, cc = pCos;
			double gcost = 0.0, total = 0.0, len = 0.0, hh = 0.0;
			float d0 = -1.0f;
			if (p < P) {
				ArcSC a;
				a.init = ppose;
				a.sinF = pSin;
				a.cosF = pCos;
				a.kappa = A.prims.kappa[p];
				a.invKappa = A.prims.invKappa[p];
				a.length = A.rp.arcLength;
				a.backward = A.prims.backward[p];
				child = a.interpolate_sc(1.0, cs, cc);
				int ix, iy, it;
				const bool onLine = discretize_pose(child, A.rp.lat, A.rp.headingAlias, ix, iy, it);
				lanePathChecks += (long long)onLine << kGuardShift;
				if (onLine) // logged at once (nothing kept live across the march): if the arc gets truncated this entry is moot, and a
					guard_log(A, q, ni, p, 1, a.length, ix, iy, it); // mismatch on it only sends the query to the CPU reference needlessly
				PP_STAMP(PH_HEUR); // [diagnostic: endpoint]
				// look-ups of the full-length child are issued before the validity march so that their
				// latency overlaps it (they are redone only when the arc gets truncated)
				bool packed = A.ks.pack(ix, iy, it, key);
				if (packed)
					st = keymap[key];
				HeurLoads hl;
				combined_heuristic_issue(A.heur, m, field, goal, child, cs, cc, hl);
				// Voronoi term of the full-length arc: its only map read (the last sample, Q8) is issued with the look-ups
				float voroRaw;
				voronoi_cost_issue(m, a, A.rp.voroDiagRes, voroRaw);
				float lastValidRatio;
				int checks = 0;
				ok = true;
				lanePathChecks++;
				// validity / distance of the child's own pose: the first march sample of ITS children (not a counted check)
				float cd0;
				const bool cIn = is_state_valid_issue(m, child.x, child.y, child.t, cd0);
#if PP_SEARCH_DIST_WINDOW
				__builtin_amdgcn_s_waitcnt(0); // the window has landed (issued a phase ago) -- and so have the look-ups just issued
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				const bool pathValid = is_path_valid_from(m, dwin, a, a.init, pDist0, lastValidRatio, checks);
#else
				const bool pathValid = is_path_valid_from(m, a, a.init, pDist0, lastValidRatio, checks);
#endif
				PP_STAMP(PH_DUP); // [diagnostic: look-up issue + validity march]
				// the values the look-ups above fetched (loaded under the march)
				hh = combined_heuristic_finish(A.heur, hl);
#if PP_SEARCH_FIELD_PREFETCH
				asm volatile("" : : "v"(prefetched)); // keeps the touch above alive; it arrived before the look-up it served
#endif
				const double voroFull = voronoi_cost_finish(voroRaw, A.rp.voroDiagRes, A.rp.voronoiMult);
				d0 = is_state_valid_finish(m, cIn, cd0) ? cd0 : -1.0f;
				if (!pathValid) {
					// PathConstantSteer::Truncate, paths/path_constant_steer.cpp:16-20
					child = a.interpolate_sc((double)lastValidRatio, cs, cc);
					a.length *= (double)lastValidRatio;
					const bool onLineT = discretize_pose(child, A.rp.lat, A.rp.headingAlias, ix, iy, it);
					lanePathChecks += (long long)onLineT << kGuardShift;
					if (onLineT)
						guard_log(A, q, ni, p, 1, a.length, ix, iy, it); // the truncated child is the one that counts
					if (ix == pix && iy == piy && it == pit)
						ok = false;
					else {
						packed = A.ks.pack(ix, iy, it, key);
						if (packed)
							st = keymap[key];
						hh = combined_heuristic_sc(A.heur, m, field, goal, child, cs, cc);
						d0 = is_state_valid(m, child.x, child.y, child.t, cd0) ? cd0 : -1.0f;
					}
				}
				laneStateChecks += checks;
				if (ok) {
					const double pathCost = (a.backward ? A.rp.reverseMult : A.rp.forwardMult) * a.length;
					const double voro = pathValid ? voroFull : voronoi_cost(m, a, A.rp.voroDiagRes, A.rp.voronoiMult);
					const double cost = pathCost + 0.0 + voro; // switching cost is always 0 (hybrid_a_star.cpp:142)
					len = a.length;
					gcost = pPathCost + cost;
					total = gcost + hh; // a_star.h:387-388
					if (!packed)
						ok = false; // outside the key map (cannot happen for poses inside the bounds)
				}
			}
			if (ok && key == pKey)
				st = kExplored; // the parent's cell was marked explored just above (a_star.h:381)
			// open-list node already in this child's cell: its pose / cost (needed by ProcessPossibleShortcut) is fetched
			// by the child's own lane now, all lanes at once, instead of one dependent load per child in the loop below
			double fpx = 0.0, fpy = 0.0, fpt = 0.0, fptot = 0.0;
			uint32_t fpFor = 0u;
			wave_vmem_sync(); // node records written by earlier expansions
			if (ok && st != 0u && st != kExplored) {
				const Node* fn = nodes + (st - 1u);
				fpx = fn->x;
				fpy = fn->y;
				fpt = fn->t;
				fptot = fn->totalCost;
				fpFor = st;
			}
			if (base == 0) {
				// probable next pop (head of the front buffer or of the heap) -> pfWord
				const int cand = frontCount > 0 ? (int)lane_read(front.node, 0) : -1; // the front holds the global best
				pfNode = cand;
				pfDead = false;
				if (cand >= 0 && lane < 24)
					pfWord = reinterpret_cast<const uint32_t*>(nodes + cand)[lane];
			}
			// does an EARLIER valid child of this batch share my cell?  (then my prefetched state may be stale)
			bool dup = false;
			const int cnt = min(64, P - base);
			for (int e = 0; e < cnt; e++) {
				const uint32_t ke = lane_read(key, e);
				const int ve = (int)lane_read(ok ? 1u : 0u, e);
				if (e < lane && ve && ke == key)
					dup = true;
			}
			// staging for the pop that follows (read back from LDS when one of these children is expanded next)
			c_key[lane] = key;
			c_x[lane] = child.x;
			c_y[lane] = child.y;
			c_t[lane] = child.t;
			c_cost[lane] = gcost;
			c_total[lane] = total;
			c_len[lane] = len;
			c_h[lane] = hh;
			c_sin[lane] = cs;
			c_cos[lane] = cc;
			c_d0[lane] = d0;
			myNode = -1;
			wave_lds_sync();
			PP_STAMP(PH_CHILD);
			// ---- insertion in child order, wave-uniform (a_star.h:391-402 + hybrid_a_star.h:199-205);
			// every per-child value is read from its lane's registers (v_readlane), not from memory
			const unsigned long long totalBits = (unsigned long long)__double_as_longlong(total);
			// Most children change nothing (their cell is explored, or holds an open-list node they do not beat): every
			// lane settles that for its own child, and only the children that push, replace, share a cell with an
			// earlier child of the batch or lack the prefetched record walk the serial path below, in child order.
			bool need = false;
			if (lane < cnt && ok) {
				if (dup || st == 0u)
					need = true;
				else if (st != kExplored) {
					if (fpFor == st) {
						const Pose fpp = { fpx, fpy, fpt };
						need = identical_poses(fpp, child) && fptot > total; // ProcessPossibleShortcut would replace it
					} else {
						need = true;
					}
				}
			}
			for (unsigned long long todo = __ballot(need); todo; todo &= todo - 1ull) {
				const int c = __ffsll((long long)todo) - 1;
				const uint32_t ckey = lane_read(key, c);
				uint32_t cst = lane_read(st, c);
				if (lane_read(dup ? 1u : 0u, c)) {
					wave_vmem_sync(); // lane 0's key-map writes of this batch
					cst = keymap[ckey];
				}
				const double ctotal = __longlong_as_double((long long)lane_read64(totalBits, c));
				bool push = false;
				if (cst == 0u) {
					push = true; // !inFrontier && !inExplored
				} else if (cst != kExplored) {
					// in the open list: replace only if the poses are identical and the new path is strictly cheaper
					const int fi = (int)cst - 1;
					const unsigned long long hitf = __ballot(myNode == fi);
					Pose fp;
					double ftotal;
					if (hitf) {
						const int fs = __ffsll((long long)hitf) - 1;
						fp = { c_x[fs], c_y[fs], c_t[fs] };
						ftotal = c_total[fs];
					} else if (lane_read(fpFor, c) == cst) {
						fp.x = __longlong_as_double((long long)lane_read64((unsigned long long)__double_as_longlong(fpx), c));
						fp.y = __longlong_as_double((long long)lane_read64((unsigned long long)__double_as_longlong(fpy), c));
						fp.t = __longlong_as_double((long long)lane_read64((unsigned long long)__double_as_longlong(fpt), c));
						ftotal = __longlong_as_double((long long)lane_read64((unsigned long long)__double_as_longlong(fptot), c));
					} else {
						wave_vmem_sync();
						const Node fn = nodes[fi];
						fp = { fn.x, fn.y, fn.t };
						ftotal = fn.totalCost;
					}
					const Pose cp = { c_x[c], c_y[c], c_t[c] };
					if (identical_poses(fp, cp) && ftotal > ctotal) {
						if (fi == pfNode)
							pfDead = true;
						if (lane == 0)
							nodes[fi].dead = 1;
						if (myNode == fi)
							myNode = -1; // its staged copy must not be used any more
						push = true;
					}
				}
				if (push) {
					if (nNodes >= maxNodes) {
						capacity = true;
						break;
					}
					const int idx = nNodes++;
					if (lane == c)
						myNode = idx;
					if (lane == 0)
						keymap[ckey] = (uint32_t)idx + 1u;
					HeapEntry e;
					e.ckey = cost_key(ctotal);
					e.nseq = 0xFFFFFFFFu - seq;
					seq++;
					e.node = (uint32_t)idx;
					push_open(e);
				}
			}
			PP_STAMP(PH_INSERT);
			// ---- every lane writes the node record of its own child
			if (myNode >= 0) {
				Node nd;
				nd.x = child.x;
				nd.y = child.y;
				nd.t = child.t;
				nd.pathCost = gcost;
				nd.totalCost = total;
				nd.length = len;
				nd.h = hh;
				nd.sinT = cs;
				nd.cosT = cc;
				nd.parent = ni;
				nd.key = key;
				nd.action = (int16_t)p;
				nd.dead = 0;
				nd.dist0 = d0;
				nodes[myNode] = nd;
			}
			PP_STAMP(PH_WRITE);
			if (capacity)
				break;
		}
		if (capacity) {
			status = -4;
			break;
		}

		// ---- Reeds-Shepp analytic expansion, gated (hybrid_a_star.cpp:81-88): the RNG is drawn
		// only when hCost >= 10 (short-circuit ||)
		bool tryRs = hCost < 10.0;
		if (!tryRs) {
			if (mtIdx >= Mt64::N) {
				Mt64::twist_wave(mt, lane);
				mtIdx = 0;
			}
			const double u = Mt64::uniform01(Mt64::temper(mt[mtIdx]));
			mtIdx++;
			nRngDraws++;
			tryRs = u < 10.0 / (hCost * hCost);
		}
		if (tryRs) {
			nRsAttempts++;
			// GetOptimalPath (reeds_shepp.cpp:654-683): lane w evaluates word w
			Pose rel;
			{
				// goal - start (geometry/2dplane.h:65-79) with the stored sin/cos of the node's heading
				const double dx = goal.x - ppose.x, dy = goal.y - ppose.y;
				const double s = -pSin, c = pCos;
				rel.x = c * dx + (-s) * dy;
				rel.y = s * dx + c * dy;
				rel.t = wrap_theta(wrap_theta(goal.t - ppose.t));
			}
			rel.x = rel.x / A.rmin;
			rel.y = rel.y / A.rmin;
			float wcost = __builtin_huge_valf();
			double wt = 0, wu = 0, wv = 0;
			bool wvalid = false;
			if (lane < rs::kNumWords) {
				double gx, gy, gt;
				rs::goal_variant(rel, lane % 4, gx, gy, gt);
				const double length = rs::base_lengths(lane / 4, gx, gy, gt, wt, wu, wv);
				if (!(length == rs::inf())) {
					rs::Segment sg;
					rs::word_segment(lane, wt, wu, wv, sg);
					wcost = rs::compute_cost(sg, A.rmin, A.rsRev, A.rsFwd, A.rsSw);
					wvalid = wcost < __builtin_huge_valf(); // NaN and +inf never win a `cost < optimalCost` test
				}
			}
			// first strictly-lowest float cost in word order
			float best = wvalid ? wcost : __builtin_huge_valf();
#pragma unroll
			for (int off = 32; off > 0; off >>= 1)
				best = fminf(best, __shfl_xor(best, off, 64));
			const unsigned long long match = __ballot(wvalid && wcost == best);
			const int word = match ? (__ffsll((long long)match) - 1) : -1;
			if (word >= 0) {
				const double bt = __shfl(wt, word, 64), bu = __shfl(wu, word, 64), bv = __shfl(wv, word, 64);
				// the winner's path is validated by one lane (the adaptive march is sequential)
				if (lane == 0) {
					rs::Path path;
					path.init = ppose;
					rs::word_segment(word, bt, bu, bv, path.seg);
					path.rmin = A.rmin;
					path.length = path.seg.length * A.rmin; // PathSegment::GetLength
					float lastRatio;
					int checks = 0;
					path.make_prefix(s_rsPre); // (see k_hybrid_search_rows)
					const rs::PrefixedPath ppath = { path, s_rsPre, path.length };
					const bool valid = is_path_valid(m, ppath, path.init, lastRatio, checks);
					c_valid[kRsSlot] = 0;
					s_rsChecks = checks;
					if (valid) {
						const double pathAndSwitchingCosts = (double)rs::compute_cost(path.seg, A.rmin, A.rsRev, A.rsFwd, A.rsSw); // PathReedsShepp::ComputeCost
						const Pose child = ppath.interpolate(1.0);
						int ix, iy, it;
						const bool onLineR = discretize_pose(child, A.rp.lat, A.rp.headingAlias, ix, iy, it);
						lanePathChecks += (long long)onLineR << kGuardShift;
						if (onLineR)
							guard_log(A, q, ni, word, 2, path.length, ix, iy, it);
						const double voro = voronoi_cost(m, ppath, A.rp.voroDiagRes, A.rp.voronoiMult);
						const double cost = pathAndSwitchingCosts + voro;
						uint32_t key;
						if (A.ks.pack(ix, iy, it, key)) {
							double s_, c_;
							sincos(child.t, &s_, &c_);
							const double hh = combined_heuristic_sc(A.heur, m, field, goal, child, s_, c_);
							c_valid[kRsSlot] = 1;
							c_key[kRsSlot] = key;
							c_x[kRsSlot] = child.x;
							c_y[kRsSlot] = child.y;
							c_t[kRsSlot] = child.t;
							c_cost[kRsSlot] = pPathCost + cost;
							c_total[kRsSlot] = (pPathCost + cost) + hh;
							c_len[kRsSlot] = path.length;
							c_h[kRsSlot] = hh;
							c_sin[kRsSlot] = s_;
							c_cos[kRsSlot] = c_;
							{
								float rd0;
								c_d0[kRsSlot] = is_state_valid(m, child.x, child.y, child.t, rd0) ? rd0 : -1.0f;
							}
							c_state[kRsSlot] = keymap[key];
							c_action[kRsSlot] = (int16_t)(1000 + word);
						}
					}
				}
				wave_lds_sync();
				rsPathChecks++;
				rsStateChecks += (long long)s_rsChecks;
				if (c_valid[kRsSlot]) {
					const uint32_t ckey = c_key[kRsSlot];
					const uint32_t cst = c_state[kRsSlot];
					bool push = false;
					if (cst == 0u)
						push = true;
					else if (cst != kExplored) {
						const int fi = (int)cst - 1;
						const unsigned long long hitf = __ballot(myNode == fi);
						Pose fp;
						double ftotal;
						if (hitf) {
							const int fs = __ffsll((long long)hitf) - 1;
							fp = { c_x[fs], c_y[fs], c_t[fs] };
							ftotal = c_total[fs];
						} else {
							wave_vmem_sync();
							const Node fn = nodes[fi];
							fp = { fn.x, fn.y, fn.t };
							ftotal = fn.totalCost;
						}
						const Pose cp = { c_x[kRsSlot], c_y[kRsSlot], c_t[kRsSlot] };
						if (identical_poses(fp, cp) && ftotal > c_total[kRsSlot]) {
							if (fi == pfNode)
								pfDead = true;
							if (lane == 0)
								nodes[fi].dead = 1;
							if (myNode == fi)
								myNode = -1;
							push = true;
						}
					}
					if (push) {
						if (nNodes >= maxNodes) {
							status = -4;
							break;
						}
						const int idx = nNodes++;
						if (lane == 0) {
							Node nd;
							nd.x = c_x[kRsSlot];
							nd.y = c_y[kRsSlot];
							nd.t = c_t[kRsSlot];
							nd.pathCost = c_cost[kRsSlot];
							nd.totalCost = c_total[kRsSlot];
							nd.length = c_len[kRsSlot];
							nd.h = c_h[kRsSlot];
							nd.sinT = c_sin[kRsSlot];
							nd.cosT = c_cos[kRsSlot];
							nd.parent = ni;
							nd.key = ckey;
							nd.action = c_action[kRsSlot];
							nd.dead = 0;
							nd.dist0 = c_d0[kRsSlot];
							nodes[idx] = nd;
							keymap[ckey] = (uint32_t)idx + 1u;
							if (nRsLog < kRsLogCap) {
								RsLogEntry le;
								le.node = idx;
								le.word = word;
								le.t = bt;
								le.u = bu;
								le.v = bv;
								rsLog[nRsLog] = le;
							}
						}
						rsNode = idx;
						nRsLog++;
						HeapEntry e;
						e.ckey = cost_key(c_total[kRsSlot]);
						e.nseq = 0xFFFFFFFFu - seq;
						seq++;
						e.node = (uint32_t)idx;
						push_open(e);
					}
				}
				wave_lds_sync();
			}
		}
		PP_STAMP(PH_RS);
	}
	if (kProfile && lane == 0)
		for (int i = 0; i < PH_COUNT; i++)
			prof[(size_t)q * PH_COUNT + i] = phase[i];
#undef PP_STAMP

	// lane-local counters -> totals
	for (int off = 32; off > 0; off >>= 1) {
		laneStateChecks += __shfl_xor(laneStateChecks, off, 64);
		lanePathChecks += __shfl_xor(lanePathChecks, off, 64);
	}
	const long long nStateChecks = laneStateChecks + rsStateChecks;
	const long long pathChecksPacked = lanePathChecks + rsPathChecks;
	const long long pathChecks = pathChecksPacked & kGuardMask;
	__syncthreads();
	if (lane == 0) {
		DevResult r;
		r.r.n_lattice_boundary_hits = (int32_t)(pathChecksPacked >> kGuardShift);
		r.r.reserved = 0;
		r.r.status = status;
		r.r.n_expanded = nExpanded;
		r.r.n_nodes = nNodes;
		r.r.n_path = 0;
		if (status == 0)
			r.r.n_path = write_path(nodes, solutionNode, pathBase + (size_t)q * A.maxPath, A.maxPath);
		r.r.cost = solutionCost;
		r.r.n_rng_draws = nRngDraws;
		r.r.n_rs_attempts = nRsAttempts;
		r.r.n_state_checks = nStateChecks;
		r.r.n_path_checks = pathChecks;
		r.solutionNode = solutionNode;
		r.nRsLog = nRsLog < kRsLogCap ? nRsLog : kRsLogCap;
		results[q] = r;
	}
}

// ---- streaming pipeline (pp_pipeline.hpp): what the persistent search grid shares with the wavefront kernel and the host ----
/// device memory, zeroed at creation
struct PipeCtl {
	unsigned long long readyTail;  // field slots appended to the ready ring by the wavefront kernel (absolute count)
	unsigned long long readyHead;  // entries claimed by search rows
	unsigned long long doneTail;   // completion records reserved
	unsigned long long nSubmitted; // queries handed to the wavefront kernel so far (written by the host, in stream order before a top-up launch)
	int stop;                      // host: leave as soon as the rows are idle
	int pad;
	unsigned long long quiesce;    // host: every result of the first `quiesce` submitted queries has been polled -- idle waves need not wait for more
	unsigned long long urgentTail; // urgent ring (WavefrontPublish::urgent): entries appended by k_pipe_scatter
	unsigned long long urgentHead; // entries claimed by wavefront workgroups
};
/// completion record in PINNED HOST memory: the row writes the result, then the stamp ((position + 1) << 32 | slot); the host consumes
/// records in position order as their stamps appear
struct PipeDone {
	unsigned long long stamp;
	DevResult r;
	unsigned long long readyTail, readyHead; // the queue counters as the announcing row saw them (pp_pipeline_backlog)
};
struct PipeView {
	PipeCtl* ctl = nullptr; // nullptr: the kernel works on a batch (no pipeline)
	unsigned long long* ready = nullptr;
	unsigned long long readyMask = 0;
	PipeDone* done = nullptr;
	unsigned long long doneMask = 0;
	int* waveAlive = nullptr;         // [waves] 1 while a wave of some launch owns that wave index (and with it the rows' buffers)
	unsigned long long lingerTicks = 0; // loop passes an idle wave stays although every submitted query has been claimed: the next submission is usually on its way
	double* pathHost = nullptr;       // pinned host memory, [capacity][pathHostCap][3]: the solution path's poses, goal first (write_path)
	int pathHostCap = 0;
	unsigned long long idleTicks = 0; // loop passes (~4 us each: a sleep and three polls) a wave waits without work before it leaves on its own
	int soloAfter = 0;                // > 0: a wave one of whose rows has passed this many expansions takes no new queries while the ready ring holds fewer than soloBacklog
	int soloBacklog = 0;
	int boostAfter = 0;               // > 0: a wave one of whose rows has passed this many expansions runs at issue priority 3 (the run's longest chains share their SIMDs with throughput work)
};

#include "pp_planner_rows.hpp"
#include "pp_postprocess.hpp"

} // namespace

// ---------------------------------------------------------------------------
struct pp_planner {
	pp_map* map = nullptr;
	pp_hybrid_params params {};
	int maxBatch = 0, maxNodes = 0;
	SearchArgs args {};
	pph::NonHoloDesc nh {};
	std::vector<double> deltas;
	// device
	double* table = nullptr;
	bool tableReady = false;
	float* costFields = nullptr;
	void* wfWorkspace = nullptr;
	int64_t wfBytesPerSlot = 0;
	int wfSlots = 0;
	int32_t* wfError = nullptr;
	int* tilesCtl = nullptr;          // control words of the tile form of the wavefront (pp_wavefront_tiles.hip; zero at allocation, set back by its kernels)
	int32_t* tilesFallback = nullptr; // [maxBatch] goals it hands to the ordered kernel
	Node* nodes = nullptr;
	HeapEntry* heaps = nullptr;
	uint32_t* keymaps = nullptr;
	uint32_t* expanded = nullptr;
	RsLogEntry* rsLogs = nullptr;
	DevResult* results = nullptr;
	unsigned long long* prof = nullptr; // diagnostic phase cycles, [maxBatch][PH_COUNT]
	bool profile = false;
	unsigned long long* mtStates = nullptr; // [searchRows][312] mt19937_64 engine state per row (rows kernel)
	int* nextQuery = nullptr;               // = wfError + 2: {query counter of the persistent rows kernel, spare slots handed out}
	SuspendRec* suspended = nullptr;        // [2][extraSlots] queries set aside by the first / second pass of the rows kernel
	int32_t* order = nullptr;               // [maxBatch] query indices, probable longest first (rows kernel)
	float* orderKeys = nullptr;             // [maxBatch] field value at each query's start pose (the sort key)
	HeapEntry* bands = nullptr;             // [slots][kBands * kBandCap] f-bands of the open list
	uint8_t* bandMeta = nullptr;            // [slots][kBands] slot fill counts of set-aside queries
	double bandInvW = 64.0;                 // bands are 1 / bandInvW wide in total cost
	int searchWaves = 0;                    // resident waves of k_hybrid_search_rows on this device
	int searchRows = 0;                     // rows (= search buffer slots) this planner runs with
	bool rowsKernel = false;                // four-queries-per-wave kernel (throughput) vs one query per wave (latency)
	int compactBelow = 0;                   // first pass: a wave with an empty queue and <= this many busy rows re-queues them
	GuardRec* guardLog = nullptr;           // [maxBatch][kGuardLogCap] lattice-line children (one-query-per-wave planners only)
	int* guardCount = nullptr;              // [maxBatch]
	PathRec* paths = nullptr;               // [maxBatch][maxPath] solution paths, goal first
	int maxPath = 0;
	double *dStarts = nullptr, *dGoals = nullptr;
	uint64_t* dSeeds = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
	hipEvent_t startAfter = nullptr; // one-shot: the next batch waits for this event (pp_planner_start_after_fields_of)
	// the probable longest queries of a batch run one per wave next to the rows kernel (PP_SEARCH_DIRECT), on their own stream
	hipStream_t directStream = nullptr;
	hipEvent_t e3 = nullptr;
	int directCount = 0;
	float wavefrontMs = 0, searchMs = 0;
	int lastBatch = 0;
	pp_pipeline* owner = nullptr; // (set with pipelineOwned)
	bool pipelineOwned = false; // the buffer set of a pp_pipeline (pp_pipeline_planner()): its rows and field slots belong to the pipeline's kernels
	std::vector<DevResult> hostResults;
	// post-processing (pp_postprocess.hpp), allocated at the first pp_planner_postprocess
	PostBuffers post {};
	int postMaxPoints = 0, postDone = 0;
	std::vector<pp_post_result> hostPost;
};

namespace {

using pph::set_error;

// Scratch the planner's kernels need per lane (largest private segment among them; tests/test_kernel_resources.py keeps
// the figure honest against the built code object) and the number of hardware queues whose first dispatch may still have to
// allocate it after this planner took its memory (bench.py runs with GPU_MAX_HW_QUEUES=16).
constexpr size_t kMaxPrivateBytes = 1024;
constexpr size_t kReserveQueues = 16;

/// Empty dispatches of the three kernels a batch launches, on the planner's stream, then a synchronisation: the queue
/// allocates their scratch here, where a failure is an error code, not at the first batch, where it is an abort.
int warm_up_kernels(pp_planner* p, pp_map* map)
{
	hipStream_t s = map->ctx->stream;
	int32_t* ctl = nullptr;
	PP_HIP_TRY(hipMalloc((void**)&ctl, 64));
	hipError_t e = hipMemsetAsync(ctl, 0, 64, s);
	if (e == hipSuccess)
		e = pph::warm_up_wavefront(s, map->view(), ctl);
	if (e == hipSuccess && map->occBits && pph::wavefront_tiles_supported(map->desc.rows, map->desc.cols))
		e = pph::warm_up_wavefront_tiles(s, map->view(), (int*)ctl + 8);
	if (e == hipSuccess) {
		SearchArgs none = p->args;
		none.rowsWaves = 0; // every wave of the rows kernel leaves at once
		none.listCap = 0;
		hipLaunchKernelGGL(k_hybrid_search_rows<false>, dim3(1), dim3(64 * PP_ROWS_WAVES_PER_WG), 0, s, none, 0, (const double*)nullptr, (const double*)nullptr, (const uint64_t*)nullptr,
			(const float*)nullptr, (Node*)nullptr, (HeapEntry*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (RsLogEntry*)nullptr, (PathRec*)nullptr, (unsigned long long*)nullptr,
			(DevResult*)nullptr, (int*)nullptr, (SuspendRec*)nullptr, (const int32_t*)nullptr, 0, (const SuspendRec*)nullptr, (const int*)nullptr, (int*)nullptr, (int*)nullptr, 0,
			(HeapEntry*)nullptr, 0.0, (uint8_t*)nullptr, PipeView {});
		e = hipGetLastError();
	}
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_hybrid_search<false>, dim3(1), dim3(64), 0, s, p->args, 0, (const double*)nullptr, (const double*)nullptr, (const uint64_t*)nullptr, (const float*)nullptr,
			(Node*)nullptr, (HeapEntry*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (RsLogEntry*)nullptr, (PathRec*)nullptr, (DevResult*)nullptr, (unsigned long long*)nullptr,
			(const SuspendRec*)nullptr, (const int*)nullptr, (const unsigned long long*)nullptr, (HeapEntry*)nullptr, 0.0, (uint8_t*)nullptr, (const int32_t*)nullptr, 0);
		e = hipGetLastError();
	}
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_postprocess, dim3(1), dim3(kPostThreads), 64, s, p->args, PostParams {}, 0, (const PathRec*)nullptr, (const RsLogEntry*)nullptr, (const DevResult*)nullptr,
			(const uint32_t*)nullptr, (const uint32_t*)nullptr, PostBuffers {});
		e = hipGetLastError();
	}
	if (e == hipSuccess)
		e = hipStreamSynchronize(s);
	(void)hipFree(ctl);
	if (e != hipSuccess)
		return pph::hip_fail(e, "planner kernel warm-up (scratch allocation)");
	return PP_OK;
}

/// The primitive table of the search kernels from the steering-angle list (StatePropagator::m_deltas): children in list order,
/// forward then backward each (hybrid_a_star.cpp:65-77); curvature per primitive with the host's libm
int primitives_from_deltas(pp_planner* p)
{
	const int P = 2 * (int)p->deltas.size();
	if (P < 2 || P > pph::kMaxPrimitives) {
		set_error("between 1 and " + std::to_string(pph::kMaxPrimitives / 2) + " steering angles (two motion primitives each)");
		return PP_ERR_INVALID;
	}
	const double wheelbase = p->params.wheelbase, rearToCenter = 0.0;
	pph::PrimTable& T = p->args.prims;
	T.n = P;
	for (size_t d = 0; d < p->deltas.size(); d++) {
		// ConstantSteer, kinematic_bicycle_model.cpp:13-17 with rearToCenter = 0
		const double tanSteering = std::tan(p->deltas[d]);
		const double beta = std::atan(rearToCenter * tanSteering / wheelbase);
		const double cosBeta = std::cos(beta);
		const double DthetaDdist = cosBeta * tanSteering / wheelbase;
		T.kappa[2 * d] = DthetaDdist;
		T.invKappa[2 * d] = T.invKappa[2 * d + 1] = DthetaDdist != 0.0 ? 1 / DthetaDdist : 0.0;
		T.backward[2 * d] = 0;
		T.kappa[2 * d + 1] = DthetaDdist;
		T.backward[2 * d + 1] = 1;
	}
	return PP_OK;
}

void free_planner(pp_planner* p)
{
	if (!p)
		return;
#if PP_ROWS_STATS
	{
		unsigned long long h[24] = {};
		if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rowsStats), sizeof h) == hipSuccess) {
			fprintf(stderr, "[rows stats] wave iterations with 1/2/3/4 busy rows: %llu %llu %llu %llu\n", h[1], h[2], h[3], h[4]);
			static const char* const names[14] = { "take", "set-aside", "pop+refill", "node", "child validity", "march", "truncate+cost", "cell-node+staging", "insert", "write", "rs", "endpoint", "keymap+heuristics", "voronoi" };
			unsigned long long tot = 0, it = h[1] + h[2] + h[3] + h[4];
			for (int i = 0; i < 14; i++)
				tot += h[8 + i];
			for (int i = 0; i < 14; i++)
				fprintf(stderr, "[rows stats]   %-18s %5.1f %%  %8.0f clk / wave iteration\n", names[i], 100.0 * (double)h[8 + i] / (double)(tot ? tot : 1), (double)h[8 + i] / (double)(it ? it : 1));
		}
	}
#endif
	void* postPtrs[] = { p->post.ratios, p->post.resampled, p->post.smoothed, p->post.cusp, p->post.optimise, p->post.edgeEnd, p->post.out };
	for (void* q : postPtrs)
		if (q)
			(void)hipFree(q);
	void* ptrs[] = { p->guardLog, p->guardCount, p->bandMeta, p->bands, p->orderKeys, p->order, p->suspended, p->paths, p->mtStates, p->table, p->costFields, p->wfWorkspace, p->wfError, p->tilesCtl, p->tilesFallback, p->nodes, p->heaps, p->keymaps, p->expanded, p->rsLogs, p->results, p->prof, p->dStarts,
		p->dGoals, p->dSeeds };
	for (void* q : ptrs)
		if (q)
			(void)hipFree(q);
	if (p->directStream)
		(void)hipStreamDestroy(p->directStream);
	if (p->e3)
		(void)hipEventDestroy(p->e3);
	if (p->e0)
		(void)hipEventDestroy(p->e0);
	if (p->e1)
		(void)hipEventDestroy(p->e1);
	if (p->e2)
		(void)hipEventDestroy(p->e2);
	pp_map* map = p->map;
	delete p;
	pph::map_release(map); // the planner kept its map (and through it the context) alive
}

} // namespace

enum class PlannerUse { Batches, Pipeline, PipelineLogged };
static int create_planner(pp_map* map, const pp_hybrid_params* params, int32_t max_batch, int32_t max_nodes_per_query, int32_t search_rows, PlannerUse use, pp_planner** out);

extern "C" {

int pp_planner_create(pp_map* map, const pp_hybrid_params* params, int32_t max_batch, int32_t max_nodes_per_query, pp_planner** out)
{
	return pp_planner_create_ex(map, params, max_batch, max_nodes_per_query, 0, out);
}

int pp_planner_create_ex(pp_map* map, const pp_hybrid_params* params, int32_t max_batch, int32_t max_nodes_per_query, int32_t search_rows, pp_planner** out)
{
	return create_planner(map, params, max_batch, max_nodes_per_query, search_rows, PlannerUse::Batches, out);
}

/// use: Batches = pp_planner_create_ex's planner; Pipeline / PipelineLogged = the buffer set of a streaming pipeline (pp_pipeline.hpp):
/// max_batch field slots, always the rows kernel, no hand-over lists, the expansion log only when asked for (4 B x max_nodes per slot)
static int create_planner(pp_map* map, const pp_hybrid_params* params, int32_t max_batch, int32_t max_nodes_per_query, int32_t search_rows, PlannerUse use, pp_planner** out)
{
	const bool forPipeline = use != PlannerUse::Batches;
	if (!map || !params || !out || max_batch < 1 || max_nodes_per_query < 16 || search_rows < 0) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	if (!map->dist || !map->occ8 || !map->pathcost) {
		set_error("map set incomplete: upload dist2, occupancy and path cost first");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	auto* p = new pp_planner();
	p->map = map;
	__atomic_add_fetch(&map->refs, 1, __ATOMIC_RELAXED);
	p->params = *params;
	p->maxBatch = max_batch;
	p->maxNodes = max_nodes_per_query;

	// StatePropagator constructor, hybrid_a_star.cpp:13-29: deltas {0, +d1, -d1, +d2, -d2, ...}
	const double wheelbase = params->wheelbase, rearToCenter = 0.0;
	// GetSteeringAngleFromTurningRadius, kinematic_bicycle_model.cpp:34-41
	const double deltaMax = std::atan(wheelbase / std::sqrt(std::pow(params->min_turning_radius, 2) - std::pow(rearToCenter, 2)));
	p->deltas.push_back(0.0);
	for (unsigned int i = 0; i < params->num_generated_motion / 2; i++) {
		double delta = (i + 1) / 2.0 * deltaMax;
		p->deltas.push_back(delta);
		p->deltas.push_back(-delta);
	}
	SearchArgs& A = p->args;
	A.m = map->view();
	if (int rc = primitives_from_deltas(p)) {
		free_planner(p);
		return rc;
	}
	A.rp.arcLength = params->spatial_resolution * 1.5;
	A.rp.spatialRes = params->spatial_resolution;
	A.rp.angularRes = params->angular_resolution;
	A.rp.lat.set(params->spatial_resolution, params->angular_resolution);
	A.rp.forwardMult = params->forward_cost_multiplier;
	A.rp.reverseMult = params->reverse_cost_multiplier;
	A.rp.voronoiMult = params->voronoi_cost_multiplier;
	A.rp.voroDiagRes = (float)(map->desc.resolution * std::sqrt(2.0));
	A.rp.headingAlias = params->heading_alias;
	int32_t dims[3];
	double offs[2];
	if (int rc = pp_nonholo_dims(map->desc.lower, map->desc.upper, params, dims, offs)) {
		free_planner(p);
		return rc;
	}
	A.heur.nx = dims[0];
	A.heur.ny = dims[1];
	A.heur.na = dims[2];
	A.heur.spatialRes = params->spatial_resolution;
	A.heur.angularRes = params->angular_resolution;
	A.heur.lat.set(params->spatial_resolution, params->angular_resolution);
	A.heur.offX = offs[0];
	A.heur.offY = offs[1];
	A.heur.minMult = std::min(params->reverse_cost_multiplier, params->forward_cost_multiplier);
	A.heur.negativeKRead = params->negative_k_read;
	// ObstaclesHeuristic constructor, heuristics.cpp:97-104
	A.heur.obstDiagRes = (float)(std::sqrt(2) * map->desc.resolution);
	A.heur.obstCostMult = (float)(std::min(params->reverse_cost_multiplier, params->forward_cost_multiplier) * map->desc.resolution);
	A.rmin = params->min_turning_radius;
	A.rsRev = (float)params->reverse_cost_multiplier;
	A.rsFwd = (float)params->forward_cost_multiplier;
	A.rsSw = (float)params->direction_switching_cost;
	A.maxNodes = max_nodes_per_query;
	A.maxPath = max_nodes_per_query < 2048 ? max_nodes_per_query : 2048;
	p->maxPath = A.maxPath;
	A.cells = map->cells();
	A.fieldElems = (int64_t)field_tiled_elems(map->desc.rows, map->desc.cols);
	// key space: every discrete pose a state inside the bounds (plus one arc of slack) can take
	{
		const double sres = params->spatial_resolution, ares = params->angular_resolution;
		const double slack = A.rp.arcLength + 1.0;
		KeySpace& ks = A.ks;
		ks.x0 = (int)std::floor((map->desc.lower[0] - slack) / sres) - 1;
		ks.y0 = (int)std::floor((map->desc.lower[1] - slack) / sres) - 1;
		ks.nx = (int)std::ceil((map->desc.upper[0] + slack) / sres) + 1 - ks.x0 + 1;
		ks.ny = (int)std::ceil((map->desc.upper[1] + slack) / sres) + 1 - ks.y0 + 1;
		// heading bins: (int)(wrap(theta) / ares) in [-tmax, tmax]; with the reference's Release-build aliasing
		// (Appendix A Q6) every bin is folded into [-3, 3] by Pose2<int>::WrapTheta
		int tmax = (int)(M_PI / ares) + 1;
		if (params->heading_alias && tmax > 3)
			tmax = 3;
		ks.t0 = -tmax;
		ks.nt = 2 * tmax + 1;
		ks.nt = (ks.nt + 3) / 4 * 4; // keeps every query's key map 16-byte aligned
	}

	const size_t B = (size_t)max_batch, N = (size_t)max_nodes_per_query;
	const size_t tableBytes = (size_t)dims[0] * dims[1] * dims[2] * sizeof(double);
	p->wfBytesPerSlot = pph::wavefront_workspace_bytes(map->desc.rows, map->desc.cols);
	{
		const int resident = pph::wavefront_resident_blocks();
		p->wfSlots = max_batch < resident ? max_batch : resident;
	}
	{
		int perCu = 0, dev = 0;
		hipDeviceProp_t prop;
		p->searchWaves = 2048;
		if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
			hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_hybrid_search_rows<false>, 64 * PP_ROWS_WAVES_PER_WG, 0) == hipSuccess && perCu >= 1)
			p->searchWaves = perCu * PP_ROWS_WAVES_PER_WG * prop.multiProcessorCount;
		// Which search kernel?  The four-queries-per-wave kernel (pp_planner_rows.hpp) issues ~4x fewer instructions per
		// expansion and needs node/heap/key-map buffers only for its resident rows, so many batches fit in HBM at once;
		// the one-query-per-wave kernel advances a single query ~1.4x faster.  Throughput-sized planners take the
		// former, small ones (the plugin's single-query path) the latter.  PP_SEARCH_ROWS=0/1 forces either.
		const char* env = getenv("PP_SEARCH_ROWS");
		p->rowsKernel = forPipeline || (env && (env[0] == '0' || env[0] == '1') ? env[0] == '1' : max_batch > 64);
		// rows (= buffer slots) of the persistent grid: as many as can be resident, unless the caller shares the GPU
		// between several planners (bench.py: resident rows / batches in flight)
		int rows = p->searchWaves * kRowsPerWave;
		if (search_rows > 0 && search_rows < rows)
			rows = search_rows;
		const int wanted = (max_batch + kRowsPerWave - 1) / kRowsPerWave * kRowsPerWave;
		if (wanted < rows)
			rows = wanted;
		p->searchRows = (rows + kRowsPerWave - 1) / kRowsPerWave * kRowsPerWave;
		// A batch ends with its longest query (65 k expansions when a goal is unreachable for the car).  Queries that reach
		// `suspendAfter` expansions are set aside by the rows kernel (open list flushed into the heap, scalars in a
		// SuspendRec, the row goes on in a spare slot) and finished one query per wave, 14 instead of ~25 us per expansion.
		// Only the extreme tail moves: per expansion the rows kernel is the cheaper one and the GPU is capacity-bound with
		// eight batches in flight (measured: 32768 -> 10.2 k plans/s, 8192 -> 8.0 k, never -> 9.8 k).  Optionally a second
		// pass of the rows kernel continues the set-aside queries up to `suspendAfter2` first (PP_SEARCH_SUSPEND_AFTER2;
		// measured no better: 8192 / 32768 -> 7.6 k, 12288 / 32768 -> 10.1 k).  PP_SEARCH_SUSPEND_AFTER=0: no hand-over.
		// tuning knobs from the environment are clamped to their meaningful ranges: none of them may change results or
		// make an allocation size negative
		auto env_int = [](const char* name, int dflt, int lo, int hi) {
			const char* v = getenv(name);
			if (!v || !*v)
				return dflt;
			const long x = strtol(v, nullptr, 10);
			return (int)(x < lo ? lo : (x > hi ? hi : x));
		};
		A.suspendAfter = p->rowsKernel && !forPipeline ? env_int("PP_SEARCH_SUSPEND_AFTER", 32768, 0, 1 << 30) : 0; // (a pipeline's rows keep their queries)
		A.suspendAfter2 = env_int("PP_SEARCH_SUSPEND_AFTER2", 0, 0, 1 << 30);
		A.extraSlots = A.suspendAfter > 0 ? env_int("PP_SEARCH_EXTRA_SLOTS", (max_batch + 15) / 16, 0, max_batch) : 0; // queries that may be set aside (the rest stays)
		A.searchRows = p->searchRows;
		A.listCap = A.extraSlots + p->searchRows;
		// A batch lasts as long as its longest query, and a query is a chain of dependent expansions: it runs faster alone in
		// a wave (k_hybrid_search: ~11 us per expansion) than as one of four (rows kernel: 15-20 us).  The wavefront kernel
		// already ranks the queries by probable length for the hand-out order; the first PP_SEARCH_DIRECT of that order get
		// a wave of their own, in slots behind the spare ones.
		p->directCount = p->rowsKernel && !forPipeline ? env_int("PP_SEARCH_DIRECT", 0, 0, max_batch / 2) : 0;
		A.directCount = 0; // set per call (only when the order is available)
		// compaction (pp_planner_rows.hpp): waves whose queue is empty and that have at most this many busy rows re-queue
		// their queries for a second pass that packs them four per wave.  Off by default: it issues fewer instructions
		// (a wave costs the same with one busy row as with four) but the passes of one batch run one after the other, and
		// with eight batches in flight the longer per-batch latency costs more than the saved issue slots
		// (measured: 8.1 k plans/s with PP_SEARCH_COMPACT=2 against 10.9 k without).
		p->compactBelow = p->rowsKernel && !forPipeline ? env_int("PP_SEARCH_COMPACT", 0, 0, kRowsPerWave) : 0;
	}
	hipError_t e = hipSuccess;
	// Headroom.  The kernels this planner launches need scratch (private segment: k_hybrid_search_rows 408 B, k_wavefront
	// 132 B, k_hybrid_search 112 B per lane, tools/kernel_resources.py), which the runtime allocates per hardware queue at a
	// kernel's FIRST dispatch: private bytes x 64 lanes x every wave slot of the device.  A planner that takes the last byte
	// of HBM makes that allocation fail and the runtime aborts the process (round 1: HSA_STATUS_ERROR_OUT_OF_RESOURCES in
	// k_wavefront after a 2048-row planner).  So: (i) the kernels are dispatched once with empty grids BEFORE the large
	// allocations, which makes this stream's queue allocate its scratch now, and (ii) the planner refuses to take memory
	// beyond free - reserve, where the reserve covers the same scratch for the other hardware queues a process may use.
	size_t planned = 0;
	std::vector<std::pair<void**, size_t>> wanted;
	auto alloc = [&](void** ptr, size_t bytes) {
		bytes = bytes ? bytes : 1;
		wanted.push_back({ ptr, bytes });
		planned += (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1); // the allocator works in 2 MiB granules
	};
	alloc((void**)&p->table, tableBytes);
	alloc((void**)&p->costFields, B * (size_t)A.fieldElems * sizeof(float));
	alloc((void**)&p->wfWorkspace, (size_t)p->wfBytesPerSlot * p->wfSlots);
	alloc((void**)&p->tilesCtl, 64);
	alloc((void**)&p->tilesFallback, B * 4);
	alloc((void**)&p->wfError, 32); // control block: {wavefront error flag, wavefront goal counter, pass-1 query counter, pass-1 set-aside count,
	                                // pass-2 record counter, pass-2 set-aside count}
	// search buffers: one set per resident row (rows kernel) or per query (one-query-per-wave kernel)
	const size_t S = p->rowsKernel ? (size_t)p->searchRows + (size_t)A.extraSlots + (size_t)p->directCount : B;
	alloc((void**)&p->suspended, 2 * (size_t)(A.listCap > 0 ? A.listCap : 1) * sizeof(SuspendRec));
	alloc((void**)&p->mtStates, (p->rowsKernel ? S : 1) * Mt64::N * sizeof(unsigned long long));
	alloc((void**)&p->nodes, S * N * sizeof(Node));
	alloc((void**)&p->bands, S * (size_t)(kBands * kBandCap) * sizeof(HeapEntry));
	alloc((void**)&p->bandMeta, S * (size_t)kBands);
	alloc((void**)&p->heaps, S * N * sizeof(HeapEntry));
	alloc((void**)&p->keymaps, S * A.ks.size() * 4);
	if (use != PlannerUse::Pipeline) // (a pipeline keeps the expansion log only for parity tests: 4 B x max_nodes per field slot)
		alloc((void**)&p->expanded, B * N * 4);
	alloc((void**)&p->order, B * 4);
	alloc((void**)&p->orderKeys, B * 4);
	alloc((void**)&p->paths, B * (size_t)A.maxPath * sizeof(PathRec));
	alloc((void**)&p->rsLogs, B * kRsLogCap * sizeof(RsLogEntry));
	alloc((void**)&p->results, B * sizeof(DevResult));
	if (!p->rowsKernel) { // the lattice-line log of pp_planner_certify_lattice: planners that keep the tree per query
		alloc((void**)&p->guardLog, B * (size_t)kGuardLogCap * sizeof(GuardRec));
		alloc((void**)&p->guardCount, B * 4);
	}
	alloc((void**)&p->prof, (forPipeline ? 1 : B) * PH_COUNT * sizeof(unsigned long long));
	alloc((void**)&p->dStarts, B * 24);
	alloc((void**)&p->dGoals, B * 24);
	alloc((void**)&p->dSeeds, B * 8);
	{
		if (int rc = warm_up_kernels(p, map)) {
			free_planner(p);
			return rc;
		}
		size_t freeB = 0, totalB = 0;
		e = hipMemGetInfo(&freeB, &totalB);
		int dev = 0;
		hipDeviceProp_t prop;
		size_t reserve = (size_t)1 << 30;
		if (e == hipSuccess && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
			const size_t waveSlots = (size_t)prop.multiProcessorCount * 32; // 8 waves on each of a CU's 4 SIMDs
			reserve += (size_t)kMaxPrivateBytes * 64 * waveSlots * kReserveQueues;
		}
		if (e == hipSuccess && planned + reserve > freeB) {
			free_planner(p);
			set_error("planner needs " + std::to_string(planned >> 20) + " MiB, the device has " + std::to_string(freeB >> 20) + " MiB free and " +
				std::to_string(reserve >> 20) + " MiB stay reserved for kernel scratch");
			return PP_ERR_CAPACITY;
		}
		for (auto& w : wanted)
			if (e == hipSuccess)
				e = hipMalloc(w.first, w.second);
	}
	if (e == hipSuccess)
		e = hipMemset(p->tilesCtl, 0, 64);
	if (e == hipSuccess)
		e = hipEventCreate(&p->e0);
	if (e == hipSuccess)
		e = hipEventCreate(&p->e1);
	if (e == hipSuccess)
		e = hipEventCreate(&p->e2);
	if (e == hipSuccess && p->directCount > 0)
		e = hipEventCreateWithFlags(&p->e3, hipEventDisableTiming);
	if (e == hipSuccess && p->directCount > 0)
		e = hipStreamCreateWithFlags(&p->directStream, hipStreamNonBlocking);
	if (e != hipSuccess) {
		free_planner(p);
		return pph::hip_fail(e, "planner allocation");
	}
	A.heur.table = p->table;
	A.guardLog = p->guardLog;
	A.guardCount = p->guardCount;
	p->nextQuery = p->wfError + 2;
	p->nh.nx = dims[0];
	*out = p;
	return PP_OK;
}

int pp_planner_destroy(pp_planner* planner)
{
	if (!planner)
		return PP_OK;
	(void)hipSetDevice(planner->map->ctx->device);
	(void)hipStreamSynchronize(planner->map->ctx->stream);
	free_planner(planner);
	return PP_OK;
}

int pp_planner_num_primitives(pp_planner* planner) { return planner ? planner->args.prims.n : 0; }

int pp_planner_set_primitives(pp_planner* planner, int32_t n_steering_angles, const double* steering_angles)
{
	if (!planner || !steering_angles || n_steering_angles < 1) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	if (planner->owner && pp_pipeline_in_flight(planner->owner) > 0) {
		// the persistent search grid received its primitive table when its waves were launched: waves of later launches would get the new
		// one, and a query's result would depend on which wave claims it
		set_error("the pipeline has queries in flight: poll them all before changing the primitives");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(planner->map->ctx->device));
	PP_HIP_TRY(hipStreamSynchronize(planner->map->ctx->stream)); // a batch in flight keeps the table it was launched with
	const std::vector<double> before = planner->deltas;
	planner->deltas.assign(steering_angles, steering_angles + n_steering_angles);
	if (int rc = primitives_from_deltas(planner)) {
		planner->deltas = before;
		(void)primitives_from_deltas(planner);
		return rc;
	}
	return PP_OK;
}

int pp_planner_set_nonholo_table(pp_planner* planner, const double* table_host)
{
	if (!planner) {
		set_error("null planner");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(planner->map->ctx->device));
	hipStream_t s = planner->map->ctx->stream;
	const HeurView& h = planner->args.heur;
	const size_t bytes = (size_t)h.nx * h.ny * h.na * sizeof(double);
	if (table_host) {
		PP_HIP_TRY(hipMemcpyAsync(planner->table, table_host, bytes, hipMemcpyHostToDevice, s));
	} else {
		if (int rc = pp_nonholo_build_dev(planner->map->ctx, planner->map->desc.lower, planner->map->desc.upper, &planner->params, planner->table))
			return rc;
	}
	PP_HIP_TRY(hipStreamSynchronize(s));
	planner->tableReady = true;
	return PP_OK;
}

int pp_planner_start_after_fields_of(pp_planner* planner, pp_planner* predecessor)
{
	if (!planner || !predecessor || planner == predecessor || planner->map->ctx->device != predecessor->map->ctx->device) {
		set_error("two different planners on the same device");
		return PP_ERR_INVALID;
	}
	planner->startAfter = predecessor->e1; // recorded behind the predecessor's wavefront launch; never recorded = no wait
	return PP_OK;
}

int pp_planner_get_nonholo_table(pp_planner* planner, double* table_host)
{
	if (!planner || !table_host || !planner->tableReady) {
		set_error("table not available");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(planner->map->ctx->device));
	const HeurView& h = planner->args.heur;
	PP_HIP_TRY(hipMemcpy(table_host, planner->table, (size_t)h.nx * h.ny * h.na * sizeof(double), hipMemcpyDeviceToHost));
	return PP_OK;
}

int pp_planner_search_batch_dev(pp_planner* planner, int32_t n_queries, const double* starts_dev, const double* goals_dev, const uint64_t* seeds_dev)
{
	if (!planner || n_queries < 0 || n_queries > planner->maxBatch || (n_queries > 0 && (!starts_dev || !goals_dev || !seeds_dev))) {
		set_error("invalid arguments (n_queries must be <= max_batch)");
		return PP_ERR_INVALID;
	}
	if (planner->pipelineOwned) {
		set_error("this planner is a pipeline's buffer set (pp_pipeline_planner): queries go through pp_pipeline_submit");
		return PP_ERR_INVALID;
	}
	if (n_queries == 0)
		return PP_OK;
	PP_HIP_TRY(hipSetDevice(planner->map->ctx->device));
	if (!planner->tableReady)
		if (int rc = pp_planner_set_nonholo_table(planner, nullptr))
			return rc;
	hipStream_t s = planner->map->ctx->stream;
	planner->args.m = planner->map->view(); // validator tunables may have changed
	const MapView& m = planner->args.m;
	if (planner->startAfter) { // phase this batch behind another planner's (throughput use, several batches in flight)
		PP_HIP_TRY(hipStreamWaitEvent(s, planner->startAfter, 0));
		planner->startAfter = nullptr;
	}
	PP_HIP_TRY(hipMemsetAsync(planner->wfError, 0, 32, s)); // the step's only fill: every counter of every kernel
	if (planner->guardCount)
		PP_HIP_TRY(hipMemsetAsync(planner->guardCount, 0, (size_t)n_queries * 4, s));
	PP_HIP_TRY(hipEventRecord(planner->e0, s));
	// the rows kernel hands the queries out longest-first (order written by the wavefront kernel's last workgroup)
	static const bool lpt = !(getenv("PP_SEARCH_ORDER") && getenv("PP_SEARCH_ORDER")[0] == '0');
	const bool ordered = planner->rowsKernel && lpt && n_queries <= 4096 && n_queries > planner->searchRows;
	// ObstaclesHeuristic::Update for every query's goal (hybrid_a_star.cpp:249)
#ifdef PP_ENABLE_DEBUG_SKIP // diagnostic builds only (tools/build_variant.py skip -DPP_ENABLE_DEBUG_SKIP=1; bench.py --debug-skip): the shipped
	// library always runs both kernels.  1 = no wavefront launch, 2 = no search launch, 3 = set-aside queries are dropped
	const char* const dbgEnv = getenv("PP_DEBUG_SKIP");
	const int dbgSkip = dbgEnv ? atoi(dbgEnv) : 0;
#else
	constexpr int dbgSkip = 0;
#endif
	if (dbgSkip != 1) {
		pph::WavefrontPublish pub;
		pub.tilesCtl = planner->tilesCtl;
		pub.tilesFallback = planner->tilesFallback;
		pub.occBits = planner->map->occBits;
		PP_HIP_TRY(pph::launch_wavefront(s, m, n_queries, nullptr, planner->costFields, planner->wfWorkspace, planner->wfBytesPerSlot, planner->wfSlots,
			planner->wfError, nullptr, /*tiledOut=*/true, /*goalPoses=*/goals_dev, /*countersZeroed=*/true, ordered ? starts_dev : nullptr, ordered ? planner->order : nullptr,
			planner->wfError + 6, planner->orderKeys, pub));
	}
	PP_HIP_TRY(hipEventRecord(planner->e1, s));
	const int nDirect = ordered && planner->directCount > 0 && dbgSkip != 2 ? (planner->directCount < n_queries / 2 ? planner->directCount : n_queries / 2) : 0;
	planner->args.directCount = nDirect;
	if (nDirect > 0) {
		hipStream_t const ds = planner->directStream;
		PP_HIP_TRY(hipStreamWaitEvent(ds, planner->e1, 0));
		hipLaunchKernelGGL(k_hybrid_search<false>, dim3(nDirect), dim3(64), 0, ds, planner->args, nDirect, starts_dev, goals_dev, seeds_dev, planner->costFields, planner->nodes,
			planner->heaps, planner->keymaps, planner->expanded, planner->rsLogs, planner->paths, planner->results, planner->prof, nullptr, nullptr, nullptr, planner->bands,
			planner->bandInvW, planner->bandMeta, planner->order, planner->searchRows + planner->args.extraSlots);
		PP_HIP_TRY(hipGetLastError());
		PP_HIP_TRY(hipEventRecord(planner->e3, ds));
	}
	if (dbgSkip == 2) {
	} else if (planner->rowsKernel) {
		// four queries per wave, taken from a counter by a persistent grid (pp_planner_rows.hpp)
		const int wavesWanted = (n_queries + kRowsPerWave - 1) / kRowsPerWave;
		const int wavesMax = planner->searchRows / kRowsPerWave;
		const int grid = wavesWanted < wavesMax ? wavesWanted : wavesMax;
		int* const ctl = planner->nextQuery; // {pass-1 query counter, list-1 count, pass-2 record counter, list-2 count, (wavefront), spare slots}
		int* const spare = planner->wfError + 7;
		SuspendRec* const list1 = planner->suspended;
		SuspendRec* const list2 = planner->suspended + planner->args.listCap;
		const int cap1 = planner->args.suspendAfter, cap2 = planner->args.suspendAfter2, cpt = planner->compactBelow;
		constexpr int kWg = PP_ROWS_WAVES_PER_WG;
		planner->args.rowsWaves = grid;
		hipLaunchKernelGGL(k_hybrid_search_rows<false>, dim3((grid + kWg - 1) / kWg), dim3(64 * kWg), 0, s, planner->args, n_queries, starts_dev, goals_dev, seeds_dev, planner->costFields, planner->nodes,
			planner->heaps, planner->keymaps, planner->expanded, planner->rsLogs, planner->paths, planner->mtStates, planner->results, ctl, list1,
			ordered ? planner->order : nullptr, cap1, nullptr, nullptr, ctl + 1, spare, cpt, planner->bands, planner->bandInvW, planner->bandMeta, PipeView {});
		PP_HIP_TRY(hipGetLastError());
		const bool secondPass = cpt > 0 || (cap1 > 0 && cap2 > cap1);
		if (secondPass) {
			// second pass of the rows kernel over list 1 (its length is read on the device); it ends waves with a single
			// busy row, and the one-query-per-wave kernel finishes those
			const int waves2 = (planner->args.listCap + kRowsPerWave - 1) / kRowsPerWave;
			planner->args.rowsWaves = waves2 < wavesMax ? waves2 : wavesMax;
			hipLaunchKernelGGL(k_hybrid_search_rows<false>, dim3((planner->args.rowsWaves + kWg - 1) / kWg), dim3(64 * kWg), 0, s, planner->args, n_queries, starts_dev, goals_dev, seeds_dev,
				planner->costFields, planner->nodes, planner->heaps, planner->keymaps, planner->expanded, planner->rsLogs, planner->paths, planner->mtStates,
				planner->results, ctl + 2, list2, nullptr, cap2 > cap1 ? cap2 : 0, list1, ctl + 1, ctl + 3, spare, cpt > 0 ? 1 : 0, planner->bands, planner->bandInvW,
				planner->bandMeta, PipeView {});
			PP_HIP_TRY(hipGetLastError());
		}
		if ((secondPass || cap1 > 0) && dbgSkip != 3) // whatever is still set aside: one wave per query (the block count is read on the device)
			hipLaunchKernelGGL(k_hybrid_search<false>, dim3(planner->args.listCap), dim3(64), 0, s, planner->args, n_queries, starts_dev, goals_dev, seeds_dev,
				planner->costFields, planner->nodes, planner->heaps, planner->keymaps, planner->expanded, planner->rsLogs, planner->paths, planner->results, planner->prof,
				secondPass ? list2 : list1, secondPass ? ctl + 3 : ctl + 1, planner->mtStates, planner->bands, planner->bandInvW, planner->bandMeta, nullptr, 0);
	} else if (planner->profile)
		hipLaunchKernelGGL(k_hybrid_search<true>, dim3(n_queries), dim3(64), 0, s, planner->args, n_queries, starts_dev, goals_dev, seeds_dev, planner->costFields,
			planner->nodes, planner->heaps, planner->keymaps, planner->expanded, planner->rsLogs, planner->paths, planner->results, planner->prof, nullptr, nullptr, nullptr, planner->bands, planner->bandInvW, planner->bandMeta, nullptr, 0);
	else
		hipLaunchKernelGGL(k_hybrid_search<false>, dim3(n_queries), dim3(64), 0, s, planner->args, n_queries, starts_dev, goals_dev, seeds_dev, planner->costFields,
			planner->nodes, planner->heaps, planner->keymaps, planner->expanded, planner->rsLogs, planner->paths, planner->results, planner->prof, nullptr, nullptr, nullptr, planner->bands, planner->bandInvW, planner->bandMeta, nullptr, 0);
	PP_HIP_TRY(hipGetLastError());
	if (nDirect > 0)
		PP_HIP_TRY(hipStreamWaitEvent(s, planner->e3, 0)); // before e2: the search time covers both kernels
	PP_HIP_TRY(hipEventRecord(planner->e2, s));
	planner->lastBatch = n_queries;
	return PP_OK;
}

int pp_planner_fetch_results(pp_planner* planner, int32_t n_queries, pp_query_result* results_host)
{
	if (!planner || n_queries < 0 || n_queries > planner->lastBatch || (n_queries > 0 && !results_host)) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	if (planner->pipelineOwned) {
		set_error("this planner is a pipeline's buffer set: its results arrive through pp_pipeline_poll");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(planner->map->ctx->device));
	hipStream_t s = planner->map->ctx->stream;
	planner->hostResults.resize(planner->lastBatch);
	int32_t err = 0;
	PP_HIP_TRY(hipMemcpyAsync(planner->hostResults.data(), planner->results, (size_t)planner->lastBatch * sizeof(DevResult), hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipMemcpyAsync(&err, planner->wfError, 4, hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipStreamSynchronize(s));
	(void)hipEventElapsedTime(&planner->wavefrontMs, planner->e0, planner->e1);
	(void)hipEventElapsedTime(&planner->searchMs, planner->e1, planner->e2);
	if (err) {
		set_error("obstacle-heuristic open list exceeded its workspace");
		return PP_ERR_CAPACITY;
	}
	for (int i = 0; i < n_queries; i++)
		results_host[i] = planner->hostResults[i].r;
	return PP_OK;
}

int pp_planner_search_batch(pp_planner* planner, int32_t n_queries, const double* starts_host, const double* goals_host, const uint64_t* seeds_host,
	pp_query_result* results_host)
{
	if (!planner || n_queries < 0 || n_queries > planner->maxBatch || (n_queries > 0 && (!starts_host || !goals_host || !seeds_host || !results_host))) {
		set_error("invalid arguments (n_queries must be <= max_batch)");
		return PP_ERR_INVALID;
	}
	if (planner->pipelineOwned) {
		set_error("this planner is a pipeline's buffer set (pp_pipeline_planner): queries go through pp_pipeline_submit");
		return PP_ERR_INVALID;
	}
	if (n_queries == 0)
		return PP_OK;
	PP_HIP_TRY(hipSetDevice(planner->map->ctx->device));
	hipStream_t s = planner->map->ctx->stream;
	PP_HIP_TRY(hipMemcpyAsync(planner->dStarts, starts_host, (size_t)n_queries * 24, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(hipMemcpyAsync(planner->dGoals, goals_host, (size_t)n_queries * 24, hipMemcpyHostToDevice, s));
	PP_HIP_TRY(hipMemcpyAsync(planner->dSeeds, seeds_host, (size_t)n_queries * 8, hipMemcpyHostToDevice, s));
	if (int rc = pp_planner_search_batch_dev(planner, n_queries, planner->dStarts, planner->dGoals, planner->dSeeds))
		return rc;
	return pp_planner_fetch_results(planner, n_queries, results_host);
}

int pp_planner_debug_nodes(pp_planner* planner, int32_t q, int32_t max_nodes, int32_t* parents_host, double* poses_host, double* costs_host, int32_t* dead_host)
{
	if (!planner || q < 0 || q >= planner->lastBatch || (int)planner->hostResults.size() <= q || max_nodes < 0) {
		set_error("no fetched result for this query (call pp_planner_fetch_results first)");
		return PP_ERR_INVALID;
	}
	if (planner->rowsKernel) {
		set_error("node records are kept per query by the one-query-per-wave kernel only (PP_SEARCH_ROWS=0 or max_batch <= 64)");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(planner->map->ctx->device));
	int n = planner->hostResults[q].r.n_nodes;
	if (n > max_nodes)
		n = max_nodes;
	std::vector<Node> nodes(n);
	if (n)
		PP_HIP_TRY(hipMemcpy(nodes.data(), planner->nodes + (size_t)q * planner->maxNodes, (size_t)n * sizeof(Node), hipMemcpyDeviceToHost));
	for (int i = 0; i < n; i++) {
		if (parents_host)
			parents_host[i] = nodes[i].parent;
		if (poses_host) {
			poses_host[3 * i] = nodes[i].x;
			poses_host[3 * i + 1] = nodes[i].y;
			poses_host[3 * i + 2] = nodes[i].t;
		}
		if (costs_host) {
			costs_host[2 * i] = nodes[i].pathCost;
			costs_host[2 * i + 1] = nodes[i].totalCost;
		}
		if (dead_host)
			dead_host[i] = nodes[i].dead;
	}
	return PP_OK;
}

int pp_planner_debug_node_actions(pp_planner* planner, int32_t q, int32_t max_nodes, int32_t* action_host, double* length_host)
{
	if (!planner || q < 0 || q >= planner->lastBatch || (int)planner->hostResults.size() <= q || max_nodes < 0) {
		set_error("no fetched result for this query (call pp_planner_fetch_results first)");
		return PP_ERR_INVALID;
	}
	if (planner->rowsKernel) {
		set_error("node records are kept per query by the one-query-per-wave kernel only (PP_SEARCH_ROWS=0 or max_batch <= 64)");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(planner->map->ctx->device));
	int n = planner->hostResults[q].r.n_nodes;
	if (n > max_nodes)
		n = max_nodes;
	std::vector<Node> nodes(n);
	if (n)
		PP_HIP_TRY(hipMemcpy(nodes.data(), planner->nodes + (size_t)q * planner->maxNodes, (size_t)n * sizeof(Node), hipMemcpyDeviceToHost));
	for (int i = 0; i < n; i++) {
		if (action_host)
			action_host[i] = nodes[i].action;
		if (length_host)
			length_host[i] = nodes[i].length;
	}
	return PP_OK;
}

/// SURVEY 7.3 H2 as a contract.  The search kernel flags every child whose DiscretizePose quotient lies within 1e-9 cells of a lattice
/// line (n_lattice_boundary_hits) -- the only poses a last-bit difference between this libm and glibc could put into another cell
/// (pose differences are ~1e-15) -- and, on planners that keep the tree, logs them: parent node, primitive, arc length, the cell the
/// device chose.  Here the host recomputes (i) every CREATED constant-steer node and (ii) every LOGGED constant-steer child with the
/// C library the reference links (glibc sin / cos), each along its own chain of ancestors from the start pose -- the arithmetic the
/// reference performs (KinematicBicycleModel::ConstantSteer, models/kinematic_bicycle_model.cpp:5-32, through
/// PathConstantSteer::Interpolate with the stored, possibly truncated, length) -- and discretises it (HybridAStar::DiscretizePose,
/// algo/hybrid_a_star.h:104-111).  n_cell_mismatches: recomputed cells that differ from the device's.  n_unverified: flagged events
/// that cannot be recomputed here (a Reeds-Shepp child on a lattice line; log entries beyond the 64 kept per query).
/// Both 0 certifies the query's discrete outputs against the reference's arithmetic; else: hand the query to the CPU reference.
/// Needs the tree, i.e. a planner of the one-query-per-wave kind (max_batch <= 64 or PP_SEARCH_ROWS=0); queries of a throughput planner
/// or pipeline that report n_lattice_boundary_hits > 0 are re-planned on such a planner (same device code, same results) first.
int pp_planner_certify_lattice(pp_planner* planner, int32_t q, int32_t* n_checked, int32_t* n_cell_mismatches, int32_t* n_unverified, double* max_pose_difference)
{
	if (!planner || q < 0 || q >= planner->lastBatch || (int)planner->hostResults.size() <= q) {
		set_error("no fetched result for this query (call pp_planner_fetch_results first)");
		return PP_ERR_INVALID;
	}
	if (planner->rowsKernel || !planner->guardLog) {
		set_error("the search tree and the lattice-line log are kept per query by the one-query-per-wave kernel only (PP_SEARCH_ROWS=0 or max_batch <= 64)");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(planner->map->ctx->device));
	const int n = planner->hostResults[q].r.n_nodes;
	std::vector<Node> nodes((size_t)(n > 0 ? n : 0));
	if (n > 0)
		PP_HIP_TRY(hipMemcpy(nodes.data(), planner->nodes + (size_t)q * planner->maxNodes, (size_t)n * sizeof(Node), hipMemcpyDeviceToHost));
	int nLog = 0;
	PP_HIP_TRY(hipMemcpy(&nLog, planner->guardCount + q, 4, hipMemcpyDeviceToHost));
	const int kept = nLog < kGuardLogCap ? nLog : kGuardLogCap;
	std::vector<GuardRec> log((size_t)kept);
	if (kept > 0)
		PP_HIP_TRY(hipMemcpy(log.data(), planner->guardLog + (size_t)q * kGuardLogCap, (size_t)kept * sizeof(GuardRec), hipMemcpyDeviceToHost));
	const SearchArgs& A = planner->args;
	auto wrap = [](double t) { // geometry/2dplane.h:36-45
		while (t > M_PI)
			t -= 2 * M_PI;
		while (t < -M_PI)
			t += 2 * M_PI;
		return t;
	};
	auto cell = [&](double x, double y, double t, int c[3]) { // hybrid_a_star.h:104-111 (x86 conversions), Pose2<int>::WrapTheta's aliasing (Appendix A Q6) when the planner models it
		c[0] = (int)(x / A.rp.spatialRes);
		c[1] = (int)(y / A.rp.spatialRes);
		int it = (int)(wrap(t) / A.rp.angularRes);
		if (A.rp.headingAlias) {
			while ((double)it > M_PI)
				it = (int)((double)it - 2 * M_PI);
			while ((double)it < -M_PI)
				it = (int)((double)it + 2 * M_PI);
		}
		c[2] = it;
	};
	auto steer = [&](double& x, double& y, double& t, int prim, double length) { // kinematic_bicycle_model.cpp:5-32 with beta = 0
		const double kappa = A.prims.kappa[prim];
		const double dist = A.prims.backward[prim] ? -length : length;
		if (std::fabs(kappa) > 1e-9) {
			const double t0 = t;
			t += dist * kappa;
			x += 1 / kappa * (std::sin(t) - std::sin(t0));
			y += 1 / kappa * (-std::cos(t) + std::cos(t0));
		} else {
			x += dist * std::cos(t);
			y += dist * std::sin(t);
		}
	};
	std::vector<double> hx((size_t)n), hy((size_t)n), ht((size_t)n);
	int checked = 0, bad = 0, unverified = nLog - kept;
	double worst = 0.0;
	for (int i = 0; i < n; i++) {
		const Node& nd = nodes[(size_t)i];
		const int a = nd.action;
		if (nd.parent < 0 || nd.parent >= i || a < 0 || a >= A.prims.n) { // root; Reeds-Shepp child (its pose is the path's end, not an arc's)
			hx[(size_t)i] = nd.x;
			hy[(size_t)i] = nd.y;
			ht[(size_t)i] = nd.t;
			continue;
		}
		double x = hx[(size_t)nd.parent], y = hy[(size_t)nd.parent], t = ht[(size_t)nd.parent];
		steer(x, y, t, a, nd.length);
		hx[(size_t)i] = x;
		hy[(size_t)i] = y;
		ht[(size_t)i] = t;
		int ch[3], cd[3];
		cell(x, y, t, ch);
		cell(nd.x, nd.y, nd.t, cd);
		checked++;
		if (ch[0] != cd[0] || ch[1] != cd[1] || ch[2] != cd[2])
			bad++;
		const double d = std::fmax(std::fmax(std::fabs(x - nd.x), std::fabs(y - nd.y)), std::fabs(t - nd.t));
		worst = d > worst ? d : worst;
	}
	for (const GuardRec& g : log) { // children on a lattice line, created or not
		if (g.kind != 1 || g.parent < 0 || g.parent >= n || g.prim < 0 || g.prim >= A.prims.n) {
			unverified++;
			continue;
		}
		double x = hx[(size_t)g.parent], y = hy[(size_t)g.parent], t = ht[(size_t)g.parent];
		steer(x, y, t, g.prim, g.length);
		int ch[3];
		cell(x, y, t, ch);
		checked++;
		if (ch[0] != g.ix || ch[1] != g.iy || ch[2] != g.it)
			bad++;
	}
	if (n_checked)
		*n_checked = checked;
	if (n_cell_mismatches)
		*n_cell_mismatches = bad;
	if (n_unverified)
		*n_unverified = unverified;
	if (max_pose_difference)
		*max_pose_difference = worst;
	return PP_OK;
}

int pp_planner_search_rows(pp_planner* planner) { return planner && planner->rowsKernel ? planner->searchRows : 0; }

int pp_planner_set_profiling(pp_planner* planner, int32_t enable)
{
	if (!planner) {
		set_error("null planner");
		return PP_ERR_INVALID;
	}
	if (enable && planner->rowsKernel) {
		set_error("phase profiling exists for the one-query-per-wave kernel only: create the planner with PP_SEARCH_ROWS=0");
		return PP_ERR_INVALID;
	}
	planner->profile = enable != 0;
	return PP_OK;
}

int pp_planner_phase_cycles(pp_planner* planner, int32_t n_queries, uint64_t* cycles_host)
{
	if (!planner || n_queries < 0 || n_queries > planner->lastBatch || !cycles_host) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(planner->map->ctx->device));
	PP_HIP_TRY(hipMemcpy(cycles_host, planner->prof, (size_t)n_queries * PH_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	return PP_OK;
}

int pp_planner_postprocess(pp_planner* planner, int32_t n_queries, float path_interpolation, const pp_smoother_params* smoother, int32_t max_points, pp_post_result* results_host)
{
	if (!planner || n_queries < 0 || n_queries > planner->lastBatch || max_points < 8 || max_points > 8 * kPostThreads || !(path_interpolation > 0.0f)) {
		set_error("invalid arguments (n_queries <= last batch, 8 <= max_points <= 2048, path_interpolation > 0)");
		return PP_ERR_INVALID;
	}
	pp_map* map = planner->map;
	if (!map->obstLabel[map->obstResult] || !map->voroLabel[map->voroResult]) {
		set_error("nearest-obstacle / nearest-edge cell grids missing: pp_map_update_gvd or pp_map_upload_nearest_cells first");
		return PP_ERR_INVALID;
	}
	if (n_queries == 0)
		return PP_OK;
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	hipStream_t s = map->ctx->stream;
	const size_t B = (size_t)planner->maxBatch;
	if (planner->postMaxPoints < max_points) {
		void* old[] = { planner->post.ratios, planner->post.resampled, planner->post.smoothed, planner->post.cusp, planner->post.optimise };
		planner->postMaxPoints = 0; // until every buffer below exists again: a failed allocation must not leave a stale capacity behind
		for (void* q : old)
			if (q)
				(void)hipFree(q);
		planner->post.ratios = planner->post.resampled = planner->post.smoothed = nullptr;
		planner->post.cusp = planner->post.optimise = nullptr;
		PP_HIP_TRY(hipMalloc((void**)&planner->post.ratios, B * max_points * 8));
		PP_HIP_TRY(hipMalloc((void**)&planner->post.resampled, B * max_points * 24));
		PP_HIP_TRY(hipMalloc((void**)&planner->post.smoothed, B * max_points * 24));
		PP_HIP_TRY(hipMalloc((void**)&planner->post.cusp, B * max_points));
		PP_HIP_TRY(hipMalloc((void**)&planner->post.optimise, B * max_points));
		planner->postMaxPoints = max_points;
	}
	if (!planner->post.edgeEnd)
		PP_HIP_TRY(hipMalloc((void**)&planner->post.edgeEnd, B * (size_t)(planner->maxPath + 1) * 8));
	if (!planner->post.out)
		PP_HIP_TRY(hipMalloc((void**)&planner->post.out, B * sizeof(pp_post_result)));
	PostParams P {};
	P.pathInterpolation = path_interpolation;
	pp_smoother_params sp { 1e-3f, 2000, 0.01f, 0.0f, 0.4f, 0.02f, 0.2f, 0.4f, 0.2f, (float)(1.0 / planner->params.min_turning_radius) }; // smoother.h:28-60, hybrid_a_star.cpp:214
	if (smoother)
		sp = *smoother;
	P.stepTolerance = sp.step_tolerance;
	P.maxIterations = sp.max_iterations;
	P.learningRate = sp.learning_rate;
	P.pathWeight = sp.path_weight;
	P.smoothWeight = sp.smooth_weight;
	P.voronoiWeight = sp.voronoi_weight;
	P.collisionWeight = sp.collision_weight;
	P.curvatureWeight = sp.curvature_weight;
	P.collisionRatio = sp.collision_ratio;
	P.maxCurvature = sp.max_curvature;
	P.alpha = 20.0f; // GVD::alpha / dMax, gvd.h:181
	P.dMax = 30.0f;
	P.maxPoints = planner->postMaxPoints;
	planner->args.m = map->view();
	const size_t lds = (size_t)planner->postMaxPoints * 16;
	hipLaunchKernelGGL(k_postprocess, dim3(n_queries), dim3(kPostThreads), lds, s, planner->args, P, n_queries, planner->paths, planner->rsLogs, planner->results,
		map->obstLabel[map->obstResult], map->voroLabel[map->voroResult], planner->post);
	PP_HIP_TRY(hipGetLastError());
	planner->hostPost.resize(n_queries);
	PP_HIP_TRY(hipMemcpyAsync(planner->hostPost.data(), planner->post.out, (size_t)n_queries * sizeof(pp_post_result), hipMemcpyDeviceToHost, s));
	PP_HIP_TRY(hipStreamSynchronize(s));
	planner->postDone = n_queries;
	if (results_host)
		for (int i = 0; i < n_queries; i++)
			results_host[i] = planner->hostPost[i];
	return PP_OK;
}

int pp_planner_get_processed_path(pp_planner* planner, int32_t q, double* sampled_host, uint8_t* cusp_host, double* smoothed_host)
{
	if (!planner || q < 0 || q >= planner->postDone) {
		set_error("no post-processed result for this query (pp_planner_postprocess first)");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(planner->map->ctx->device));
	const size_t n = (size_t)planner->hostPost[q].n_points, cap = (size_t)planner->postMaxPoints;
	if (n == 0)
		return PP_OK;
	if (sampled_host)
		PP_HIP_TRY(hipMemcpy(sampled_host, planner->post.resampled + (size_t)q * cap * 3, n * 24, hipMemcpyDeviceToHost));
	if (cusp_host)
		PP_HIP_TRY(hipMemcpy(cusp_host, planner->post.cusp + (size_t)q * cap, n, hipMemcpyDeviceToHost));
	if (smoothed_host)
		PP_HIP_TRY(hipMemcpy(smoothed_host, planner->post.smoothed + (size_t)q * cap * 3, n * 24, hipMemcpyDeviceToHost));
	return PP_OK;
}

int pp_planner_last_timings(pp_planner* planner, float* wavefront_ms, float* search_ms)
{
	if (!planner) {
		set_error("null planner");
		return PP_ERR_INVALID;
	}
	if (wavefront_ms)
		*wavefront_ms = planner->wavefrontMs;
	if (search_ms)
		*search_ms = planner->searchMs;
	return PP_OK;
}

int pp_planner_get_path(pp_planner* planner, int32_t q, double* poses_host, int32_t* kind_host, int32_t* prim_host, double* length_host, double* tuv_host)
{
	if (!planner || q < 0 || q >= planner->lastBatch || (int)planner->hostResults.size() <= q) {
		set_error("no fetched result for this query (call pp_planner_fetch_results first)");
		return PP_ERR_INVALID;
	}
	const DevResult& r = planner->hostResults[q];
	if (r.r.status != 0 || r.solutionNode < 0)
		return PP_OK; // empty path
	PP_HIP_TRY(hipSetDevice(planner->map->ctx->device));
	const int n = r.r.n_path;
	if (n > planner->maxPath) {
		set_error("solution path longer than the planner's path buffer");
		return PP_ERR_CAPACITY;
	}
	std::vector<PathRec> recs(n); // goal first (write_path)
	if (n)
		PP_HIP_TRY(hipMemcpy(recs.data(), planner->paths + (size_t)q * planner->maxPath, (size_t)n * sizeof(PathRec), hipMemcpyDeviceToHost));
	std::vector<RsLogEntry> rslog(r.nRsLog);
	if (r.nRsLog)
		PP_HIP_TRY(hipMemcpy(rslog.data(), planner->rsLogs + (size_t)q * kRsLogCap, (size_t)r.nRsLog * sizeof(RsLogEntry), hipMemcpyDeviceToHost));
	for (int i = 0; i < n; i++) {
		const PathRec& nd = recs[n - 1 - i];
		if (poses_host) {
			poses_host[3 * i] = nd.x;
			poses_host[3 * i + 1] = nd.y;
			poses_host[3 * i + 2] = nd.t;
		}
		const int kind = nd.action < 0 ? 0 : (nd.action >= 1000 ? 2 : 1);
		if (kind_host)
			kind_host[i] = kind;
		if (prim_host)
			prim_host[i] = kind == 2 ? nd.action - 1000 : nd.action;
		if (length_host)
			length_host[i] = nd.length;
		if (tuv_host) {
			tuv_host[3 * i] = tuv_host[3 * i + 1] = tuv_host[3 * i + 2] = 0.0;
			if (kind == 2)
				for (const auto& le : rslog)
					if (le.node == nd.node) {
						tuv_host[3 * i] = le.t;
						tuv_host[3 * i + 1] = le.u;
						tuv_host[3 * i + 2] = le.v;
					}
		}
	}
	return PP_OK;
}

int pp_planner_get_expanded(pp_planner* planner, int32_t q, int32_t* cells_host)
{
	if (!planner || q < 0 || q >= planner->lastBatch || (int)planner->hostResults.size() <= q || !cells_host) {
		set_error("no fetched result for this query (call pp_planner_fetch_results first)");
		return PP_ERR_INVALID;
	}
	if (!planner->expanded) {
		set_error("this planner keeps no expansion log (a pipeline created with log_expansions = 0)");
		return PP_ERR_INVALID;
	}
	const DevResult& r = planner->hostResults[q];
	PP_HIP_TRY(hipSetDevice(planner->map->ctx->device));
	const int ne = r.r.n_expanded;
	if (ne == 0)
		return PP_OK;
	std::vector<uint32_t> keys(ne); // packed discrete pose of every expanded node, in expansion order
	PP_HIP_TRY(hipMemcpy(keys.data(), planner->expanded + (size_t)q * planner->maxNodes, (size_t)ne * 4, hipMemcpyDeviceToHost));
	const KeySpace& ks = planner->args.ks;
	for (int i = 0; i < ne; i++) {
		int ix = 0, iy = 0, it = 0;
		if (keys[i] != kNoKey)
			ks.unpack(keys[i], ix, iy, it);
		cells_host[3 * i] = ix;
		cells_host[3 * i + 1] = iy;
		cells_host[3 * i + 2] = it;
	}
	return PP_OK;
}

} // extern "C"

#include "pp_pipeline.hpp"
