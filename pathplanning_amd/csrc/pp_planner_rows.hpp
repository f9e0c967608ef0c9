// k_hybrid_search_rows -- the batched Hybrid-A* graph search with FOUR queries per wave.
//
// Included by pp_planner.hip (inside its anonymous namespace, after Node / SearchArgs / DevResult).
//
// Why: one expansion creates P = 10 children, so the one-query-per-wave kernel keeps 10 of 64 lanes busy, and its
// time per expansion (~12 us) is the same with 64 queries on the GPU as with 4096 (measured): it is bound by the
// instruction stream of a single wave (f64 sincos / Reeds-Shepp / collision march issued for 10 lanes) plus a chain
// of dependent memory round trips, not by memory bandwidth.  Here a query owns one DPP row (16 lanes) and the wave
// runs the expansions of four queries in lock step: the same instruction stream now carries four expansions.
//   * everything the old kernel kept wave-uniform (open-list sizes, node counts, the popped node...) is held
//     replicated in the 16 lanes of the row; control flow diverges only at row granularity, so DPP row operations
//     and ds_bpermute always find their source lanes active;
//   * the sorted front buffer of the open list is 16 entries per query (one per lane, shifted with DPP row_shr /
//     row_shl); what falls off goes to the same 64-ary heap in HBM, popped by 16 lanes x 4 children each;
//   * rows take queries from a device counter as they finish (query lengths differ by 100x), and leave when it runs
//     dry, so the grid is persistent: min(ceil(n/4), resident waves) workgroups of one wave;
//   * the Mersenne-Twister state of a query lives in HBM (2.5 KB per query, one 8-byte read per expansion).
// Pop order, node numbering, RNG draws and every counter are identical to the one-query-per-wave kernel and to the
// oracle (tests/test_gpu_hybrid.py checks the expanded sequence).
#pragma once

#ifndef PP_ROWS_STATS
#define PP_ROWS_STATS 0 // diagnostic build: histogram of busy rows per wave iteration, printed when a planner is destroyed
#endif
#if PP_ROWS_STATS
__device__ unsigned long long g_rowsStats[24];
#endif
#ifndef PP_ROWS_WAVES_PER_WG
#define PP_ROWS_WAVES_PER_WG 4
#endif
#if PP_ROWS_STATS
#undef PP_ROWS_WAVES_PER_WG
#define PP_ROWS_WAVES_PER_WG 1 // the statistics build keeps wave-level sums in workgroup-shared words
#endif
#ifndef PP_ROWS_EAGER_REFILL
#define PP_ROWS_EAGER_REFILL 1
#endif

// --------------------------------------------------------------------------------------------------- kernel --
#ifndef PP_PIPE_POLL_EVERY
#define PP_PIPE_POLL_EVERY 8
#endif
constexpr int kPollEvery = PP_PIPE_POLL_EVERY; // pipeline: an idle row that found the ring empty looks again every kPollEvery-th pass of its wave

template <bool kPiped>
__global__ void __launch_bounds__(64 * PP_ROWS_WAVES_PER_WG, PP_SEARCH_WAVES_PER_SIMD) k_hybrid_search_rows(SearchArgs A, int nQueries, const double* __restrict__ starts,
	const double* __restrict__ goals, const uint64_t* __restrict__ seeds, const float* __restrict__ costFields, Node* __restrict__ nodesBase,
	HeapEntry* __restrict__ heapBase, uint32_t* __restrict__ keymapBase, uint32_t* __restrict__ expandedBase, RsLogEntry* __restrict__ rsLogBase,
	PathRec* __restrict__ pathBase, unsigned long long* __restrict__ mtBase, DevResult* __restrict__ results, int* __restrict__ nextQuery,
	SuspendRec* __restrict__ suspended, const int32_t* __restrict__ order, int suspendAfter, const SuspendRec* __restrict__ resumeList,
	const int* __restrict__ nResumeDev, int* __restrict__ suspendedCount, int* __restrict__ spareCount, int compactBelow, HeapEntry* __restrict__ bandBase, double bandInvW,
	uint8_t* __restrict__ bandMetaBase, PipeView pipe)
{
	// Two uses.  (a) resumeList == nullptr: the rows take the batch's queries (nextQuery[0] counts them, order[] gives the
	// hand-out order); a query that reaches `suspendAfter` expansions is written to suspended[] and its row continues in a
	// spare slot (suspendedCount counts both).  (b) resumeList != nullptr: the rows take the records of that list
	// (*nResumeDev of them) and continue those queries in the records' own slots; one that reaches `suspendAfter` is written
	// to suspended[] again (for the one-query-per-wave kernel).
#if PP_SEARCH_SETPRIO
	// the searches are chains of dependent steps (a batch lasts as long as its longest query), the wavefront kernels they share
	// the SIMDs with are throughput work: search waves issue first
	__builtin_amdgcn_s_setprio(PP_ROWS_PRIO);
#endif
	// The waves of a workgroup are independent (no workgroup barrier anywhere below); they are launched together only so
	// that they land on ONE compute unit: with the longest-first hand-out order the first waves hold the longest queries,
	// and the long-lived waves of a batch then sit on few CUs instead of one CU each (every CU that hosts a search wave
	// has room for only one of the two wavefront workgroups it could run).
	// waves per workgroup: the batch form launches four together so that a batch's long-lived waves share CUs; the pipeline form launches
	// single waves -- its grid is topped up at every submission, and a topped-up workgroup in which one wave found its index free and
	// three found theirs owned holds four waves' LDS and registers for one wave's work (measured: with 4096 rows and more, a third of the
	// grid's rows never became resident again)
	constexpr int kW = kPiped ? 1 : PP_ROWS_WAVES_PER_WG;
	const int lane = threadIdx.x & 63;
	const int waveIdx = (int)blockIdx.x * kW + (int)(threadIdx.x >> 6);
	if (waveIdx >= A.rowsWaves)
		return;
	const int rl = lane & (kRowLanes - 1);
	const int sb = (lane >> 4) * kRowSlots; // first staging slot of this row
	// search buffers (node records, heap, key map, engine state) belong to the ROW, not to the query: the row's
	// queries use them one after the other, so a planner needs them for its resident rows only
	// (a row that hands its query over to the one-query-per-wave kernel leaves the slot to it and takes a spare one)
	// Third use (c), pipe.ctl != nullptr: the grid is the consumer of a streaming pipeline (pp_pipeline.hpp).  Rows take FIELD SLOTS from
	// the ready ring the wavefront kernel appends to (`q` below is then the slot: start / goal / seed, field, path and log buffers are
	// all indexed by it), announce results in a ring in host memory, and the wave leaves when nothing is left to claim; launches
	// only top the grid up -- a wave whose index is still owned by a wave of an earlier launch leaves at once.
	constexpr bool piped = kPiped; // (a separate instantiation: the batch form keeps its registers, the pipeline form drops the hand-over machinery)
	if (piped) {
		int owner = 0;
		if (lane == 0)
			owner = __hip_atomic_compare_exchange_strong(pipe.waveAlive + waveIdx, &owner, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0 : 1;
		if (__builtin_amdgcn_readfirstlane(owner))
			return;
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // the previous owner's last stores to the rows' buffers
	}
	int idleIters = 0; // (wave-uniform) consecutive loop passes with every row idle
	bool boosted = false; // (wave-uniform, pipeline) the wave runs at issue priority 3: it hosts a query past pipe.boostAfter expansions
	unsigned passCount = 0; // (wave-uniform, pipeline) loop passes: a row that found the ring empty looks again every kPollEvery-th pass only
	bool pollNow = true;    // (pipeline) this row looks at the ring on the next pass whatever the pass count (it has just finished a query, or lost a race)
	int slot = waveIdx * kRowsPerWave + (lane >> 4);
	Node* nodes = nodesBase + (size_t)slot * A.maxNodes;
	uint32_t* keymap = keymapBase + (size_t)slot * A.ks.size();
	// heap, engine state and bands are addressed from the slot where they are used (refills, flushes, one draw per expansion): six registers
	// that need not live across the expansion loop
#define ROW_HEAP (heapBase + (size_t)slot * A.maxNodes)
#define ROW_MT (mtBase + (size_t)slot * Mt64::N)
#define ROW_BANDS (bandBase + (size_t)slot * (size_t)(kBands * kBandCap))
	const int firstSpareSlot = A.searchRows; // spare slots follow the rows' own
	bool noSuspend = false; // no spare slot was left for this query

	// staging of the children of the node being expanded (per row); kept until the next expansion so that a child popped
	// right away is read back from LDS instead of HBM
	constexpr int kS = kRowsPerWave * kRowSlots;
	struct WaveLds {
		double c_x[kS], c_y[kS], c_t[kS], c_cost[kS], c_total[kS], c_len[kS], c_h[kS], c_sin[kS], c_cos[kS];
		double f_x[kS], f_y[kS], f_t[kS], f_tot[kS]; // open-list node already in the child's cell (shortcut test)
		HeapEntry spill[kRowsPerWave][kRowLanes];
		// f-bands of the open list (pp_search_device.hpp): entries per ring slot, four u8 counters per word
		uint32_t bandCnt[kRowsPerWave][kBands / 4];
		uint32_t c_key[kS], c_state[kS], f_for[kS];
		float c_d0[kS]; // obstacle distance at the child's pose (< 0: invalid state), see Node::dist0
		int rsChecks[kRowsPerWave];
		double rsPre[kRowsPerWave][24]; // rs::Path::make_prefix of the row's Reeds-Shepp attempt (23 doubles)
		int16_t c_action[kS];
		uint8_t c_flags[kS], c_valid[kS]; // flags: 1 = valid child, 2 = an earlier child of the batch shares its cell
	};
	static_assert(offsetof(WaveLds, bandCnt) % 16 == 0 && sizeof(WaveLds) % 16 == 0, "band counters are copied as uint4");
	__shared__ __attribute__((aligned(16))) WaveLds s_wave[kW];
	WaveLds& W = s_wave[threadIdx.x >> 6];
	double *const c_x = W.c_x, *const c_y = W.c_y, *const c_t = W.c_t, *const c_cost = W.c_cost, *const c_total = W.c_total, *const c_len = W.c_len, *const c_h = W.c_h,
		   *const c_sin = W.c_sin, *const c_cos = W.c_cos, *const f_x = W.f_x, *const f_y = W.f_y, *const f_t = W.f_t, *const f_tot = W.f_tot;
	uint32_t *const c_key = W.c_key, *const c_state = W.c_state, *const f_for = W.f_for;
	float* const c_d0 = W.c_d0;
	uint8_t *const c_flags = W.c_flags, *const c_valid = W.c_valid;
	int16_t* const c_action = W.c_action;
	int* const s_rsChecks = W.rsChecks;
	HeapEntry* const spillBuf = W.spill[lane >> 4];
	uint32_t* const bandCnt = W.bandCnt[lane >> 4];

	const MapView& m = A.m;
	const int P = A.prims.n;
	const int maxNodes = A.maxNodes;

	// ---- per-row state (replicated in the row's lanes)
	int q = -1;
	bool act = false, done = false;
	const float* field = nullptr;
	Pose goal = { 0, 0, 0 };
	int myNode = -1, rsNode = -1;
	int pfNode = -1;               // node whose record was prefetched as the probable next pop
	unsigned long long pf64 = 0ull; // lane k < 11: 8-byte word k of that record
	bool pfDead = false;           // it was replaced (ProcessPossibleShortcut) after the fetch
	FrontLane front;
	front_clear(front);
	int frontCount = 0, heapSize = 0, nSpill = 0;
	int nOutside = 0;              // open-list entries outside the front buffer (bands + heap + spill buffer)
	long long bandLo = 0;          // bottom of the band window (multiple of 4), valid once bandLoSet
	bool bandLoSet = false;
	unsigned long long lowK = ~0ull; // lower bound of everything outside
	uint32_t lowS = ~0u;
	HeapEntry heapTop;
	heapTop.ckey = ~0ull;
	heapTop.nseq = ~0u;
	heapTop.node = 0;
	int nNodes = 0;
	uint32_t seq = 0;
	int mtIdx = 0;
	int nExpanded = 0, nRngDraws = 0, nRsAttempts = 0, nRsLog = 0;
	long long laneStateChecks = 0, lanePathChecks = 0, rsStateChecks = 0, rsPathChecks = 0;

	// Entries that leave the front buffer are staged in LDS (spillBuf) and flushed sixteen at a time: every lane routes
	// one entry to the ring slot of its f-band (a slot's position comes from an LDS atomic on the packed counters), the
	// rare ones outside the window or in a full slot go to the heap one by one.
	auto flush_spills = [&]() { // row-uniform nSpill > 0
		wave_lds_sync();
		bool toHeap = false;
		if (rl < nSpill) {
			const HeapEntry mine = spillBuf[rl];
			const long long B = band_of_key(mine.ckey, bandInvW);
			toHeap = true;
			if (B >= bandLo && B < bandLo + kBands) {
				const int sl = (int)(B & (kBands - 1)), sh = (sl & 3) * 8;
				const uint32_t old = atomicAdd(&bandCnt[sl >> 2], 1u << sh);
				const int pos = (int)((old >> sh) & 0xFFu);
				if (pos < kBandCap) {
					ROW_BANDS[sl * kBandCap + pos] = mine;
					toHeap = false;
				} else {
					atomicSub(&bandCnt[sl >> 2], 1u << sh);
				}
			}
		}
		uint32_t hm = row_bits(__ballot(toHeap), lane);
		if (hm) {
			wave_vmem_sync(); // earlier heap writes
			if (rl == 0) {
				int hs = heapSize;
				for (uint32_t mm = hm; mm; mm &= mm - 1)
					heap_push(ROW_HEAP, hs, spillBuf[__ffs((int)mm) - 1]);
			}
			for (; hm; hm &= hm - 1) {
				const HeapEntry e = spillBuf[__ffs((int)hm) - 1];
				if (heapSize == 0 || heap_before(e, heapTop))
					heapTop = e;
				heapSize++;
			}
		}
		nSpill = 0;
		wave_lds_sync();
		wave_vmem_sync();
	};
	auto spill = [&](const HeapEntry& e) {
		if (!bandLoSet) { // the window starts one cost unit below the first entry that leaves the front buffer
			bandLo = (band_of_key(e.ckey, bandInvW) - 64) & ~3ll;
			bandLoSet = true;
		}
		if (rl == 0)
			spillBuf[nSpill] = e;
		nSpill++;
		nOutside++;
		if (key_before(e.ckey, e.nseq, lowK, lowS)) {
			lowK = e.ckey;
			lowS = e.nseq;
		}
		if (nSpill == kRowLanes)
			flush_spills();
	};
	// An entry joins the front buffer exactly when "front <= everything outside" demands or allows it (see the
	// one-query-per-wave kernel's push_open)
	auto push_open = [&](const HeapEntry& e) {
		bool toFront = true;
		if (frontCount < kRowLanes)
			toFront = nOutside == 0 || key_before(e.ckey, e.nseq, lowK, lowS) ||
				(frontCount > 0 && key_before(e.ckey, e.nseq, row_read64(front.ckey, lane, frontCount - 1), row_read(front.nseq, lane, frontCount - 1)));
		if (toFront) {
			HeapEntry sp;
			if (front_insert_row(front, frontCount, e, rl, lane, sp))
				spill(sp);
		} else {
			spill(e);
		}
	};
	// The front buffer ran empty (row-uniform; nOutside > 0): load the lowest band -- one entry per lane --, sort it in the
	// row, then pull in whatever the heap holds below the buffer's last entry.
	auto refill = [&]() {
		if (nSpill > 0)
			flush_spills();
		// lowest non-empty band: ring scan from the window's bottom, every lane looks at four slots (one word) per step
		long long bAbs = 0x7FFFFFFFFFFFFFFFll;
		const int w0 = (int)(bandLo >> 2);
		for (int step = 0; step < kBands / 64; step++) {
			const uint32_t w = bandCnt[(w0 + step * kRowLanes + rl) & (kBands / 4 - 1)];
			const uint32_t hit = row_bits(__ballot(w != 0u), lane);
			if (hit) {
				const int fl = __ffs((int)hit) - 1;
				const uint32_t wf = row_read(w, lane, fl);
				bAbs = bandLo + 4ll * (step * kRowLanes + fl) + ((__ffs((int)wf) - 1) >> 3);
				break;
			}
		}
		long long loadedTop = bandLo - 1; // highest band that has certainly been emptied
		if (bAbs != 0x7FFFFFFFFFFFFFFFll) {
			const int sl = (int)(bAbs & (kBands - 1)), sh = (sl & 3) * 8;
			const int n = (int)((bandCnt[sl >> 2] >> sh) & 0xFFu);
			HeapEntry e;
			e.ckey = ~0ull;
			e.nseq = ~0u;
			e.node = 0;
			if (rl < n)
				e = ROW_BANDS[sl * kBandCap + rl];
			row_sort_entries(e.ckey, e.nseq, e.node, rl);
			front.ckey = e.ckey;
			front.nseq = e.nseq;
			front.node = e.node;
			frontCount = n;
			nOutside -= n;
			wave_lds_sync();
			if (rl == 0)
				bandCnt[sl >> 2] &= ~(0xFFu << sh);
			wave_lds_sync();
			bandLo = bAbs & ~3ll; // every band below the lowest one was empty: the window moves up
			loadedTop = bAbs;
		}
		// (the bound of the outside part is rebuilt from here: spill() lowers it for every entry the loop below pushes out of a full buffer
		// -- those go back into bands at or below `loadedTop` when the spill buffer fills up, where the band-boundary term further down does
		// not see them.  Without this a later, worse entry could enter the buffer ahead of them: found on a 44 597-expansion query of the
		// full-size batch, whose expansion order left the oracle's at expansion 41 196; tests/test_gpu_fullsize.py)
		lowK = ~0ull;
		lowS = ~0u;
		// heap entries that come before the buffer's last entry (or, with an empty buffer, the heap's best) move in
		while (heapSize > 0 && (frontCount == 0 || key_before(heapTop.ckey, heapTop.nseq, row_read64(front.ckey, lane, frontCount - 1), row_read(front.nseq, lane, frontCount - 1)))) {
			wave_vmem_sync();
			const HeapEntry he = heap_pop_row(ROW_HEAP, heapSize, rl, lane, heapTop);
			wave_vmem_sync();
			nOutside--;
			HeapEntry sp;
			if (front_insert_row(front, frontCount, he, rl, lane, sp))
				spill(sp);
		}
		// lower bound of what is outside now: the heap's best, the start of the first band that was not loaded, the spill buffer
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
		wave_lds_sync();
		for (int i = 0; i < nSpill; i++) {
			const HeapEntry e = spillBuf[i];
			if (key_before(e.ckey, e.nseq, lowK, lowS)) {
				lowK = e.ckey;
				lowS = e.nseq;
			}
		}
	};
	auto set_slot = [&](size_t sl_) {
		slot = (int)sl_;
		nodes = nodesBase + (size_t)slot * A.maxNodes;
		keymap = keymapBase + (size_t)slot * A.ks.size();
	};
	// (status / solution are parameters, not row state: they exist at the call sites only -- four registers less to keep alive across the expansion loop)
	auto finish = [&](const int status, const int solutionNode, const double solutionCost) { // writes the result record of the row's query and frees the row
		const long long nStateChecks = row_sum_i64(laneStateChecks, lane) + rsStateChecks;
		const long long pathChecksPacked = row_sum_i64(lanePathChecks, lane) + rsPathChecks;
		const long long pathChecks = pathChecksPacked & kGuardMask;
		wave_vmem_sync();
		if (rl == 0) {
			DevResult r;
			r.r.n_lattice_boundary_hits = (int32_t)(pathChecksPacked >> kGuardShift);
			r.r.reserved = 0;
			r.r.status = status;
			r.r.n_expanded = nExpanded;
			r.r.n_nodes = nNodes;
			r.r.n_path = 0;
			if (status == 0)
				r.r.n_path = write_path(nodes, solutionNode, pathBase + (size_t)q * A.maxPath, A.maxPath, piped ? pipe.pathHost + (size_t)q * (size_t)(3 * pipe.pathHostCap) : nullptr,
					piped ? pipe.pathHostCap : 0);
			r.r.cost = solutionCost;
			r.r.n_rng_draws = nRngDraws;
			r.r.n_rs_attempts = nRsAttempts;
			r.r.n_state_checks = nStateChecks;
			r.r.n_path_checks = pathChecks;
			r.solutionNode = solutionNode;
			r.nRsLog = nRsLog < kRsLogCap ? nRsLog : kRsLogCap;
			results[q] = r; // (pipeline: the slot's record for pp_planner_postprocess and the other accessors of pp_pipeline_planner() on a held slot)
			if (piped) {
				// path records, logs and the record above reach memory (the host may fetch them once it has seen the announcement), then the
				// completion record goes to the ring in host memory, its stamp last
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				const unsigned long long d = __hip_atomic_fetch_add(&pipe.ctl->doneTail, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				PipeDone* const rec = pipe.done + (d & pipe.doneMask);
				rec->r = r;
				rec->readyTail = __hip_atomic_load(&pipe.ctl->readyTail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				rec->readyHead = __hip_atomic_load(&pipe.ctl->readyHead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				__hip_atomic_store(&rec->stamp, ((d + 1ull) << 32) | (unsigned long long)(uint32_t)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			}
		}
		act = false;
		pollNow = true;
	};

#if PP_ROWS_STATS
	unsigned long long statIter[5] = { 0, 0, 0, 0, 0 }; // wave iterations by number of busy rows (diagnostic build)
	// wave time by code region: the clock is read by the first active lane and booked to the region that just ended
	// (rows diverge, so "last stamp" and the sums are wave-level values in LDS)
	__shared__ unsigned long long s_phase[16], s_tlast;
	if (lane < 16)
		s_phase[lane] = 0ull;
	if (lane == 0)
		s_tlast = clock64();
	wave_lds_sync();
#define ROWS_STAMP(ph)                                                     \
	{                                                                      \
		const unsigned long long now_ = clock64();                        \
		if (lane == __ffsll((long long)__ballot(true)) - 1) {             \
			s_phase[ph] += now_ - s_tlast;                                \
			s_tlast = now_;                                               \
		}                                                                  \
	}
#else
#define ROWS_STAMP(ph)
#endif
	for (;;) {
		// (pipeline) A wave iterates in ~30 us with four busy rows and in ~13 us with one, and a run ends with its longest chains of dependent expansions
		// (the lattice-exhausting queries: 65 k): once a row's query has passed `soloAfter` expansions the wave's other rows take nothing new while the
		// ready ring is short -- rows are idle then anyway (the wavefront stage paces the pipeline), and these are the ones that should be.  With a
		// backlog of soloBacklog fields or more every row claims as before.
		const bool waveLong = piped && pipe.soloAfter > 0 && __ballot(act && nExpanded >= pipe.soloAfter) != 0ull;
		// (pipeline) the same queries from the other side: their chains run at ~13 us per expansion on an empty chip and at ~22 us next to the tile waves and the other
		// search waves of a full one; a wave that hosts one asks for the SIMD's issue slots first (s_setprio 3) until the query ends
		if (piped && pipe.boostAfter > 0) {
			const bool hot = __ballot(act && nExpanded >= pipe.boostAfter) != 0ull;
			if (hot != boosted) {
				boosted = hot;
				if (hot)
					__builtin_amdgcn_s_setprio(3);
				else
					__builtin_amdgcn_s_setprio(PP_ROWS_PRIO);
			}
		}
		// ================= rows without a query take the next one =================
		if (!act && !done) {
			bool none = false;
			if (piped) {
				// the ring's head entry, if it carries the stamp of its position (the wavefront kernel stores an entry after reserving its
				// place, so the head may be reserved but not yet written: then there is nothing to take right now)
				// An idle row's look at the ring is two dependent reads at the L2 (about 2 us) that the BUSY rows of the wave wait for: with the
				// ring empty most of the time (the wavefront kernel paces the pipeline) every pass of a wave with an idle row paid them.  So a
				// row that found the ring empty looks again on every kPollEvery-th pass of its wave (all idle rows of a wave on the same pass); a
				// row that has just finished, or lost the race for an entry that was there, looks on the next pass.  (Measured neutral on the
				// bench's plans/s -- 15.4-15.9 k over 20 steps either way: the busy rows' passes are bound elsewhere -- and kept for the eighth
				// of the atomic traffic on the control block.)
				int got = -1; // -1: nothing there (or not looked), -2: an entry was there and another row took it
				const bool look = pollNow || (passCount % (unsigned)kPollEvery) == 0u;
				if (rl == 0 && look) {
					unsigned long long h = __hip_atomic_load(&pipe.ctl->readyHead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					const unsigned long long e = __hip_atomic_load(pipe.ready + (h & pipe.readyMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					bool take = (uint32_t)(e >> 32) == (uint32_t)(h + 1ull);
					if (take && waveLong) // leave it to a wave without a long query unless fields are piling up
						take = __hip_atomic_load(&pipe.ctl->readyTail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - h >= (unsigned long long)pipe.soloBacklog;
					if (take)
						got = __hip_atomic_compare_exchange_strong(&pipe.ctl->readyHead, &h, h + 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
							? (int)(uint32_t)e : -2;
				}
				q = (int)row_read((uint32_t)got, lane, 0);
				none = q < 0;
				pollNow = q == -2; // an entry was there and another row took it
				if (!none) {
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // the field, and the slot's start / goal / seed
				}
			} else {
				int nq = 0;
				if (rl == 0)
					nq = atomicAdd(nextQuery, 1);
				q = (int)row_read((uint32_t)nq, lane, 0);
				if (!resumeList)
					q += A.directCount; // the first entries of the hand-out order run one query per wave (k_hybrid_search)
				const int nAvail = resumeList ? min(*nResumeDev, A.listCap) : nQueries;
				if (q >= nAvail) {
					done = true;
					none = true;
				}
			}
			if (none) {
			} else if (!piped && resumeList) {
				// ---- continue a suspended query where it stopped, in the slot that holds its nodes / heap / key map / engine
				const SuspendRec rec = resumeList[q];
				q = rec.q;
				set_slot((size_t)rec.slot);
				field = costFields + (size_t)q * A.fieldElems;
				goal = { goals[3 * q], goals[3 * q + 1], wrap_theta(goals[3 * q + 2]) };
				myNode = -1;
				rsNode = -1;
				pfNode = -1;
				pfDead = false;
				noSuspend = false;
				front_clear(front);
				frontCount = 0;
				nSpill = 0;
				heapSize = rec.heapSize;
				heapTop.ckey = ~0ull;
				heapTop.nseq = ~0u;
				heapTop.node = 0;
				if (heapSize > 0)
					heapTop = ROW_HEAP[0];
				// the open list was moved out of the front buffer at suspension: band window and slot counts come back, the
				// lowest possible bound keeps everything out of the empty buffer until the first refill
				{
					const uint4* src = reinterpret_cast<const uint4*>(bandMetaBase + (size_t)slot * (size_t)kBands);
					wave_lds_sync();
					for (int i = rl; i < kBands / 16; i += kRowLanes)
						reinterpret_cast<uint4*>(bandCnt)[i] = src[i];
					wave_lds_sync();
				}
				bandLo = rec.bandLo;
				bandLoSet = true;
				nOutside = rec.nOutside;
				lowK = 0ull;
				lowS = 0u;
				nNodes = rec.nNodes;
				seq = rec.seq;
				nExpanded = rec.nExpanded;
				nRngDraws = rec.nRngDraws;
				nRsAttempts = rec.nRsAttempts;
				nRsLog = rec.nRsLog;
				mtIdx = rec.mtIdx;
				laneStateChecks = rl == 0 ? rec.stateChecks : 0; // the totals so far ride in the row's first lane
				lanePathChecks = rl == 0 ? rec.pathChecks : 0;
				rsStateChecks = rsPathChecks = 0;
				act = true;
			} else {
				if (order)
					q = order[q]; // probable longest first (written by the wavefront kernel's last workgroup)
				field = costFields + (size_t)q * A.fieldElems;
				{ // the slot's key map still holds the previous query of this row
					const size_t n = A.ks.size(), n4 = n / 4;
					const uint4 z = { 0, 0, 0, 0 };
					if ((((uintptr_t)keymap) & 15) == 0) {
						for (size_t i = rl; i < n4; i += kRowLanes)
							reinterpret_cast<uint4*>(keymap)[i] = z;
						for (size_t i = n4 * 4 + rl; i < n; i += kRowLanes)
							keymap[i] = 0;
					} else {
						for (size_t i = rl; i < n; i += kRowLanes)
							keymap[i] = 0;
					}
					wave_vmem_sync();
				}
				// goal / start poses go through the Pose2d constructor on the caller's side (theta wrapped)
				const Pose start = { starts[3 * q], starts[3 * q + 1], wrap_theta(starts[3 * q + 2]) };
				goal = { goals[3 * q], goals[3 * q + 1], wrap_theta(goals[3 * q + 2]) };
				// ---- InitializeSearch, a_star.h:350-364
				myNode = -1;
				rsNode = -1;
				pfNode = -1;
				pfDead = false;
				noSuspend = false;
				front_clear(front);
				frontCount = 0;
				heapSize = 0;
				nSpill = 0;
				nOutside = 0;
				bandLo = 0;
				bandLoSet = false;
				lowK = ~0ull;
				lowS = ~0u;
				wave_lds_sync();
				for (int i = rl; i < kBands / 4; i += kRowLanes)
					bandCnt[i] = 0u;
				wave_lds_sync();
				heapTop.ckey = ~0ull;
				heapTop.nseq = ~0u;
				heapTop.node = 0;
				nNodes = 1;
				seq = 1;
				nExpanded = nRngDraws = nRsAttempts = nRsLog = 0;
				laneStateChecks = lanePathChecks = rsStateChecks = rsPathChecks = 0;
				if (rl == 0)
					Mt64::seed(ROW_MT, seeds[q]);
				mtIdx = Mt64::N; // engine freshly seeded: first draw twists
				double rs_, rc_;
				sincos(start.t, &rs_, &rc_);
				int ix, iy, it;
				const bool startOnBoundary = discretize_pose(start, A.rp.lat, A.rp.headingAlias, ix, iy, it);
				if (rl == 0)
					lanePathChecks += (long long)startOnBoundary << kGuardShift; // (guard band, pp_device.hpp: the count shares this counter's upper bits)
				uint32_t key = kNoKey;
				const bool ok = A.ks.pack(ix, iy, it, key);
				if (rl == 0) {
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
				front_insert_row(front, frontCount, e, rl, lane, sp);
				wave_vmem_sync();
				act = true;
			}
		}
		const unsigned long long actMask = __ballot(act);
		passCount++;
#if PP_ROWS_STATS
		if (lane == 0 && actMask)
			statIter[__popcll(actMask) / kRowLanes]++;
#endif
		ROWS_STAMP(0) // taking queries
		if (!actMask) {
			if (!piped)
				break; // every row has run out of queries
			// ---- pipeline: every row is idle.  Leave when the host says so, when every submitted query has been claimed (a later
			// submission brings its own launch), or after idleTicks without work (a safety net: no wave waits for ever on a producer
			// that cannot run); else wait a little and look at the ring again.
			int leave = 0;
			if (lane == 0) {
				const unsigned long long sub = __hip_atomic_load(&pipe.ctl->nSubmitted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const unsigned long long head = __hip_atomic_load(&pipe.ctl->readyHead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const int stop = __hip_atomic_load(&pipe.ctl->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const bool timedOut = (unsigned long long)idleIters > pipe.idleTicks;
				// (a wave that left because everything submitted had been claimed used to come back with the next submission's top-up launch -- but a
				// top-up launch runs only when the previous launch on its stream has ENDED, and with a few waves of every earlier launch still alive
				// no stream's launch ever ends: measured, 64-step run, 1024 waves alive for 3.8 s, then 245 for the remaining 12 s with 12 000 fields
				// waiting.  So idle waves now stay -- for `lingerTicks`, by default the idle time-out -- until the host says that every result has been
				// polled (`quiesce`), and the top-up launches only replace waves that timed out.)
				const unsigned long long qui = __hip_atomic_load(&pipe.ctl->quiesce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (stop || (head >= sub && ((unsigned long long)idleIters >= pipe.lingerTicks || qui >= sub)) || timedOut) {
					// the wave index goes back first, THEN the submission count is read again: a top-up launch that found this index
					// still owned was started after its submission count was written, so one of the two sees the other
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					const int was = __hip_atomic_exchange(pipe.waveAlive + waveIdx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					leave = 1 + (was & 0); // (the exchange's result is consumed: it has completed before the loads below are issued)
					const unsigned long long sub2 = __hip_atomic_load(&pipe.ctl->nSubmitted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					const unsigned long long head2 = __hip_atomic_load(&pipe.ctl->readyHead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if (!stop && !timedOut && head2 < sub2) {
						int expect = 0;
						if (__hip_atomic_compare_exchange_strong(pipe.waveAlive + waveIdx, &expect, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
							leave = 0; // new work arrived in between and nobody took the index: carry on
					}
				}
			}
			if (__builtin_amdgcn_readfirstlane(leave))
				break;
			idleIters++;
			__builtin_amdgcn_s_sleep(64);
			continue;
		}
		idleIters = 0;
		// compaction trigger (all lanes vote): some row found the queue empty and few rows of the wave are still busy
		const bool compact = !piped && compactBelow > 0 && __ballot(done) != 0ull && __popcll(actMask) <= compactBelow * kRowLanes;
		if (!act)
			continue;

		// ================= one step of SearchPath's main loop (a_star.h:337-345) for every active row =================
		if (!(frontCount > 0 || nOutside > 0)) {
			finish(-1, -1, __builtin_huge_val()); // open list exhausted: status -1
			continue;
		}
		// ---- setting a query aside: its open list goes entirely into the heap, the scalars into a SuspendRec of the next
		// stage's list.  Two triggers: (a) the query reached `suspendAfter` expansions -- a row working on the batch then
		// continues with the next query in a spare slot, a row working on a resume list takes the next record (and its
		// slot); (b) compaction: the queue is empty and at most `compactBelow` rows of this wave are still busy -- a wave
		// costs the same with one busy row as with four, so the leftovers of all such waves are re-packed four per wave
		// by the next stage and this wave ends.
		const bool capHit = !piped && suspendAfter > 0 && nExpanded >= suspendAfter && !noSuspend;
		if (capHit || compact) {
			bool ok = true;
			int sp = 0;
			if (!compact && !resumeList) { // the row goes on: it needs a spare slot
				if (rl == 0)
					sp = atomicAdd(spareCount, 1);
				sp = (int)row_read((uint32_t)sp, lane, 0);
				ok = sp < A.extraSlots;
			}
			if (!ok) {
				noSuspend = true;
			} else {
				int ns = 0;
				if (rl == 0)
					ns = atomicAdd(suspendedCount, 1);
				ns = (int)row_read((uint32_t)ns, lane, 0); // < listCap: at most one record per row and spare slot
				if (nSpill > 0)
					flush_spills(); // the front's entries need the spill buffer
				if (rl < frontCount) {
					HeapEntry e;
					e.ckey = front.ckey;
					e.nseq = front.nseq;
					e.node = front.node;
					spillBuf[rl] = e;
				}
				if (frontCount > 0 && !bandLoSet) {
					bandLo = (band_of_key(row_read64(front.ckey, lane, 0), bandInvW) - 64) & ~3ll;
					bandLoSet = true;
				}
				nSpill = frontCount;
				nOutside += frontCount;
				frontCount = 0;
				wave_lds_sync();
				if (nSpill > 0)
					flush_spills();
				{
					uint4* dst = reinterpret_cast<uint4*>(bandMetaBase + (size_t)slot * (size_t)kBands);
					for (int i = rl; i < kBands / 16; i += kRowLanes)
						dst[i] = reinterpret_cast<const uint4*>(bandCnt)[i];
				}
				const long long sc = row_sum_i64(laneStateChecks, lane) + rsStateChecks, pc = row_sum_i64(lanePathChecks, lane) + rsPathChecks;
				if (rl == 0 && ns < A.listCap) {
					SuspendRec r;
					r.q = q;
					r.slot = (int32_t)slot;
					r.heapSize = heapSize;
					r.nNodes = nNodes;
					r.nExpanded = nExpanded;
					r.nRngDraws = nRngDraws;
					r.nRsAttempts = nRsAttempts;
					r.nRsLog = nRsLog;
					r.mtIdx = mtIdx;
					r.seq = seq;
					r.stateChecks = sc;
					r.pathChecks = pc;
					r.bandLo = bandLo;
					r.nOutside = nOutside;
					r.pad = 0;
					suspended[ns] = r;
				}
				if (compact) {
					done = true; // nothing left to fetch: the wave ends once its rows have stored their records
				} else if (!resumeList) {
					set_slot((size_t)firstSpareSlot + (size_t)sp);
				}
				wave_vmem_sync();
				act = false;
				continue;
			}
		}
		ROWS_STAMP(1) // set-aside logic
		if (frontCount == 0)
			refill();
		const HeapEntry top = front_pop_row(front, frontCount, lane); // the front buffer holds the globally best entries
#if PP_ROWS_EAGER_REFILL
		if (frontCount == 0 && nOutside > 0)
			refill(); // now rather than at the next pop: the prefetch below then knows the probable next node
#endif
		ROWS_STAMP(2) // pop + refill
		const int ni = (int)top.node;
		// ---- the popped node: from the staging of the previous expansion when it is one of its children, else from the
		// prefetch of the probable next pop, else from HBM
		double px, py, pt, pPathCost, pH, pSin, pCos;
		uint32_t pKey;
		float pDist0;
		bool pDead = false;
		{
			const uint32_t hit = row_bits(__ballot(myNode == ni), lane);
			const int slot = hit ? (__ffs((int)hit) - 1) : (rsNode == ni ? kRowRs : -1);
			if (slot >= 0) {
				px = c_x[sb + slot];
				py = c_y[sb + slot];
				pt = c_t[sb + slot];
				pPathCost = c_cost[sb + slot];
				pH = c_h[sb + slot];
				pSin = c_sin[sb + slot];
				pCos = c_cos[sb + slot];
				pKey = c_key[sb + slot];
				pDist0 = c_d0[sb + slot];
			} else if (ni == pfNode) {
				// lane k of the row holds the k-th 8-byte word of the record (see the prefetch below)
				auto d64 = [&](int k) { return __longlong_as_double((long long)row_read64(pf64, lane, k)); };
				px = d64(0);
				py = d64(1);
				pt = d64(2);
				pPathCost = d64(3);
				pH = d64(6);
				pSin = d64(7);
				pCos = d64(8);
				pKey = (uint32_t)(row_read64(pf64, lane, 9) >> 32);                     // { parent, key }
				pDead = pfDead || ((row_read64(pf64, lane, 10) >> 16) & 0xFFull) != 0ull; // { action, dead, pad, dist0 }
				pDist0 = __uint_as_float((uint32_t)(row_read64(pf64, lane, 10) >> 32));
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
		wave_lds_sync(); // staging is about to be overwritten
		if (pDead)
			continue; // entry of a node replaced by ProcessPossibleShortcut
		const Pose ppose = { px, py, pt };
		if (identical_poses(ppose, goal)) { // IsSolution, hybrid_a_star.h:193-196
			finish(0, ni, pPathCost);
			continue;
		}
		// ---- Expand, a_star.h:377-409
		if (rl == 0) {
			if (pKey != kNoKey)
				keymap[pKey] = kExplored; // children in the parent's own cell are caught by a key compare below
			if (!piped || expandedBase) // (a pipeline keeps the log only when asked to)
				(expandedBase + (size_t)q * maxNodes)[nExpanded] = pKey; // packed discrete pose of the expanded node
		}
		rsNode = -1;
		nExpanded++;
		int pix, piy, pit;
		discretize_pose(ppose, A.rp.lat, A.rp.headingAlias, pix, piy, pit);
		const double hCost = pH; // RS gate input (hybrid_a_star.cpp:81): computed when the node was created
		// the raw 64-bit draw the RS gate may need is fetched now (one word of the query's engine state in HBM)
		unsigned long long mtRaw = 0ull;
		const bool gateDraws = !(hCost < 10.0);
		if (gateDraws && mtIdx < Mt64::N)
			mtRaw = ROW_MT[mtIdx];

		bool capacity = false;
		ROWS_STAMP(3) // node record, solution test, bookkeeping
		// ---- constant-steer children, reference order p = 2*deltaIndex + direction (hybrid_a_star.cpp:65-77)
		for (int base = 0; base < P && !capacity; base += kRowLanes) {
			const int p = base + rl;
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
				lanePathChecks += (long long)discretize_pose(child, A.rp.lat, A.rp.headingAlias, ix, iy, it) << kGuardShift;
				ROWS_STAMP(11) // endpoint
				// look-ups of the full-length child are issued before the validity march so that their latency
				// overlaps it (they are redone only when the arc gets truncated)
				bool packed = A.ks.pack(ix, iy, it, key);
				if (packed)
					st = keymap[key];
				HeurLoads hl;
				combined_heuristic_issue(A.heur, m, field, goal, child, cs, cc, hl);
				ROWS_STAMP(12) // key map + heuristics
				// Voronoi term of the full-length arc: its only map read (the last sample, Q8) is issued with the look-ups
				float voroRaw;
				voronoi_cost_issue(m, a, A.rp.voroDiagRes, voroRaw);
				ROWS_STAMP(13) // Voronoi term
				float lastValidRatio;
				int checks = 0;
				ok = true;
				lanePathChecks++;
				// validity / distance of the child's own pose: the first march sample of ITS children (not a counted check)
				float cd0;
				const bool cIn = is_state_valid_issue(m, child.x, child.y, child.t, cd0);
				ROWS_STAMP(4) // child's own validity
				const bool pathValid = is_path_valid_from(m, a, a.init, pDist0, lastValidRatio, checks);
				ROWS_STAMP(5) // validity march
				// the values the look-ups above fetched (loaded under the march)
				hh = combined_heuristic_finish(A.heur, hl);
				const double voroFull = voronoi_cost_finish(voroRaw, A.rp.voroDiagRes, A.rp.voronoiMult);
				d0 = is_state_valid_finish(m, cIn, cd0) ? cd0 : -1.0f;
				if (!pathValid) {
					// PathConstantSteer::Truncate, paths/path_constant_steer.cpp:16-20
					child = a.interpolate_sc((double)lastValidRatio, cs, cc);
					a.length *= (double)lastValidRatio;
					lanePathChecks += (long long)discretize_pose(child, A.rp.lat, A.rp.headingAlias, ix, iy, it) << kGuardShift;
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
			ROWS_STAMP(6) // truncation, costs
			if (ok && key == pKey)
				st = kExplored; // the parent's cell was marked explored just above (a_star.h:381)
			// open-list node already in this child's cell: its pose / cost (needed by ProcessPossibleShortcut) is fetched by
			// the child's own lane, all lanes at once, instead of one dependent load per child in the loop below
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
				// probable next pop (head of the front buffer or of the heap): lane k < 11 fetches 8-byte word k of its record
				int cand = -1;
				if (frontCount > 0)
					cand = (int)row_read(front.node, lane, 0);
				pfNode = cand;
				pfDead = false;
				if (cand >= 0 && rl < 11)
					pf64 = reinterpret_cast<const unsigned long long*>(nodes + cand)[rl];
			}
			// does an EARLIER valid child of this batch share my cell?  (then my prefetched state may be stale)
			const bool dup = row_earlier_same(key, ok, rl);
			// staging: read by the insertion loop below and by the pop that follows
			c_key[sb + rl] = key;
			c_state[sb + rl] = st;
			c_flags[sb + rl] = (uint8_t)((ok ? 1 : 0) | (dup ? 2 : 0));
			c_x[sb + rl] = child.x;
			c_y[sb + rl] = child.y;
			c_t[sb + rl] = child.t;
			c_cost[sb + rl] = gcost;
			c_total[sb + rl] = total;
			c_len[sb + rl] = len;
			c_h[sb + rl] = hh;
			c_sin[sb + rl] = cs;
			c_cos[sb + rl] = cc;
			c_d0[sb + rl] = d0;
			f_x[sb + rl] = fpx;
			f_y[sb + rl] = fpy;
			f_t[sb + rl] = fpt;
			f_tot[sb + rl] = fptot;
			f_for[sb + rl] = fpFor;
			myNode = -1;
			wave_lds_sync();
			ROWS_STAMP(7) // open-list node of the cell, prefetch, duplicate test, staging
			// ---- insertion in child order (a_star.h:391-402 + hybrid_a_star.h:199-205)
			const int cnt = min(kRowLanes, P - base);
			// Most children change nothing (their cell is explored, or holds an open-list node they do not beat): every
			// lane settles that for its own child, and only the children that push, replace, share a cell with an
			// earlier child of the batch or lack the prefetched record walk the serial path, in child order.
			bool need = false;
			if (rl < cnt && ok) {
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
			// every row walks ITS OWN list of such children, the rows side by side: pass j handles the j-th listed child of
			// each row (the passes of a wave = the longest list, not the union of the lists)
			uint32_t todo = row_bits(__ballot(need), lane);
			while (todo != 0u) { // row-uniform condition: a row leaves when its list is done
				const int c = __ffs((int)todo) - 1;
				todo &= todo - 1u;
				const uint32_t flags = c_flags[sb + c];
				if (!(flags & 1u) || capacity)
					continue;
				const uint32_t ckey = c_key[sb + c];
				uint32_t cst = c_state[sb + c];
				if (flags & 2u) {
					wave_vmem_sync(); // lane 0's key-map writes of this batch
					cst = keymap[ckey];
				}
				const double ctotal = c_total[sb + c];
				bool push = false;
				if (cst == 0u) {
					push = true; // !inFrontier && !inExplored
				} else if (cst != kExplored) {
					// in the open list: replace only if the poses are identical and the new path is strictly cheaper
					const int fi = (int)cst - 1;
					const uint32_t hitf = row_bits(__ballot(myNode == fi), lane);
					Pose fp;
					double ftotal;
					if (hitf) {
						const int fs = __ffs((int)hitf) - 1;
						fp = { c_x[sb + fs], c_y[sb + fs], c_t[sb + fs] };
						ftotal = c_total[sb + fs];
					} else if (f_for[sb + c] == cst) {
						fp = { f_x[sb + c], f_y[sb + c], f_t[sb + c] };
						ftotal = f_tot[sb + c];
					} else {
						wave_vmem_sync();
						const Node fn = nodes[fi];
						fp = { fn.x, fn.y, fn.t };
						ftotal = fn.totalCost;
					}
					const Pose cp = { c_x[sb + c], c_y[sb + c], c_t[sb + c] };
					if (identical_poses(fp, cp) && ftotal > ctotal) {
						if (fi == pfNode)
							pfDead = true;
						if (rl == 0)
							nodes[fi].dead = 1;
						if (myNode == fi)
							myNode = -1; // its staged copy must not be used any more
						push = true;
					}
				}
				if (push) {
					if (nNodes >= maxNodes) {
						capacity = true;
						continue;
					}
					const int idx = nNodes++;
					if (rl == c)
						myNode = idx;
					if (rl == 0)
						keymap[ckey] = (uint32_t)idx + 1u;
					HeapEntry e;
					e.ckey = cost_key(ctotal);
					e.nseq = 0xFFFFFFFFu - seq;
					seq++;
					e.node = (uint32_t)idx;
					push_open(e);
				}
			}
			ROWS_STAMP(8) // insertion
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
		}
		ROWS_STAMP(9) // node records
		if (capacity) {
			finish(-4, -1, __builtin_huge_val());
			continue;
		}

		// ---- Reeds-Shepp analytic expansion, gated (hybrid_a_star.cpp:81-88): the RNG is drawn only when hCost >= 10
		// (short-circuit ||)
		bool tryRs = !gateDraws;
		if (gateDraws) {
			if (mtIdx >= Mt64::N) {
				wave_vmem_sync();
				mt_twist_row(ROW_MT, rl);
				mtIdx = 0;
				mtRaw = ROW_MT[0];
			}
			const double u = Mt64::uniform01(Mt64::temper(mtRaw));
			mtIdx++;
			nRngDraws++;
			tryRs = u < 10.0 / (hCost * hCost);
		}
		if (tryRs) {
			nRsAttempts++;
			// GetOptimalPath (reeds_shepp.cpp:654-683): lane l evaluates words l, l + 16, l + 32
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
			int wword = 0x7FFFFFFF;
			double wt = 0, wu = 0, wv = 0;
			for (int k = 0; k < rs::kNumWords / kRowLanes; k++) {
				const int w = rl + kRowLanes * k;
				double gx, gy, gt, t_, u_, v_;
				rs::goal_variant(rel, w % 4, gx, gy, gt);
				const double length = rs::base_lengths(w / 4, gx, gy, gt, t_, u_, v_);
				if (!(length == rs::inf())) {
					rs::Segment sg;
					rs::word_segment(w, t_, u_, v_, sg);
					const float cst = rs::compute_cost(sg, A.rmin, A.rsRev, A.rsFwd, A.rsSw);
					// NaN and +inf never win a `cost < optimalCost` test; words ascend, so the first strict minimum is kept
					if (cst < __builtin_huge_valf() && cst < wcost) {
						wcost = cst;
						wword = w;
						wt = t_;
						wu = u_;
						wv = v_;
					}
				}
			}
			// first strictly-lowest float cost in word order (costs are >= 0: their bit patterns order like the values)
			const uint32_t bestBits = row_min_u32(__float_as_uint(wcost));
			const bool mine = wword != 0x7FFFFFFF && __float_as_uint(wcost) == bestBits;
			const uint32_t wsel = row_min_u32(mine ? (uint32_t)wword : 0xFFFFFFFFu);
			const int word = wsel == 0xFFFFFFFFu ? -1 : (int)wsel;
			if (word >= 0) {
				const int owner = word & (kRowLanes - 1);
				const double bt = row_read_f64(wt, lane, owner), bu = row_read_f64(wu, lane, owner), bv = row_read_f64(wv, lane, owner);
				// the winner's path is validated by one lane (the adaptive march is sequential)
				if (rl == 0) {
					rs::Path path;
					path.init = ppose;
					rs::word_segment(word, bt, bu, bv, path.seg);
					path.rmin = A.rmin;
					path.length = path.seg.length * A.rmin; // PathSegment::GetLength
					float lastRatio;
					int checks = 0;
					// every sample of the march continues from the stored start of its motion instead of walking the word from its
					// beginning (same operations on the same values: rs::Path::make_prefix)
					double* const pre = W.rsPre[lane >> 4];
					path.make_prefix(pre);
					const rs::PrefixedPath ppath = { path, pre, path.length };
					const bool valid = is_path_valid(m, ppath, path.init, lastRatio, checks);
					c_valid[sb + kRowRs] = 0;
					s_rsChecks[lane >> 4] = checks;
					if (valid) {
						const double pathAndSwitchingCosts = (double)rs::compute_cost(path.seg, A.rmin, A.rsRev, A.rsFwd, A.rsSw); // PathReedsShepp::ComputeCost
						const Pose child = ppath.interpolate(1.0);
						int ix, iy, it;
						lanePathChecks += (long long)discretize_pose(child, A.rp.lat, A.rp.headingAlias, ix, iy, it) << kGuardShift;
						const double voro = voronoi_cost(m, ppath, A.rp.voroDiagRes, A.rp.voronoiMult);
						const double cost = pathAndSwitchingCosts + voro;
						uint32_t key;
						if (A.ks.pack(ix, iy, it, key)) {
							double s_, c_;
							sincos(child.t, &s_, &c_);
							const double hh = combined_heuristic_sc(A.heur, m, field, goal, child, s_, c_);
							c_valid[sb + kRowRs] = 1;
							c_key[sb + kRowRs] = key;
							c_x[sb + kRowRs] = child.x;
							c_y[sb + kRowRs] = child.y;
							c_t[sb + kRowRs] = child.t;
							c_cost[sb + kRowRs] = pPathCost + cost;
							c_total[sb + kRowRs] = (pPathCost + cost) + hh;
							c_len[sb + kRowRs] = path.length;
							c_h[sb + kRowRs] = hh;
							c_sin[sb + kRowRs] = s_;
							c_cos[sb + kRowRs] = c_;
							{
								float rd0;
								c_d0[sb + kRowRs] = is_state_valid(m, child.x, child.y, child.t, rd0) ? rd0 : -1.0f;
							}
							c_state[sb + kRowRs] = keymap[key];
							c_action[sb + kRowRs] = (int16_t)(1000 + word);
						}
					}
				}
				wave_lds_sync();
				rsPathChecks++;
				rsStateChecks += (long long)s_rsChecks[lane >> 4];
				if (c_valid[sb + kRowRs]) {
					const uint32_t ckey = c_key[sb + kRowRs];
					const uint32_t cst = c_state[sb + kRowRs];
					bool push = false;
					if (cst == 0u)
						push = true;
					else if (cst != kExplored) {
						const int fi = (int)cst - 1;
						const uint32_t hitf = row_bits(__ballot(myNode == fi), lane);
						Pose fp;
						double ftotal;
						if (hitf) {
							const int fs = __ffs((int)hitf) - 1;
							fp = { c_x[sb + fs], c_y[sb + fs], c_t[sb + fs] };
							ftotal = c_total[sb + fs];
						} else {
							wave_vmem_sync();
							const Node fn = nodes[fi];
							fp = { fn.x, fn.y, fn.t };
							ftotal = fn.totalCost;
						}
						const Pose cp = { c_x[sb + kRowRs], c_y[sb + kRowRs], c_t[sb + kRowRs] };
						if (identical_poses(fp, cp) && ftotal > c_total[sb + kRowRs]) {
							if (fi == pfNode)
								pfDead = true;
							if (rl == 0)
								nodes[fi].dead = 1;
							if (myNode == fi)
								myNode = -1;
							push = true;
						}
					}
					if (push) {
						if (nNodes >= maxNodes) {
							finish(-4, -1, __builtin_huge_val());
							continue;
						}
						const int idx = nNodes++;
						if (rl == 0) {
							Node nd;
							nd.x = c_x[sb + kRowRs];
							nd.y = c_y[sb + kRowRs];
							nd.t = c_t[sb + kRowRs];
							nd.pathCost = c_cost[sb + kRowRs];
							nd.totalCost = c_total[sb + kRowRs];
							nd.length = c_len[sb + kRowRs];
							nd.h = c_h[sb + kRowRs];
							nd.sinT = c_sin[sb + kRowRs];
							nd.cosT = c_cos[sb + kRowRs];
							nd.parent = ni;
							nd.key = ckey;
							nd.action = c_action[sb + kRowRs];
							nd.dead = 0;
							nd.dist0 = c_d0[sb + kRowRs];
							nodes[idx] = nd;
							keymap[ckey] = (uint32_t)idx + 1u;
							if (nRsLog < kRsLogCap) {
								RsLogEntry le;
								le.node = idx;
								le.word = word;
								le.t = bt;
								le.u = bu;
								le.v = bv;
								(rsLogBase + (size_t)q * kRsLogCap)[nRsLog] = le;
							}
						}
						rsNode = idx;
						nRsLog++;
						HeapEntry e;
						e.ckey = cost_key(c_total[sb + kRowRs]);
						e.nseq = 0xFFFFFFFFu - seq;
						seq++;
						e.node = (uint32_t)idx;
						push_open(e);
					}
				}
				wave_lds_sync();
			}
		}
		ROWS_STAMP(10) // Reeds-Shepp expansion
	}
#if PP_ROWS_STATS
	if (lane == 0)
		for (int i = 1; i < 5; i++)
			atomicAdd(&g_rowsStats[i], statIter[i]);
	wave_lds_sync();
	if (lane < 14)
		atomicAdd(&g_rowsStats[8 + lane], s_phase[lane]);
#endif
#undef ROWS_STAMP
#undef ROW_HEAP
#undef ROW_MT
#undef ROW_BANDS
}
