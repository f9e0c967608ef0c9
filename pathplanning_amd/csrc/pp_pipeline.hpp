// pp_pipeline -- HybridAStar::SearchPath (algo/hybrid_a_star.cpp:237-257: ObstaclesHeuristic::Update for the goal, then the graph
// search) for a STREAM of independent queries: one persistent search grid per GPU fed by the wavefront kernel through a device-side
// queue, field slots recycled as queries end.
//
// Included by pp_planner.hip (it shares that file's kernels and buffer set).
//
// Why: the batch planner (pp_planner_search_batch_dev) runs "all wavefronts of a batch, then all searches of that batch"; a batch lasts as
// long as its longest query (a query that exhausts the lattice takes ~1 s while the mean takes a few ms), so most of a planner's rows
// idle while its 17 GB of fields wait, and hiding that took eight planners, a refill loop and a measured start stagger in bench.py
// (round 2).  Here there is no batch boundary:
//   * `capacity` field slots (one obstacle-heuristic field, start / goal / seed, path and Reeds-Shepp log each); a submitted query takes
//     a free slot;
//   * the wavefront kernel builds the slots' fields (one launch per submission, two streams so that consecutive launches overlap) and
//     appends every finished slot to the READY RING (agent-scope release, stamped entries);
//   * the rows of the search grid (k_hybrid_search_rows, pp_planner_rows.hpp use (c)) claim slots from the ring as they become free -- a
//     row is busy as long as anything is ready, whatever its neighbours in the wave are doing, and a long query holds one row, not a batch;
//   * a finished query's record goes to a ring in pinned host memory; pp_pipeline_poll hands records out and returns their slots to the
//     free list.
// The search grid is "persistent" only while there is work: a wave leaves when every submitted query has been claimed and its rows are
// idle (or after `idleTicks` without work -- no wave waits for ever on a producer that cannot run), and every submission launches the
// grid again; a wave of the new launch whose index is still owned by an older wave leaves at once (PipeView::waveAlive).  So no host
// thread, no spin on host memory, and a device synchronisation returns as soon as the submitted work is done.
#pragma once

#include <chrono>
#include <cstring>
#include <deque>
#include <unordered_map>

namespace {

/// start / goal / seed of a submission into their field slots.  Queries whose start or goal pose stands closer than `urgentClearance` to an
/// obstacle or to the edge of the state space are ALSO appended to the urgent ring (WavefrontPublish::urgent): on the bench map such poses are 17 % of the queries and
/// carry 17 of the 18 queries per 4096 that exhaust the lattice (65 k expansions, ~0.9 s: tools/study_failures.py) -- the queries that decide
/// when a run ends.  Their fields are built by whatever wavefront launch is running, ahead of the submissions queued before them, so that
/// their long searches start at once instead of after their launch's turn.  Results do not depend on any of this (order of work only).
__global__ void __launch_bounds__(256) k_pipe_scatter(int n, const int32_t* __restrict__ slotList, const double* __restrict__ startsIn, const double* __restrict__ goalsIn,
	const uint64_t* __restrict__ seedsIn, double* __restrict__ starts, double* __restrict__ goals, uint64_t* __restrict__ seeds, MapView m, float urgentClearance, PipeCtl* ctl,
	unsigned long long* __restrict__ urgent, unsigned long long urgentMask, int* __restrict__ claimed)
{
	const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	bool isUrgent = false;
	uint32_t entry = 0;
	if (i < n) {
		entry = (uint32_t)slotList[i]; // slot | generation << kSlotBits
		const size_t s = (size_t)(entry & pph::kSlotMask);
		for (int k = 0; k < 3; k++) {
			starts[3 * s + k] = startsIn[3 * (size_t)i + k];
			goals[3 * s + k] = goalsIn[3 * (size_t)i + k];
		}
		seeds[s] = seedsIn[i];
		if (claimed) {
			claimed[s] = (int)((entry >> pph::kSlotBits) << 1); // this generation, not yet claimed (see k_wavefront's hand-out)
			if (urgentClearance > 0.0f) {
				float clearance = __builtin_huge_valf();
				for (int k = 0; k < 2; k++) {
					const double* p = (k ? goalsIn : startsIn) + 3 * (size_t)i;
					int row, col;
					world_to_cell(m, p[0], p[1], row, col);
					if (inside_map(m, row, col))
						clearance = fminf(clearance, m.dist[(size_t)row * m.cols + col]);
					// the edge of the state space confines the car like a wall (the distance grid knows obstacles only)
					clearance = fminf(clearance, (float)fmin(fmin(p[0] - m.lbx, m.ubx - p[0]), fmin(p[1] - m.lby, m.uby - p[1])));
				}
				isUrgent = clearance < urgentClearance;
			}
		}
	}
	// The slots' inputs and claim words reach memory before an entry can be seen (the consumer may belong to a launch that is already
	// running): ONE release per workgroup, behind a barrier that has waited for every thread's stores.  (A release per urgent thread wrote
	// this XCD's whole L2 back hundreds of times per submission -- the fields under construction are in there -- and made the pipeline
	// slower the more queries were urgent: 12 % at a third of them, 2x at all of them.)
	if (__syncthreads_or(isUrgent)) {
		if (threadIdx.x == 0) {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		__syncthreads();
		if (isUrgent) {
			const unsigned long long t = __hip_atomic_fetch_add(&ctl->urgentTail, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(urgent + (t & urgentMask), ((t + 1ull) << 32) | (unsigned long long)entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

/// Can kernels on the pipeline's other streams start while one stream holds a launch whose workgroups cannot all be placed?  HIP maps streams
/// onto GPU_MAX_HW_QUEUES hardware queues (default 4; fixed when the runtime starts; no API reports it), and a hardware queue hands out its
/// packets in order: behind a launch that is still waiting for room for its last workgroups -- a wavefront launch on a chip full of search
/// waves, the search grid's top-up -- every other stream mapped to that queue waits too (round 2 measured 6.9 k instead of 9.8 k plans/s;
/// with the persistent grid it is the grid's 50 ms idle time-out that lets the other stream through).  The probe makes exactly that
/// situation: `k_queue_probe_hog`, far more workgroups than the chip holds, each spinning until every other stream's `k_queue_probe_set`
/// has run (or 10 ms after the first of them started).  Small kernels on streams that share a queue DO overlap (a probe of eight one-wave
/// kernels passes with GPU_MAX_HW_QUEUES=2), which is why the probe needs the oversubscribed launch.
__global__ void __launch_bounds__(64) k_queue_probe_hog(unsigned int* bits, unsigned int want, unsigned long long* start)
{
	// (more than half a CU's LDS: one workgroup per CU, so the launch stays oversubscribed -- thousands of workgroups pending -- while
	// every CU keeps wave slots free for the other streams' kernels)
	__shared__ volatile unsigned int hold[24 * 1024];
	hold[threadIdx.x] = 0u;
	if (threadIdx.x != 0)
		return;
	const unsigned long long now = wall_clock64(); // 100 MHz
	unsigned long long t0 = 0ull;
	if (__hip_atomic_compare_exchange_strong(start, &t0, now | 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
		t0 = now | 1ull;
	while ((__hip_atomic_load(bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & want) != want) {
		if (wall_clock64() - t0 > 1000000ull) {
			// gave up: the setters whose bits are missing could not start beside this launch
			__hip_atomic_fetch_or(bits + 1, 0x80000000u | (want & ~__hip_atomic_load(bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return;
		}
		__builtin_amdgcn_s_sleep(32);
	}
}
__global__ void __launch_bounds__(64) k_queue_probe_set(unsigned int* bits, unsigned int bit)
{
	if (threadIdx.x == 0)
		__hip_atomic_fetch_or(bits, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kPipeSearchStreams = 4;
constexpr int kPipeWavefrontStreams = 4; // at most; pp_pipeline::nWf of them are used (2 unless PP_PIPE_WF_STREAMS says otherwise)

} // namespace

struct pp_pipeline {
	pp_planner* pl = nullptr; // the buffer set: fields, per-slot inputs / paths / logs, the rows' node / heap / key-map buffers
	int capacity = 0;
	int waves = 0; // waves of the search grid (= rows / 4)
	// device
	PipeCtl* ctl = nullptr;
	unsigned long long* ready = nullptr;
	unsigned long long readyMask = 0;
	int* waveAlive = nullptr;
	unsigned long long* urgent = nullptr; // ring of urgent slots (k_pipe_scatter -> any running wavefront launch), same size as the ready ring
	int* claimed = nullptr;               // [capacity] 0 -> 1 by the workgroup that builds the slot's field
	float urgentClearance = -1.0f;        // [m] queries with a start or goal pose closer than this to an obstacle (or the edge) go through the urgent
	                                      // ring; 0 = none; < 0 = twice the validator's minimum safe radius (2 m with the reference's default)
	int32_t* slotLists = nullptr; // ring of slot lists, one segment per wavefront launch in flight
	size_t slotListCap = 0, slotListPos = 0;
	// the segments of the launches that may not have finished, oldest first, each with an event recorded behind its launch: a segment is
	// not written again before its launch is done (a launch queued behind long ones reads its list late, and with the urgent ring a slot
	// can be built, searched, polled and refilled many times meanwhile -- capacity alone does not bound the entries submitted since)
	struct Segment {
		size_t begin = 0, end = 0;
		hipEvent_t done = nullptr;
	};
	std::deque<Segment> segments;
	std::vector<hipEvent_t> segmentEvents; // spare events
	long long segmentWaits = 0;            // times a submission had to wait for a launch before reusing its segment (diagnostic)
	void* wfWorkspace[kPipeWavefrontStreams] = {};
	uint32_t* tilesQueue[kPipeWavefrontStreams] = {}; // per wavefront stream: the tile queues of a launch's waves in global memory (large maps: pph::wavefront_tiles_queue_words)
	int tilesQueueWaves = 0;
	int32_t* wfCtl[kPipeWavefrontStreams] = {}; // per wavefront stream: {error flag, goal counter, exit counter, ...}
	// The tile form's control words and hand-over lists, one SET per launch in flight: the ordered kernel's launch over a tile launch's
	// handed-over goals runs on `fbStream`, behind the tile launch and beside the wavefront stream's next one, and reads its set until it ends
	// (fbDone[i], recorded behind it; a set is reused only after that).
	static constexpr int kFbSets = 8;
	int32_t* fbCtl[kFbSets] = {};  // 16 ints each, zero at allocation (the kernels set them back)
	int32_t* fbList[kFbSets] = {}; // [capacity]
	hipEvent_t fbDone[kFbSets] = {}, fbAfterTiles[kFbSets] = {};
	bool fbUsed[kFbSets] = {};
	int nextFbSet = 0;
	hipStream_t fbStream = nullptr;
	// pinned host
	PipeDone* done = nullptr;
	unsigned long long doneMask = 0;
	int32_t* slotStage = nullptr;             // staging of the slot lists (same ring positions as slotLists)
	unsigned long long* submittedStage = nullptr; // ring of submission counts on their way to ctl->nSubmitted
	int submittedStagePos = 0;
	int32_t* errFlags = nullptr; // [streams] the wavefront kernels' error flags, in pinned host memory: written by the device, read by poll
	double* pathHost = nullptr;  // [capacity][pathHostCap][3]: the poses of every finished query's solution path, goal first, written by the row that finished it
	int pathHostCap = 192;       // poses per slot in that ring (longer paths: the rest is fetched from the device records); PP_PIPE_PATH_POSES
	unsigned long long lingerTicks = 0; // PP_PIPE_LINGER_MS: how long an idle search wave stays after everything submitted has been claimed (default: the idle time-out)
	unsigned long long quiesced = 0;    // the submission count last written to PipeCtl::quiesce
	bool dead = false;           // a submission failed half way: the pipeline's accounting is no longer trustworthy (every later call fails)
	unsigned long long lastTail = 0, lastHead = 0; // the ready queue's counters as the latest completion record saw them
	// streams
	hipStream_t wfStream[kPipeWavefrontStreams] = {}, searchStream[kPipeSearchStreams] = {}, ctlStream = nullptr;
	hipEvent_t evIngest = nullptr, evCtl = nullptr;
	int nextWf = 0, nextSearch = 0;
	// host bookkeeping
	std::vector<int32_t> freeSlots;
	std::vector<uint64_t> ticketOfSlot;
	std::vector<uint8_t> slotState; // 0 free, 1 in flight, 2 completed and held for the caller
	std::vector<uint32_t> slotGen;  // times the slot has been filled (mod kGenMask + 1, never 0): tags its list / ring entries and its claim word
	std::unordered_map<uint64_t, int32_t> slotOfTicket;
	unsigned long long nSubmitted = 0, doneHead = 0, nTickets = 0;
	std::chrono::steady_clock::time_point lastLaunch {};
	// launch durations (HIP events on the streams the kernels are launched on), harvested by pp_pipeline_poll
	struct Timed {
		hipEvent_t a = nullptr, b = nullptr;
		int kind = 0; // 0 wavefront, 1 search grid
		long long units = 0; // goals of a wavefront launch
	};
	std::vector<Timed> timedFree, timedBusy;
	double wfMs = 0, searchMs = 0, searchMaxMs = 0;
	long long wfLaunches = 0, wfGoals = 0, searchLaunches = 0;
	ppd::MapView lastView {}; // the map view of the last submission (see pp_pipeline_submit_dev)
	bool viewValid = false;
	int boostAfter = 0; // PP_PIPE_BOOST_AFTER: expansions after which a query's wave runs at issue priority 3 (0 = off)
	int soloAfter = 0, soloBacklog = 256; // PP_PIPE_SOLO_AFTER / PP_PIPE_SOLO_BACKLOG: see k_hybrid_search_rows (0 = off, the default: measured neutral, profiles/r04_solo_sweep.txt)
	unsigned long long idleTicks = 250000ull; // idle loop passes of ~4 us: about 1 s.  (50 ms until round 4: shorter than the ~100 ms the first fields of a run take, so the
	                                         // grid's waves left before their first work arrived and came back by the luck of the top-up launches.)  Idle waves leave at once when
	                                         // the host has polled every result (PipeCtl::quiesce), so the time-out only matters when a producer really cannot run.
	int nWf = 2;      // wavefront streams in use: consecutive submissions' launches overlap (the tail of one under the head of the next)
	int wfBlocks = 0; // workgroups per wavefront launch (<= the resident number): the wavefront kernel's share of the chip
};

namespace {

void free_pipeline(pp_pipeline* P)
{
	if (!P)
		return;
	for (hipStream_t s : P->wfStream)
		if (s)
			(void)hipStreamDestroy(s);
	for (hipStream_t s : P->searchStream)
		if (s)
			(void)hipStreamDestroy(s);
	if (P->ctlStream)
		(void)hipStreamDestroy(P->ctlStream);
	if (P->fbStream)
		(void)hipStreamDestroy(P->fbStream);
	for (hipEvent_t ev : P->fbDone)
		if (ev)
			(void)hipEventDestroy(ev);
	for (hipEvent_t ev : P->fbAfterTiles)
		if (ev)
			(void)hipEventDestroy(ev);
	if (P->evIngest)
		(void)hipEventDestroy(P->evIngest);
	if (P->evCtl)
		(void)hipEventDestroy(P->evCtl);
	for (auto& sg : P->segments)
		if (sg.done)
			(void)hipEventDestroy(sg.done);
	for (hipEvent_t ev : P->segmentEvents)
		(void)hipEventDestroy(ev);
	for (auto* v : { &P->timedFree, &P->timedBusy })
		for (auto& t : *v) {
			if (t.a)
				(void)hipEventDestroy(t.a);
			if (t.b)
				(void)hipEventDestroy(t.b);
		}
	void* dev[] = { P->ctl, P->ready, P->waveAlive, P->urgent, P->claimed, P->slotLists, P->wfWorkspace[0], P->wfWorkspace[1], P->wfWorkspace[2], P->wfWorkspace[3], P->tilesQueue[0], P->tilesQueue[1], P->tilesQueue[2], P->tilesQueue[3], P->wfCtl[0], P->wfCtl[1], P->wfCtl[2], P->wfCtl[3], P->fbCtl[0], P->fbCtl[1], P->fbCtl[2], P->fbCtl[3], P->fbCtl[4], P->fbCtl[5], P->fbCtl[6], P->fbCtl[7],
		P->fbList[0], P->fbList[1], P->fbList[2], P->fbList[3], P->fbList[4], P->fbList[5], P->fbList[6], P->fbList[7] };
	for (void* q : dev)
		if (q)
			(void)hipFree(q);
	void* host[] = { P->done, P->slotStage, P->submittedStage, P->errFlags, P->pathHost };
	for (void* q : host)
		if (q)
			(void)hipHostFree(q);
	if (P->pl)
		free_planner(P->pl);
	delete P;
}

/// an event pair around a launch on stream s: take() before the launch, done() after it
pp_pipeline::Timed timed_take(pp_pipeline* P, hipStream_t s, int kind, long long units)
{
	pp_pipeline::Timed t;
	if (!P->timedFree.empty()) {
		t = P->timedFree.back();
		P->timedFree.pop_back();
	} else if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) {
		t.a = t.b = nullptr;
	}
	t.kind = kind;
	t.units = units;
	if (t.a)
		(void)hipEventRecord(t.a, s);
	return t;
}
void timed_done(pp_pipeline* P, hipStream_t s, const pp_pipeline::Timed& t)
{
	if (t.a && t.b) {
		(void)hipEventRecord(t.b, s);
		P->timedBusy.push_back(t);
	}
}
void timed_harvest(pp_pipeline* P)
{
	for (size_t i = 0; i < P->timedBusy.size();) {
		pp_pipeline::Timed& t = P->timedBusy[i];
		if (hipEventQuery(t.b) != hipSuccess) {
			i++;
			continue;
		}
		float ms = 0;
		if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
			if (t.kind == 0) {
				P->wfMs += ms;
				P->wfLaunches++;
				P->wfGoals += t.units;
			} else {
				P->searchMs += ms;
				P->searchLaunches++;
				P->searchMaxMs = ms > P->searchMaxMs ? ms : P->searchMaxMs;
			}
		}
		P->timedFree.push_back(t);
		P->timedBusy[i] = P->timedBusy.back();
		P->timedBusy.pop_back();
	}
}

PipeView pipe_view(const pp_pipeline* P)
{
	PipeView v;
	v.ctl = P->ctl;
	v.ready = P->ready;
	v.readyMask = P->readyMask;
	v.done = P->done;
	v.doneMask = P->doneMask;
	v.waveAlive = P->waveAlive;
	v.idleTicks = P->idleTicks;
	v.lingerTicks = P->lingerTicks;
	v.soloAfter = P->soloAfter;
	v.soloBacklog = P->soloBacklog;
	v.boostAfter = P->boostAfter;
	v.pathHost = P->pathHost;
	v.pathHostCap = P->pathHostCap;
	return v;
}

/// the submission count goes to the device on the control stream, then a launch of the whole grid behind it: waves whose index is
/// free start working, the others leave at once
int pipe_launch_search(pp_pipeline* P)
{
	pp_planner* pl = P->pl;
	// (the staging ring has 64 entries and the copies are asynchronous: the control stream is drained before an entry can come round again --
	// every submission synchronises it anyway; this bounds the top-up launches of a long wait between submissions.  Round 3's crash under
	// rocprofv3 --pmc was an UNBOUNDED stream of such small asynchronous copies issued from the poll path: DESIGN.md 4.10)
	if ((P->submittedStagePos & 31) == 31)
		PP_HIP_TRY(hipStreamSynchronize(P->ctlStream));
	unsigned long long* const src = P->submittedStage + (P->submittedStagePos++ & 63);
	*src = P->nSubmitted;
	PP_HIP_TRY(hipMemcpyAsync(&P->ctl->nSubmitted, src, 8, hipMemcpyHostToDevice, P->ctlStream));
	PP_HIP_TRY(hipEventRecord(P->evCtl, P->ctlStream));
	// a launch starts only when the previous launch on its stream has ended, i.e. when every wave of that launch has left: the top-up goes to
	// a stream that is idle; if every stream still carries live waves there is nothing to top up through (the waves alive keep taking work)
	hipStream_t s = nullptr;
	for (int i = 0; i < kPipeSearchStreams && !s; i++) {
		const int k = (P->nextSearch + i) % kPipeSearchStreams;
		if (hipStreamQuery(P->searchStream[k]) == hipSuccess) {
			s = P->searchStream[k];
			P->nextSearch = (k + 1) % kPipeSearchStreams;
		}
	}
	if (!s) {
		P->lastLaunch = std::chrono::steady_clock::now();
		return PP_OK;
	}
	PP_HIP_TRY(hipStreamWaitEvent(s, P->evCtl, 0));
	constexpr int kWg = 1; // (single-wave workgroups: see k_hybrid_search_rows)
	pl->args.rowsWaves = P->waves;
	pl->args.m = pl->map->view(); // validator tunables may have changed
	const pp_pipeline::Timed tm = timed_take(P, s, 1, 0);
	hipLaunchKernelGGL(k_hybrid_search_rows<true>, dim3((P->waves + kWg - 1) / kWg), dim3(64 * kWg), 0, s, pl->args, 0, pl->dStarts, pl->dGoals, pl->dSeeds, pl->costFields, pl->nodes, pl->heaps,
		pl->keymaps, pl->expanded, pl->rsLogs, pl->paths, pl->mtStates, pl->results, (int*)nullptr, (SuspendRec*)nullptr, (const int32_t*)nullptr, 0, (const SuspendRec*)nullptr,
		(const int*)nullptr, (int*)nullptr, (int*)nullptr, 0, pl->bands, pl->bandInvW, pl->bandMeta, pipe_view(P));
	PP_HIP_TRY(hipGetLastError());
	timed_done(P, s, tm);
	P->lastLaunch = std::chrono::steady_clock::now();
	return PP_OK;
}

} // namespace

extern "C" {

int pp_pipeline_create(pp_map* map, const pp_hybrid_params* params, int32_t capacity, int32_t max_nodes_per_query, int32_t search_rows, int32_t log_expansions, pp_pipeline** out)
{
	if (!map || !params || !out || capacity < 4 || capacity > (int32_t)pph::kSlotMask || max_nodes_per_query < 16 || search_rows < 0) {
		set_error("invalid arguments (4 <= capacity < 2^20)");
		return PP_ERR_INVALID;
	}
	PP_HIP_TRY(hipSetDevice(map->ctx->device));
	auto* P = new pp_pipeline();
	P->capacity = capacity;
	if (search_rows == 0)
		search_rows = 4096; // measured optimum on MI355X with the tile form of the wavefront (profiles/r04_pipeline_sweeps.txt; 2560 with the ordered kernel, round 3)
	if (int rc = create_planner(map, params, capacity, max_nodes_per_query, search_rows, log_expansions ? PlannerUse::PipelineLogged : PlannerUse::Pipeline, &P->pl)) {
		delete P;
		return rc;
	}
	pp_planner* pl = P->pl;
	P->waves = pl->searchRows / kRowsPerWave;
	size_t ring = 4;
	while (ring < (size_t)capacity * 2)
		ring <<= 1;
	P->readyMask = P->doneMask = ring - 1;
	P->slotListCap = (size_t)capacity * 4;
	hipError_t e = hipMalloc((void**)&P->ctl, sizeof(PipeCtl));
	if (e == hipSuccess)
		e = hipMalloc((void**)&P->ready, ring * 8);
	if (e == hipSuccess)
		e = hipMalloc((void**)&P->waveAlive, (size_t)P->waves * 4);
	if (e == hipSuccess)
		e = hipMalloc((void**)&P->slotLists, P->slotListCap * 4);
	if (e == hipSuccess)
		e = hipMalloc((void**)&P->urgent, ring * 8);
	if (e == hipSuccess)
		e = hipMalloc((void**)&P->claimed, (size_t)capacity * 4);
	if (const char* v = getenv("PP_PIPE_URGENT_CLEARANCE")) { // [m]; 0 switches the urgent ring off
		const double x = strtod(v, nullptr);
		if (x >= 0.0 && x < 1.0e6)
			P->urgentClearance = (float)x;
	}
	if (const char* v = getenv("PP_PIPE_WF_STREAMS")) {
		const long x = strtol(v, nullptr, 10);
		if (x >= 1 && x <= kPipeWavefrontStreams)
			P->nWf = (int)x;
	}
	// Experiment switch PP_PIPE_WF_CUS=n: the wavefront streams run on the first n compute units only and the search streams on the others
	// (hipExtStreamCreateWithCUMask): no stage can take the other's LDS or registers.  0 / unset = no masks.
	int wfCus = 0, totalCus = 0;
	{
		int dev = 0;
		hipDeviceProp_t prop;
		if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
			totalCus = prop.multiProcessorCount;
		if (const char* v = getenv("PP_PIPE_WF_CUS")) {
			const long x = strtol(v, nullptr, 10);
			if (x > 0 && x < totalCus && totalCus <= 512)
				wfCus = (int)x;
		}
	}
	auto masked_stream = [&](hipStream_t* out, bool forWavefront) -> hipError_t {
		uint32_t mask[16] = {};
		for (int c = 0; c < totalCus; c++)
			if ((c < wfCus) == forWavefront)
				mask[c >> 5] |= 1u << (c & 31);
		return hipExtStreamCreateWithCUMask(out, (uint32_t)((totalCus + 31) / 32), mask);
	};
	for (int i = 0; i < P->nWf && e == hipSuccess; i++) {
		e = hipMalloc(&P->wfWorkspace[i], (size_t)pl->wfBytesPerSlot * pl->wfSlots);
		if (const size_t qw = pph::wavefront_tiles_queue_words(pl->map->desc.rows, pl->map->desc.cols)) {
			P->tilesQueueWaves = 2048; // (256 CUs x 8 waves of a pack)
			if (e == hipSuccess)
				e = hipMalloc((void**)&P->tilesQueue[i], qw * 4 * (size_t)P->tilesQueueWaves);
		}
		if (e == hipSuccess)
			e = hipMalloc((void**)&P->wfCtl[i], 64);
		if (e == hipSuccess)
			e = hipMemset(P->wfCtl[i], 0, 64);

		if (e == hipSuccess)
			e = wfCus ? masked_stream(&P->wfStream[i], true) : hipStreamCreateWithFlags(&P->wfStream[i], hipStreamNonBlocking);
	}
	for (int i = 0; i < pp_pipeline::kFbSets && e == hipSuccess; i++) {
		e = hipMalloc((void**)&P->fbCtl[i], 64);
		if (e == hipSuccess)
			e = hipMemset(P->fbCtl[i], 0, 64);
		if (e == hipSuccess)
			e = hipMalloc((void**)&P->fbList[i], (size_t)capacity * 4);
		if (e == hipSuccess)
			e = hipEventCreateWithFlags(&P->fbDone[i], hipEventDisableTiming);
		if (e == hipSuccess)
			e = hipEventCreateWithFlags(&P->fbAfterTiles[i], hipEventDisableTiming);
	}
	if (e == hipSuccess)
		e = wfCus ? masked_stream(&P->fbStream, true) : hipStreamCreateWithFlags(&P->fbStream, hipStreamNonBlocking);
	// The wavefront workgroups of a launch in flight stay until its list AND the urgent ring are empty, and launches queue: room on the chip
	// frees rarely and in bursts.  When it does, the waves that top up the search grid and the scatter kernel of a new submission should get
	// it before the next wavefront launch's pending workgroups refill the chip: their streams have the highest priority.  A safeguard, not a
	// measured gain: 14 driver-style runs each way give 16.8-17.4 k with it and 17.0-17.5 k without (PP_PIPE_FLAT_PRIORITY=1); one run in
	// about forty had come in at 12 k before, consistent results, cause not established (profiles/r03_repeat_runs.txt).
	int prioLow = 0, prioHigh = 0;
	(void)hipDeviceGetStreamPriorityRange(&prioLow, &prioHigh);
	const bool flatPriority = getenv("PP_PIPE_FLAT_PRIORITY") != nullptr || prioHigh == prioLow;
	if (flatPriority)
		prioHigh = prioLow = 0;
	for (int i = 0; i < kPipeSearchStreams && e == hipSuccess; i++)
		e = wfCus ? masked_stream(&P->searchStream[i], false) : hipStreamCreateWithPriority(&P->searchStream[i], hipStreamNonBlocking, prioHigh);
	if (e == hipSuccess)
		e = hipStreamCreateWithPriority(&P->ctlStream, hipStreamNonBlocking, prioHigh);
	if (e == hipSuccess)
		e = hipEventCreateWithFlags(&P->evIngest, hipEventDisableTiming);
	if (e == hipSuccess)
		e = hipEventCreateWithFlags(&P->evCtl, hipEventDisableTiming);
	if (e == hipSuccess)
		e = hipHostMalloc((void**)&P->done, ring * sizeof(PipeDone), hipHostMallocDefault);
	if (e == hipSuccess)
		e = hipHostMalloc((void**)&P->slotStage, P->slotListCap * 4, hipHostMallocDefault);
	if (e == hipSuccess)
		e = hipHostMalloc((void**)&P->submittedStage, 64 * 8, hipHostMallocDefault);
	if (e == hipSuccess)
		e = hipHostMalloc((void**)&P->errFlags, 64, hipHostMallocDefault);
	if (const char* v = getenv("PP_PIPE_PATH_POSES")) {
		const long x = strtol(v, nullptr, 10);
		if (x >= 0 && x <= 2048)
			P->pathHostCap = (int)x;
	}
	if (P->pathHostCap > pl->maxPath)
		P->pathHostCap = pl->maxPath;
	if (e == hipSuccess && P->pathHostCap > 0)
		e = hipHostMalloc((void**)&P->pathHost, (size_t)capacity * (size_t)P->pathHostCap * 24, hipHostMallocDefault);
	if (e == hipSuccess)
		e = hipMemset(P->ctl, 0, sizeof(PipeCtl));
	if (e == hipSuccess)
		e = hipMemset(P->ready, 0, ring * 8);
	if (e == hipSuccess)
		e = hipMemset(P->waveAlive, 0, (size_t)P->waves * 4);
	if (e == hipSuccess)
		e = hipMemset(P->urgent, 0, ring * 8);
	if (e == hipSuccess)
		e = hipMemset(P->claimed, 0, (size_t)capacity * 4);
	if (e == hipSuccess)
		e = hipDeviceSynchronize();
	if (e != hipSuccess) {
		free_pipeline(P);
		return pph::hip_fail(e, "pipeline allocation");
	}
	{
		// the precondition the pipeline's speed rests on, checked instead of assumed (see k_queue_probe_hog)
		std::vector<hipStream_t> all;
		for (int i = 0; i < P->nWf; i++)
			all.push_back(P->wfStream[i]);
		for (hipStream_t st : P->searchStream)
			all.push_back(st);
		all.push_back(P->fbStream);
		const size_t nLong = all.size(); // the streams that carry launches which may wait for room
		all.push_back(P->ctlStream);
		unsigned int* probe = nullptr; // {setter bits, "the hog gave up" flag, the hog's start stamp (2 words)}
		e = hipMalloc((void**)&probe, 16);
		// (a stream gets its hardware queue at its first launch, which takes milliseconds: every stream runs one kernel before the clock matters)
		for (size_t j = 0; j < all.size() && e == hipSuccess; j++)
			hipLaunchKernelGGL(k_queue_probe_set, dim3(1), dim3(64), 0, all[j], probe, 1u << j);
		if (e == hipSuccess)
			e = hipDeviceSynchronize();
		int seen[2] = { 1, 1 };
		int blockedBy = -1;
		unsigned int blockedMask = 0;
		for (size_t x = 0; x < nLong && e == hipSuccess && seen[1]; x++) {
			e = hipMemset(probe, 0, 16);
			if (e == hipSuccess)
				e = hipDeviceSynchronize();
			// (the search and control streams have a higher priority than the wavefront streams: a high-priority launch waiting for room holds
			// the lower-priority queues back by design -- measured: it does with any number of hardware queues -- so a launch on a search stream
			// is only asked to let its own class through)
			const bool xHigh = !flatPriority && x >= (size_t)P->nWf && x < (size_t)P->nWf + kPipeSearchStreams;
			unsigned int want = 0;
			for (size_t j = 0; j < all.size(); j++) {
				const bool jHigh = !flatPriority && ((j >= (size_t)P->nWf && j < (size_t)P->nWf + kPipeSearchStreams) || j + 1 == all.size());
				if (j != x && (!xHigh || jHigh))
					want |= 1u << j;
			}
			hipLaunchKernelGGL(k_queue_probe_hog, dim3(16384), dim3(64), 0, all[x], probe, want, reinterpret_cast<unsigned long long*>(probe + 2));
			for (size_t j = 0; j < all.size(); j++)
				if (j != x)
					hipLaunchKernelGGL(k_queue_probe_set, dim3(1), dim3(64), 0, all[j], probe, 1u << j);
			if (e == hipSuccess)
				e = hipDeviceSynchronize();
			unsigned int flags[2] = { 0, 0 };
			if (e == hipSuccess)
				e = hipMemcpy(flags, probe, 8, hipMemcpyDeviceToHost);
			seen[1] = flags[1] == 0 ? 1 : 0; // no workgroup of the oversubscribed launch had to give up: every other stream ran beside it
			if (!seen[1]) {
				blockedBy = (int)x;
				blockedMask = flags[1] & 0x7FFFFFFFu;
			}
		}
		if (probe)
			(void)hipFree(probe);
		if (e != hipSuccess) {
			free_pipeline(P);
			return pph::hip_fail(e, "pipeline queue probe");
		}
		const char* allow = getenv("PP_PIPE_ALLOW_SHARED_QUEUES");
		if (!seen[1] && !(allow && allow[0] == '1')) {
			free_pipeline(P);
			set_error("the pipeline's " + std::to_string(all.size()) + " streams do not run side by side (a launch waiting for room on stream " + std::to_string(blockedBy) +
				" holds back streams 0x" + [&] { char b[16]; snprintf(b, sizeof b, "%x", blockedMask); return std::string(b); }() +
				"; streams: wavefront, 4 x search, hand-over, control): the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware "
				"queues (default 4, fixed when the runtime starts) and a wavefront launch queued behind the persistent search grid would wait for the grid's idle "
				"time-out.  Set GPU_MAX_HW_QUEUES=16 in the environment before the process's first HIP call (pathplanning_amd, bench.py and the tests do), or "
				"PP_PIPE_ALLOW_SHARED_QUEUES=1 to run anyway");
			return PP_ERR_INVALID;
		}
	}
	std::memset(P->done, 0, ring * sizeof(PipeDone));
	std::memset(P->errFlags, 0, 64);
	P->freeSlots.resize((size_t)capacity);
	for (int i = 0; i < capacity; i++)
		P->freeSlots[(size_t)i] = capacity - 1 - i; // slot 0 is handed out first
	P->ticketOfSlot.assign((size_t)capacity, 0);
	P->slotState.assign((size_t)capacity, 0);
	P->slotGen.assign((size_t)capacity, 0u);
	pl->hostResults.resize((size_t)capacity);
	pl->lastBatch = capacity;
	pl->pipelineOwned = true;
	pl->owner = P;
	{
		P->wfBlocks = pl->wfSlots;
		if (const char* b = getenv("PP_PIPE_WF_BLOCKS")) { // tuning: fewer resident wavefront workgroups leave more of the chip to the search grid
			const long x = strtol(b, nullptr, 10);
			if (x >= 1 && x < P->wfBlocks)
				P->wfBlocks = (int)x;
		}
		const char* v = getenv("PP_PIPE_IDLE_MS"); // how long a wave waits for work that does not come before it leaves on its own
		if (v && *v) {
			const long ms = strtol(v, nullptr, 10);
			P->idleTicks = (unsigned long long)(ms < 1 ? 1 : (ms > 10000 ? 10000 : ms)) * 250ull;
		}
		if (const char* ba = getenv("PP_PIPE_BOOST_AFTER")) {
			const long x = strtol(ba, nullptr, 10);
			P->boostAfter = x < 0 ? 0 : (x > 0x7FFFFFFF ? 0x7FFFFFFF : (int)x);
		}
		if (const char* sa = getenv("PP_PIPE_SOLO_AFTER")) {
			const long x = strtol(sa, nullptr, 10);
			P->soloAfter = x < 0 ? 0 : (x > 0x7FFFFFFF ? 0x7FFFFFFF : (int)x);
		}
		if (const char* sb = getenv("PP_PIPE_SOLO_BACKLOG")) {
			const long x = strtol(sb, nullptr, 10);
			P->soloBacklog = x < 0 ? 0 : (x > 0x7FFFFFFF ? 0x7FFFFFFF : (int)x);
		}
		P->lingerTicks = P->idleTicks; // idle waves stay until the idle time-out or until the host has polled everything (see k_hybrid_search_rows)
		if (const char* lg = getenv("PP_PIPE_LINGER_MS")) {
			const double ms = strtod(lg, nullptr);
			P->lingerTicks = (unsigned long long)((ms < 0 ? 0 : (ms > 1000 ? 1000 : ms)) * 250.0);
		}
	}
	*out = P;
	return PP_OK;
}

int pp_pipeline_destroy(pp_pipeline* P)
{
	if (!P)
		return PP_OK;
	(void)hipSetDevice(P->pl->map->ctx->device);
	// waves that still wait for work leave at once; queries in flight are finished first
	const int one = 1;
	(void)hipMemcpyAsync(&P->ctl->stop, &one, 4, hipMemcpyHostToDevice, P->ctlStream);
	(void)hipStreamSynchronize(P->ctlStream);
	for (hipStream_t s : P->wfStream)
		if (s)
			(void)hipStreamSynchronize(s);
	if (P->fbStream)
		(void)hipStreamSynchronize(P->fbStream);
	for (hipStream_t s : P->searchStream)
		if (s)
			(void)hipStreamSynchronize(s);
	free_pipeline(P);
	return PP_OK;
}

int pp_pipeline_capacity(pp_pipeline* P) { return P ? P->capacity : 0; }
int pp_pipeline_search_rows(pp_pipeline* P) { return P ? P->pl->searchRows : 0; }
int pp_pipeline_in_flight(pp_pipeline* P) { return P ? (int)(P->nSubmitted - P->doneHead) : 0; }
int pp_pipeline_free_slots(pp_pipeline* P) { return P ? (int)P->freeSlots.size() : 0; }
pp_planner* pp_pipeline_planner(pp_pipeline* P) { return P ? P->pl : nullptr; }

int pp_pipeline_submit_dev(pp_pipeline* P, int32_t n_queries, const double* starts_dev, const double* goals_dev, const uint64_t* seeds_dev, uint64_t* tickets_out, int32_t* n_accepted)
{
	if (!P || n_queries < 0 || !n_accepted || (n_queries > 0 && (!starts_dev || !goals_dev || !seeds_dev))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	*n_accepted = 0;
	if (P->dead) {
		set_error("the pipeline failed in an earlier submission and must be destroyed");
		return PP_ERR_HIP;
	}
	pp_planner* pl = P->pl;
	PP_HIP_TRY(hipSetDevice(pl->map->ctx->device));
	if (!pl->tableReady)
		if (int rc = pp_planner_set_nonholo_table(pl, nullptr))
			return rc;
	int k = n_queries < (int)P->freeSlots.size() ? n_queries : (int)P->freeSlots.size();
	if (k > (int)(P->slotListCap / 4))
		k = (int)(P->slotListCap / 4);
	if (k == 0)
		return PP_OK;
	// ---- the map as the kernels see it (MapView: pointers, bounds, the validator's tunables) is a launch argument: the persistent grid's waves keep the
	// view they were launched with.  A view that changed since the last submission (pp_map_set_validator, a rebuilt grid) must not meet queries in flight --
	// some would be searched under the old view, some under the new one -- and with none in flight the old grid's waves are waited for first: they leave as
	// soon as they have seen that everything was polled.
	{
		const ppd::MapView now = pl->map->view();
		const ppd::MapView& was = P->lastView;
		const bool same = now.rows == was.rows && now.cols == was.cols && now.res == was.res && now.invRes == was.invRes && now.gx == was.gx && now.gy == was.gy && now.lox == was.lox &&
			now.loy == was.loy && now.lbx == was.lbx && now.lby == was.lby && now.lbt == was.lbt && now.ubx == was.ubx && now.uby == was.uby && now.ubt == was.ubt &&
			now.minSafeRadius == was.minSafeRadius && now.minInterp == was.minInterp && now.dist == was.dist && now.pathcost == was.pathcost && now.occ8 == was.occ8 &&
			now.validBits == was.validBits; // (field by field: the struct has padding)
		if (P->viewValid && !same) {
			if (P->nSubmitted != P->doneHead) {
				set_error("the map's view (validator tunables, grids) changed while " + std::to_string(P->nSubmitted - P->doneHead) + " queries of this pipeline are in flight: poll them first");
				return PP_ERR_INVALID;
			}
			for (int i = 0; i < kPipeSearchStreams; i++)
				PP_HIP_TRY(hipStreamSynchronize(P->searchStream[i]));
		}
		P->lastView = now;
		P->viewValid = true;
	}
	// ---- slots and tickets
	if (P->slotListPos + (size_t)k > P->slotListCap)
		P->slotListPos = 0; // (a segment never wraps)
	while (!P->segments.empty() && hipEventQuery(P->segments.front().done) == hipSuccess) {
		P->segmentEvents.push_back(P->segments.front().done);
		P->segments.pop_front();
	}
	for (const auto& sg : P->segments)
		if (sg.begin < P->slotListPos + (size_t)k && P->slotListPos < sg.end) { // its launch still reads (or has yet to read) these entries
			P->segmentWaits++;
			PP_HIP_TRY(hipEventSynchronize(sg.done));
		}
	pp_pipeline::Segment seg;
	seg.begin = P->slotListPos;
	seg.end = P->slotListPos + (size_t)k;
	if (!P->segmentEvents.empty()) {
		seg.done = P->segmentEvents.back();
		P->segmentEvents.pop_back();
	} else {
		PP_HIP_TRY(hipEventCreateWithFlags(&seg.done, hipEventDisableTiming));
	}
	// from here on a failure leaves slots taken and work half queued: the pipeline is marked dead instead of pretending to account for it
	struct DeadGuard {
		pp_pipeline* P;
		bool armed = true;
		~DeadGuard()
		{
			if (armed)
				P->dead = true;
		}
	} guard { P };
	P->segmentEvents.push_back(seg.done); // (owned by the pool until the launch below has been recorded)
	int32_t* const stage = P->slotStage + P->slotListPos;
	int32_t* const listDev = P->slotLists + P->slotListPos;
	P->slotListPos += (size_t)k;
	for (int i = 0; i < k; i++) {
		const int32_t s = P->freeSlots.back();
		P->freeSlots.pop_back();
		uint32_t& gen = P->slotGen[(size_t)s];
		gen = gen >= pph::kGenMask ? 1u : gen + 1u;
		stage[i] = (int32_t)((uint32_t)s | (gen << pph::kSlotBits));
		P->slotState[(size_t)s] = 1;
		const uint64_t ticket = P->nTickets++;
		P->ticketOfSlot[(size_t)s] = ticket;
		P->slotOfTicket[ticket] = s;
		if (tickets_out)
			tickets_out[i] = ticket;
	}
	// ---- inputs into their slots: on the control stream (never busy for long), so the caller's arrays are free when this returns
	hipStream_t const w = P->wfStream[P->nextWf];
	const int wfIdx = P->nextWf;
	int32_t* const wctl = P->wfCtl[P->nextWf];
	const int fbSet = P->nextFbSet;
	P->nextFbSet = (P->nextFbSet + 1) % pp_pipeline::kFbSets;
	if (P->fbUsed[fbSet])
		PP_HIP_TRY(hipEventSynchronize(P->fbDone[fbSet])); // (eight launches back: done long ago unless the chip is stuck)
	int32_t* const werr = P->errFlags + P->nextWf; // (pinned host memory: the kernel's plain store reaches it, poll reads it)
	// (with the tile form the ordered kernel's launches all run on fbStream, one after the other: one workspace; without it they are the
	// wavefront streams' own launches)
	const bool tilesOn = pl->map->occBits && pph::wavefront_tiles_enabled() && pph::wavefront_tiles_supported(pl->map->desc.rows, pl->map->desc.cols);
	void* const wws = P->wfWorkspace[tilesOn ? 0 : P->nextWf];
	P->nextWf = (P->nextWf + 1) % P->nWf;
	PP_HIP_TRY(hipMemcpyAsync(listDev, stage, (size_t)k * 4, hipMemcpyHostToDevice, P->ctlStream));
	pl->args.m = pl->map->view();
	hipLaunchKernelGGL(k_pipe_scatter, dim3((k + 255) / 256), dim3(256), 0, P->ctlStream, k, listDev, starts_dev, goals_dev, seeds_dev, pl->dStarts, pl->dGoals, pl->dSeeds, pl->args.m,
		P->urgentClearance < 0.0f ? 2.0f * pl->args.m.minSafeRadius : P->urgentClearance, P->ctl, P->urgent, P->readyMask, P->claimed);
	PP_HIP_TRY(hipGetLastError());
	PP_HIP_TRY(hipEventRecord(P->evIngest, P->ctlStream));
	PP_HIP_TRY(hipStreamSynchronize(P->ctlStream)); // (measured: skipping it -- a caller that keeps its arrays alive -- changes nothing, 17.1 k either way)
	// ---- ObstaclesHeuristic::Update for every goal (hybrid_a_star.cpp:249); each finished slot is appended to the ready ring
	PP_HIP_TRY(hipStreamWaitEvent(w, P->evIngest, 0));
	pl->args.m = pl->map->view();
	pph::WavefrontPublish pub;
	pub.slotList = listDev;
	pub.readyTail = &P->ctl->readyTail;
	pub.ready = P->ready;
	pub.readyMask = P->readyMask;
	pub.goalCounter = wctl + 1; // 0 at creation; the last workgroup of every launch sets it back
	pub.exitCounter = wctl + 2;
	pub.claimed = P->claimed;
	pub.occBits = pl->map->occBits;
	pub.tilesCtl = P->fbCtl[fbSet];
	pub.tilesFallback = P->fbList[fbSet];
	pub.tilesQueue = P->tilesQueue[wfIdx];
	pub.tilesQueueWaves = P->tilesQueue[wfIdx] ? P->tilesQueueWaves : 0;
	pub.fallbackStream = P->fbStream;
	pub.fallbackEvent = P->fbAfterTiles[fbSet];
	if (P->urgentClearance != 0.0f) {
		pub.urgent = P->urgent;
		pub.urgentHead = &P->ctl->urgentHead;
		pub.urgentMask = P->readyMask;
	}
	const pp_pipeline::Timed tm = timed_take(P, w, 0, k);
	PP_HIP_TRY(pph::launch_wavefront(w, pl->args.m, k, nullptr, pl->costFields, wws, pl->wfBytesPerSlot, P->wfBlocks, werr, nullptr, /*tiledOut=*/true, /*goalPoses=*/pl->dGoals,
		/*countersZeroed=*/true, nullptr, nullptr, nullptr, nullptr, pub));
	timed_done(P, w, tm);
	PP_HIP_TRY(hipEventRecord(P->fbDone[fbSet], P->fbStream));
	P->fbUsed[fbSet] = true;
	PP_HIP_TRY(hipEventRecord(seg.done, w));
	P->segmentEvents.pop_back();
	P->segments.push_back(seg);
	P->nSubmitted += (unsigned long long)k;
	*n_accepted = k;
	const int rc = pipe_launch_search(P);
	guard.armed = rc != PP_OK;
	return rc;
}

int pp_pipeline_submit(pp_pipeline* P, int32_t n_queries, const double* starts_host, const double* goals_host, const uint64_t* seeds_host, uint64_t* tickets_out, int32_t* n_accepted)
{
	if (!P || n_queries < 0 || !n_accepted || (n_queries > 0 && (!starts_host || !goals_host || !seeds_host))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	*n_accepted = 0;
	int k = n_queries < (int)P->freeSlots.size() ? n_queries : (int)P->freeSlots.size();
	if (k == 0)
		return PP_OK;
	PP_HIP_TRY(hipSetDevice(P->pl->map->ctx->device));
	double *ds = nullptr, *dg = nullptr;
	uint64_t* dz = nullptr;
	hipError_t e = hipMalloc((void**)&ds, (size_t)k * 24);
	if (e == hipSuccess)
		e = hipMalloc((void**)&dg, (size_t)k * 24);
	if (e == hipSuccess)
		e = hipMalloc((void**)&dz, (size_t)k * 8);
	if (e == hipSuccess)
		e = hipMemcpy(ds, starts_host, (size_t)k * 24, hipMemcpyHostToDevice);
	if (e == hipSuccess)
		e = hipMemcpy(dg, goals_host, (size_t)k * 24, hipMemcpyHostToDevice);
	if (e == hipSuccess)
		e = hipMemcpy(dz, seeds_host, (size_t)k * 8, hipMemcpyHostToDevice);
	int rc = e == hipSuccess ? pp_pipeline_submit_dev(P, k, ds, dg, dz, tickets_out, n_accepted) : pph::hip_fail(e, "pp_pipeline_submit");
	for (void* q : { (void*)ds, (void*)dg, (void*)dz })
		if (q)
			(void)hipFree(q);
	return rc;
}

int pp_pipeline_poll(pp_pipeline* P, int32_t max_results, uint64_t* tickets_out, pp_query_result* results_out, int32_t release, int32_t* n_out)
{
	if (!P || max_results < 0 || !n_out || (max_results > 0 && (!tickets_out || !results_out))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	*n_out = 0;
	if (P->dead) {
		set_error("the pipeline failed earlier and must be destroyed");
		return PP_ERR_HIP;
	}
	pp_planner* pl = P->pl;
	int n = 0;
	while (n < max_results) {
		PipeDone* const rec = P->done + (P->doneHead & P->doneMask);
		const unsigned long long stamp = __atomic_load_n(&rec->stamp, __ATOMIC_ACQUIRE);
		if ((uint32_t)(stamp >> 32) != (uint32_t)(P->doneHead + 1ull))
			break; // the next record has not been written yet
		const int32_t slot = (int32_t)(uint32_t)stamp;
		if (slot < 0 || slot >= P->capacity || P->slotState[(size_t)slot] != 1) {
			set_error("pipeline completion ring corrupted");
			return PP_ERR_HIP;
		}
		pl->hostResults[(size_t)slot] = rec->r;
		P->lastTail = rec->readyTail;
		P->lastHead = rec->readyHead;
		tickets_out[n] = P->ticketOfSlot[(size_t)slot];
		results_out[n] = rec->r.r;
		if (release) {
			P->slotOfTicket.erase(P->ticketOfSlot[(size_t)slot]);
			P->slotState[(size_t)slot] = 0;
			P->freeSlots.push_back(slot);
		} else {
			P->slotState[(size_t)slot] = 2;
		}
		P->doneHead++;
		n++;
	}
	*n_out = n;
	timed_harvest(P);
	// the wavefront kernels raise their error flags in pinned host memory: no copy, no HIP call on this path
	if (__atomic_load_n(&P->errFlags[0], __ATOMIC_RELAXED) || __atomic_load_n(&P->errFlags[1], __ATOMIC_RELAXED) || __atomic_load_n(&P->errFlags[2], __ATOMIC_RELAXED) ||
		__atomic_load_n(&P->errFlags[3], __ATOMIC_RELAXED)) {
		// a field could not be built (the ordered kernel's open list outgrew its workspace): that goal's query will never be announced.  The
		// records consumed above ARE returned (*n_out), the pipeline is dead from here on.
		P->dead = true;
		set_error("obstacle-heuristic open list exceeded its workspace: the pipeline must be destroyed (the results returned with this call are valid)");
		return PP_ERR_CAPACITY;
	}
	if (P->nSubmitted == P->doneHead && P->quiesced != P->nSubmitted) {
		// every result has been polled: the idle waves of the search grid may leave at once instead of waiting out their time-out (a device
		// synchronisation then returns promptly).  One 8-byte copy per such transition -- not per poll.
		PP_HIP_TRY(hipSetDevice(pl->map->ctx->device));
		if ((P->submittedStagePos & 31) == 31)
			PP_HIP_TRY(hipStreamSynchronize(P->ctlStream));
		unsigned long long* const src = P->submittedStage + (P->submittedStagePos++ & 63);
		*src = P->nSubmitted;
		PP_HIP_TRY(hipMemcpyAsync(&P->ctl->quiesce, src, 8, hipMemcpyHostToDevice, P->ctlStream));
		P->quiesced = P->nSubmitted;
	}
	if (P->nSubmitted > P->doneHead) {
		// waves that left on their own (no work for idleTicks) are replaced while queries are outstanding
		const auto now = std::chrono::steady_clock::now();
		if (std::chrono::duration_cast<std::chrono::milliseconds>(now - P->lastLaunch).count() >= 100) { // (twice the waves' own idle time-out)
			PP_HIP_TRY(hipSetDevice(pl->map->ctx->device));
			return pipe_launch_search(P);
		}
	}
	return PP_OK;
}

int pp_pipeline_release(pp_pipeline* P, int32_t n, const uint64_t* tickets)
{
	if (!P || n < 0 || (n > 0 && !tickets)) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	for (int i = 0; i < n; i++) {
		auto it = P->slotOfTicket.find(tickets[i]);
		if (it == P->slotOfTicket.end() || P->slotState[(size_t)it->second] != 2) {
			set_error("ticket is not a completed, held query");
			return PP_ERR_INVALID;
		}
		P->slotState[(size_t)it->second] = 0;
		P->freeSlots.push_back(it->second);
		P->slotOfTicket.erase(it);
	}
	return PP_OK;
}

int pp_pipeline_get_paths(pp_pipeline* P, int32_t n, const uint64_t* tickets, int32_t max_poses, double* poses_host, int32_t* n_poses_host, int32_t release)
{
	if (!P || n < 0 || max_poses < 1 || (n > 0 && (!tickets || !poses_host || !n_poses_host))) {
		set_error("invalid arguments");
		return PP_ERR_INVALID;
	}
	pp_planner* pl = P->pl;
	for (int i = 0; i < n; i++) {
		auto it = P->slotOfTicket.find(tickets[i]);
		if (it == P->slotOfTicket.end() || P->slotState[(size_t)it->second] != 2) {
			set_error("ticket is not a completed, held query");
			return PP_ERR_INVALID;
		}
		const int32_t slot = it->second;
		const DevResult& r = pl->hostResults[(size_t)slot];
		double* out = poses_host + (size_t)i * (size_t)max_poses * 3;
		int np = r.r.status == 0 && r.solutionNode >= 0 ? r.r.n_path : 0;
		if (np > pl->maxPath) {
			set_error("solution path longer than the planner's path buffer");
			return PP_ERR_CAPACITY;
		}
		n_poses_host[i] = np;
		const int want = np < max_poses ? np : max_poses; // the first `want` poses of the path, start first = records np-1 .. np-want
		const int inRing = np < P->pathHostCap ? np : P->pathHostCap; // records 0 .. inRing-1 (goal first) arrived with the completion record
		const double* ring = P->pathHost + (size_t)slot * (size_t)P->pathHostCap * 3;
		std::vector<PathRec> rest;
		if (np > inRing || !P->pathHost) { // the ring holds the path's goal end; a longer path's start end comes from the device records (one copy)
			const int first = P->pathHost ? inRing : 0;
			rest.resize((size_t)(np - first));
			if (np - first > 0) {
				PP_HIP_TRY(hipSetDevice(pl->map->ctx->device));
				PP_HIP_TRY(hipMemcpy(rest.data(), pl->paths + (size_t)slot * pl->maxPath + first, (size_t)(np - first) * sizeof(PathRec), hipMemcpyDeviceToHost));
			}
			for (int k = 0; k < want; k++) {
				const int rec = np - 1 - k;
				if (rec >= first) {
					const PathRec& pr = rest[(size_t)(rec - first)];
					out[3 * k] = pr.x;
					out[3 * k + 1] = pr.y;
					out[3 * k + 2] = pr.t;
				} else {
					out[3 * k] = ring[3 * rec];
					out[3 * k + 1] = ring[3 * rec + 1];
					out[3 * k + 2] = ring[3 * rec + 2];
				}
			}
		} else {
			for (int k = 0; k < want; k++) {
				const int rec = np - 1 - k;
				out[3 * k] = ring[3 * rec];
				out[3 * k + 1] = ring[3 * rec + 1];
				out[3 * k + 2] = ring[3 * rec + 2];
			}
		}
	}
	if (release)
		return pp_pipeline_release(P, n, tickets);
	return PP_OK;
}

/// Launch durations seen so far (HIP events on the launching streams; harvested by pp_pipeline_poll) and reset.  Wavefront: one launch per
/// submission.  Search grid: every submission launches the grid, but only waves whose index is free stay -- a launch that tops up
/// a full grid lasts microseconds, the launch that (re)starts it lasts until the work runs out; search_max_ms is the longest.
int pp_pipeline_timings(pp_pipeline* P, double* wavefront_ms_total, int64_t* wavefront_launches, int64_t* wavefront_goals, double* search_ms_total, int64_t* search_launches,
	double* search_max_ms)
{
	if (!P) {
		set_error("null pipeline");
		return PP_ERR_INVALID;
	}
	timed_harvest(P);
	if (wavefront_ms_total)
		*wavefront_ms_total = P->wfMs;
	if (wavefront_launches)
		*wavefront_launches = P->wfLaunches;
	if (wavefront_goals)
		*wavefront_goals = P->wfGoals;
	if (search_ms_total)
		*search_ms_total = P->searchMs;
	if (search_launches)
		*search_launches = P->searchLaunches;
	if (search_max_ms)
		*search_max_ms = P->searchMaxMs;
	P->wfMs = P->searchMs = P->searchMaxMs = 0;
	P->wfLaunches = P->wfGoals = P->searchLaunches = 0;
	return PP_OK;
}

/// Where the queries in flight are, as the row that announced the latest polled result saw the queue's counters (every completion
/// record carries them: no copy, no HIP call): ready = fields built and waiting for a search row; searching = claimed by a row and not yet polled;
/// the rest of pp_pipeline_in_flight() is still with the wavefront kernel (or completed and not yet polled).
/// ready near 0 with rows to spare = the wavefront stage is the bottleneck; a long ready queue = the search grid is.
int pp_pipeline_backlog(pp_pipeline* P, int64_t* ready, int64_t* searching)
{
	if (!P) {
		set_error("null pipeline");
		return PP_ERR_INVALID;
	}
	if (ready)
		*ready = (int64_t)(P->lastTail - P->lastHead);
	if (searching)
		*searching = (int64_t)(P->lastHead - P->doneHead);
	return PP_OK;
}

/// Diagnostics: how many waves of the search grid own their index right now (a blocking copy of the ownership words: not for the hot path).
int pp_pipeline_alive_waves(pp_pipeline* P)
{
	if (!P)
		return -1;
	std::vector<int> a((size_t)P->waves);
	if (hipSetDevice(P->pl->map->ctx->device) != hipSuccess || hipMemcpy(a.data(), P->waveAlive, (size_t)P->waves * 4, hipMemcpyDeviceToHost) != hipSuccess)
		return -1;
	int n = 0;
	for (int v : a)
		n += v != 0;
	return n;
}

/// field slot of a completed query that is still held (polled with release = 0): the index the pp_planner_get_* accessors of
/// pp_pipeline_planner() take
int pp_pipeline_slot_of(pp_pipeline* P, uint64_t ticket)
{
	if (!P)
		return -1;
	auto it = P->slotOfTicket.find(ticket);
	return it == P->slotOfTicket.end() || P->slotState[(size_t)it->second] != 2 ? -1 : it->second;
}

} // extern "C"
