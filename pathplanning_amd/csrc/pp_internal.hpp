// Internal host-side declarations shared by the .hip translation units of libpphip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/pp_hip.h"
#include "pp_device.hpp"

namespace pph {

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what);

#define PP_HIP_TRY(expr)                       \
	do {                                       \
		hipError_t _e = (expr);                \
		if (_e != hipSuccess)                  \
			return pph::hip_fail(_e, #expr);   \
	} while (0)

constexpr int kMaxPrimitives = 128;
struct PrimTable {
	int n;
	double kappa[kMaxPrimitives];
	double invKappa[kMaxPrimitives]; // 1 / kappa as the host divides it (= the device's own `1 / kappa`, both correctly rounded); unused when |kappa| <= 1e-9
	int8_t backward[kMaxPrimitives];
};

struct RolloutParams {
	double arcLength;        // 1.5 * spatialResolution (hybrid_a_star.cpp:115)
	double spatialRes, angularRes;
	ppd::Resolutions lat;    // the same two with their reciprocals (discretize_pose)
	double forwardMult, reverseMult, voronoiMult;
	float voroDiagRes;       // (float)(resolution * sqrt(2.0)), hybrid_a_star.cpp:38
	int headingAlias;
};

struct NonHoloDesc {
	int nx, ny, na;
	double spatialRes, angularRes, offX, offY;
	double rmin;
	float reverseCost, forwardCost, switchCost;
	double minMult; // min(reverse, forward) as doubles (heuristics.cpp:90)
	int negativeKRead;
};

// ---- launchers (pp_kernels_basic.hip) ------------------------------------
hipError_t launch_d2_to_distance(hipStream_t s, const int32_t* d2, float* dist, int64_t n, float res);
hipError_t launch_occ_to_u8(hipStream_t s, const int32_t* occ, uint8_t* occ8, int64_t n);
hipError_t launch_check_states(hipStream_t s, const ppd::MapView& m, int64_t n, const double* poses, uint8_t* valid);
hipError_t launch_check_states_fused(hipStream_t s, const ppd::MapView& m, int64_t n, uint64_t seed, uint64_t* count);
hipError_t launch_check_arcs(hipStream_t s, const ppd::MapView& m, int64_t n, const double* from, const double* kappa, const double* length, const int32_t* dir,
	uint8_t* valid, float* last);
hipError_t launch_check_segments(hipStream_t s, const ppd::MapView& m, int64_t n, const double* from, const double* to, uint8_t* valid);
hipError_t launch_rollout(hipStream_t s, const ppd::MapView& m, const RolloutParams& rp, const PrimTable& prims, int64_t nParents, const double* parents,
	uint8_t* valid, double* pose, int32_t* key, double* cost, double* length);
hipError_t launch_rs_solve(hipStream_t s, int64_t n, const double* from, const double* to, double rmin, float rev, float fwd, float sw, int32_t* word,
	double* tuv, float* cost, double* segLength);
/// bits[cell / 32] bit (cell % 32) = dist[cell] >= minSafeRadius
hipError_t launch_valid_bits(hipStream_t s, const float* dist, int64_t cells, float minSafeRadius, uint32_t* bits);
hipError_t launch_nonholo_build(hipStream_t s, const NonHoloDesc& d, double* table);
hipError_t launch_knn(hipStream_t s, int64_t nPoints, const double* pts, int64_t nQueries, const double* q, int k, int32_t* idx, double* d2);

// ---- wavefront (pp_wavefront.hip) -----------------------------------------
struct WavefrontWorkspace;
int64_t wavefront_workspace_bytes(int rows, int cols);
/// workgroups of the wavefront kernel that are resident at once on the current device (occupancy API x CUs)
int wavefront_resident_blocks();
/// one empty dispatch of the wavefront kernel (no goal to take): makes the stream's queue allocate the kernel's scratch now;
/// ctlDev: >= 8 zeroed bytes of device memory (error flag, goal counter)
hipError_t warm_up_wavefront(hipStream_t s, const ppd::MapView& m, int32_t* ctlDev);
/// Runs nGoals wavefronts; goalCells[g] = row*cols+col or -1 (goal outside the map -> field stays +inf).
/// orderStartsDev / orderOutDev / doneCounterDev / orderKeysDev[nGoals] (optional, nGoals <= 4096): the last workgroup writes the goal indices ordered by
/// decreasing field value at the start pose (x, y, theta triples) -- the planner's hand-out order; *doneCounterDev must be 0.
/// goalPosesDev (optional): (x, y, theta) triples from which the kernel derives the goal cells itself (goalCellsDev unused);
/// countersZeroed: the caller has already cleared errorFlagDev[0..1] on the stream.
/// tiledOut: costDev is [nGoals][field_tiled_elems] in the 8 x 8-tiled layout of pp_device.hpp (what the search kernel
/// reads); otherwise [nGoals][rows*cols] row-major (the public a8 entry points).
/// Pipeline use of the wavefront kernel (pp_pipeline.hpp): entry i of a launch works on field slot slotList[i] (goal pose and output
/// field are indexed by the slot), and every finished slot is appended to the ready ring the search grid consumes.
constexpr int kSlotBits = 20;                 // pipeline list / ring entries: field slot in the low bits, the slot's generation above
constexpr uint32_t kSlotMask = (1u << kSlotBits) - 1u;
constexpr uint32_t kGenMask = (1u << (31 - kSlotBits)) - 1u; // (entries are non-negative int32)
struct WavefrontPublish {
	const int32_t* slotList = nullptr;
	unsigned long long* readyTail = nullptr; // entries appended so far (absolute)
	unsigned long long* ready = nullptr;     // ring of (position + 1) << 32 | slot
	unsigned long long readyMask = 0;        // ring size - 1 (a power of two)
	int* goalCounter = nullptr;              // the launch's goal counter when it is not the word behind the error flag (the pipeline keeps
	                                         // the error flag in pinned host memory, where no device atomic should go)
	int* exitCounter = nullptr;              // != nullptr: the last workgroup to leave sets goalCounter (and this word) back to 0 for the stream's next
	                                         // launch -- a 4-byte memset in front of every launch is a kernel that waits tens of ms for a free
	                                         // slot on a GPU filled with long-running workgroups
	// Goals that should not wait for their launch's turn (pp_pipeline.hpp, "urgent"): a ring of stamped slot numbers shared by ALL launches of a
	// pipeline -- a workgroup of any launch in flight serves it before it takes the next goal of its own list -- and one claim word per slot
	// (generation << 1 -> generation << 1 | 1 by whoever builds the slot's field: a slot sits in its launch's list AND, if urgent, in the ring;
	// list and ring entries are slot | generation << kSlotBits, see k_wavefront's hand-out).
	unsigned long long* urgent = nullptr;
	unsigned long long* urgentHead = nullptr; // entries claimed so far (absolute)
	unsigned long long urgentMask = 0;
	int* claimed = nullptr;
	// The tile form of the wavefront (pp_wavefront_tiles.hip) builds the fields when the caller provides its control words; the goals it cannot
	// certify (a tie of its fixed-point equation, a run that does not settle) are rebuilt by the ordered kernel, launched behind it on the same stream.
	const uint64_t* occBits = nullptr; // pp_map::occBits: the occupancy as padded bit rows (word (row + 1) * wpr + col / 64 + 1, bit col % 64; one word of padding on
	                                   // every side, everything outside the map occupied; occ_bits_dims), rebuilt with occ8
	uint32_t* tilesQueue = nullptr;   // optional: tilesQueueWaves regions of wavefront_tiles_queue_words() words, one per wave of a launch (the tile queue in global memory)
	int tilesQueueWaves = 0;
	int* tilesCtl = nullptr;          // >= 8 ints, zero at allocation (the kernels set them back): tile goal counter, exit counter, handed-over count, ordered goal counter, exit counter
	int32_t* tilesFallback = nullptr; // [>= number of goals a launch may take] the handed-over goals
	unsigned long long* tilesStats = nullptr; // optional, 16 words: goals, tile visits, rounds, candidate passes, cells, handed over, wave cycles, ...
	// optional: the ordered kernel's launch over the handed-over goals goes to this stream, behind `fallbackEvent` recorded on the launching
	// stream -- so that the launching stream's NEXT tile launch does not wait for it (a dozen goals of 4096 take the ordered kernel 70-110 ms on
	// a busy chip: a third of a wavefront stream's time when it ran in line).  The caller keeps tilesCtl / tilesFallback / the workspace
	// untouched until that launch has finished.
	hipStream_t fallbackStream = nullptr;
	hipEvent_t fallbackEvent = nullptr;
	// (set by launch_wavefront for the ordered kernel's launch over the handed-over goals)
	const int* nGoalsDev = nullptr;   // the launch's number of goals lives on the device
	int* resetOnExit = nullptr;       // one more word the last workgroup sets back to 0
	bool agentPoseLoads = false;      // read goal poses with agent-scope loads although no claim words are in use
};
/// occupancy bits of the tile form (WavefrontPublish::occBits): words per row and rows of the padded word grid
void occ_bits_dims(int rows, int cols, int& wpr, int& nWordRows);
hipError_t launch_occ_bits(hipStream_t s, const uint8_t* occ8, int rows, int cols, uint64_t* bits);
bool wavefront_tiles_supported(int rows, int cols);
int wavefront_tiles_resident_blocks(int rows, int cols);
/// words of global memory per wave of a launch for the tile queue (0: the map is small enough for the queue to stay in LDS)
size_t wavefront_tiles_queue_words(int rows, int cols);
hipError_t warm_up_wavefront_tiles(hipStream_t s, const ppd::MapView& m, int* ctlDev);
hipError_t launch_wavefront_tiles(hipStream_t s, const ppd::MapView& m, int nGoals, const int32_t* goalCellsDev, float* costDev, bool tiledOut, const double* goalPosesDev,
	const double* orderStartsDev, float* orderKeysDev, const WavefrontPublish& pub);
hipError_t launch_order_by_key(hipStream_t s, int n, const float* keysDev, int32_t* orderOutDev);
/// is the tile form in use (PP_WF_TILES=0 switches it off: every goal through the ordered kernel)
bool wavefront_tiles_enabled();
hipError_t launch_wavefront(hipStream_t s, const ppd::MapView& m, int nGoals, const int32_t* goalCellsDev, float* costDev, void* workspaceDev,
	int64_t workspaceBytesPerSlot, int nSlots, int32_t* errorFlagDev, unsigned long long* profDev = nullptr, bool tiledOut = false,
	const double* goalPosesDev = nullptr, bool countersZeroed = false, const double* orderStartsDev = nullptr, int32_t* orderOutDev = nullptr,
	int* doneCounterDev = nullptr, float* orderKeysDev = nullptr, const WavefrontPublish& pub = WavefrontPublish());

} // namespace pph

struct pp_ctx;
struct pp_map;
namespace pph {
struct GvdReference;
void gvd_reference_free(pp_map* map); // pp_gvd.hip
/// occ8 and occBits from occ32 (every writer of the occupancy grid ends here)
int refresh_occupancy_views(pp_map* map, hipStream_t s);
void ctx_release(pp_ctx* ctx); // drops one reference, frees at zero
void map_release(pp_map* map);
} // namespace pph

// Lifetimes: a map keeps its context alive and a planner its map (reference counts), so the handles may be destroyed
// in any order -- e.g. by a garbage collector that finalises a reference cycle in arbitrary order.  pp_*_destroy drops
// the caller's reference; the object goes when the last dependent has gone.
struct pp_ctx {
	int refs = 1;
	int device = 0;
	hipStream_t stream = nullptr;
	bool ownsStream = false;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

struct pp_map {
	int refs = 1;
	pp_ctx* ctx = nullptr;
	pp_map_desc desc {};
	float minSafeRadius = 1.0f, minInterp = 0.1f;
	int32_t* d2 = nullptr;
	float* dist = nullptr;
	float* pathcost = nullptr;
	uint8_t* occ8 = nullptr;
	uint32_t* validBits = nullptr; // one bit per cell: dist >= minSafeRadius
	uint64_t* occBits = nullptr;   // occupancy as padded bit rows (WavefrontPublish::occBits), rebuilt whenever occ8 is
	// map authoring / field construction on the device (pp_gvd.hip)
	int32_t* occ32 = nullptr;      // occupancy ids as the reference holds them (-1 free)
	uint32_t* obstLabel[2] = { nullptr, nullptr }; // nearest obstacle cell (row << 16 | col), ping-pong
	uint32_t* voroLabel[2] = { nullptr, nullptr }; // nearest Voronoi-edge cell
	int obstResult = 0, voroResult = 0;            // which of the two holds the fixed point
	int32_t* voroD2 = nullptr;
	uint8_t* voroEdge = nullptr;
	int32_t* gvdFlag = nullptr;
	// reference-order field construction (pp_brushfire_host.hpp): the host brushfire's persistent state and the ordered cell
	// edits (cell, value pairs) made since it last ran.  journalReset: the grid was replaced as a whole before the recorded
	// edits (pp_map_upload_occupancy); journalLost: edits were dropped, the next run re-seeds from the device grid.
	pph::GvdReference* gvdRef = nullptr;
	std::vector<int32_t> journal;
	bool journalReset = false, journalLost = false;
	ppd::MapView view() const;
	size_t cells() const { return (size_t)desc.rows * desc.cols; }
};
