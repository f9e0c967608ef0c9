// TEST INFRASTRUCTURE ONLY -- CPU oracle (see ppo_geometry.hpp header).
//
// Search layer restated from the reference: the open list ("Frontier"), the two
// Hybrid-A* heuristics, the generic A* engine specialised for Hybrid A* and for
// 8-connected grid A*, and RRT / RRT*.
#pragma once

#include "ppo_world.hpp"

#include <cstring>
#include <functional>
#include <memory>
#include <random>
#include <unordered_map>

namespace ppo {

// ---------------------------------------------------------------------------
// utils/frontier.h -- literal restatement (sorted vector + membership set).
// Used on small cases to pin the fast (cost, -seq) heaps below; itself pinned
// against the reference's own header compiled into oracle/_ref (see Makefile).
// ---------------------------------------------------------------------------
template <typename T, typename Compare>
struct SortedFrontier {
	Compare compare;
	std::vector<T> vec; // back() is the top

	explicit SortedFrontier(Compare c = Compare()) :
		compare(c) { }

	// frontier.h:111-129 -- first position p with compare(value, vec[p]) true
	size_t FindInsertPos(const T& value) const
	{
		size_t begin = 0, end = vec.size(), mid = 0;
		while (begin != end) {
			mid = begin + (end - begin) / 2;
			size_t midNext = mid + 1;
			if (compare(value, vec[mid]))
				end = mid;
			else if (midNext != vec.size() ? !compare(value, vec[midNext]) : false)
				begin = midNext;
			else {
				mid = midNext;
				break;
			}
		}
		return mid;
	}
	void Push(const T& e) { vec.insert(vec.begin() + FindInsertPos(e), e); } // frontier.h:39-48 (uniqueness handled by callers)
	T Pop()
	{
		// frontier.h:83-91
		T e = vec.back();
		vec.pop_back();
		return e;
	}
	bool Empty() const { return vec.empty(); }
};

/// Binary min-heap on (cost, -seq): pops the smallest cost, and among equal
/// costs the most recently pushed entry -- exactly the pop order of the
/// reference Frontier (Appendix A Q1).  Entries may be lazily invalidated.
/// Optional trace of open-list traffic (design studies of the device open list): +1 = push, -1 = pop.
struct FrontierEvent {
	int kind;
	double cost;
	unsigned long long seq;
};
inline std::vector<FrontierEvent>*& FrontierTraceSink()
{
	static thread_local std::vector<FrontierEvent>* sink = nullptr;
	return sink;
}

template <typename CostT>
struct LifoHeap {
	struct Entry {
		CostT cost;
		uint64_t seq;
		uint32_t id;
	};
	std::vector<Entry> h;
	uint64_t nextSeq = 0;
	static bool Before(const Entry& a, const Entry& b) { return a.cost < b.cost || (a.cost == b.cost && a.seq > b.seq); }
	void Clear()
	{
		h.clear();
		nextSeq = 0;
	}
	bool Empty() const { return h.empty(); }
	void Push(CostT cost, uint32_t id)
	{
		Entry e { cost, nextSeq++, id };
		if (sizeof(CostT) == sizeof(double)) // the graph search's open list (the obstacle wavefront uses float costs)
			if (auto* t = FrontierTraceSink())
				t->push_back({ 1, (double)cost, (unsigned long long)e.seq });
		size_t i = h.size();
		h.push_back(e);
		while (i > 0) {
			size_t p = (i - 1) / 2;
			if (!Before(e, h[p]))
				break;
			h[i] = h[p];
			i = p;
		}
		h[i] = e;
	}
	Entry Pop()
	{
		Entry top = h[0];
		if (sizeof(CostT) == sizeof(double))
			if (auto* t = FrontierTraceSink())
				t->push_back({ -1, (double)top.cost, (unsigned long long)top.seq });
		Entry last = h.back();
		h.pop_back();
		size_t n = h.size();
		if (n > 0) {
			size_t i = 0;
			for (;;) {
				size_t c = 2 * i + 1;
				if (c >= n)
					break;
				if (c + 1 < n && Before(h[c + 1], h[c]))
					c++;
				if (!Before(h[c], last))
					break;
				h[i] = h[c];
				i = c;
			}
			h[i] = last;
		}
		return top;
	}
	const Entry& Top() const { return h[0]; }
};

// ---------------------------------------------------------------------------
// algo/heuristics.{h,cpp}
// ---------------------------------------------------------------------------
struct HybridParams { // algo/hybrid_a_star.h:29-50
	double wheelbase = 2.6;
	double minTurningRadius = 2.0;
	double directionSwitchingCost = 0.0;
	double reverseCostMultiplier = 1.0;
	double forwardCostMultiplier = 1.0;
	double voronoiCostMultiplier = 1.0;
	unsigned int numGeneratedMotion = 5;
	double spatialResolution = 1.0;
	double angularResolution = 0.0872;
};

/// Behaviour switches for the reference quirks.  Defaults
/// reproduce the Release-build behaviour of the reference.
struct Quirks {
	bool headingAlias = true; // Q6: Pose2<int>::WrapTheta generic instantiation at -O2
	bool negativeKRead = true; // Q7: m_values[i][j][k<0] heap aliasing (row stride 74 doubles)
};

struct NonHolonomicHeuristic { // heuristics.cpp:7-95
	double spatialResolution, angularResolution, minTurningRadius;
	double reverseCostMultiplier, forwardCostMultiplier, directionSwitchingCost;
	unsigned int numSpatialX, numSpatialY, numAngular;
	double offsetX, offsetY;
	std::vector<double> values; // [i][j][k], k fastest, stride numAngular
	Pose2d goal;
	bool negativeKRead = true;

	static std::unique_ptr<NonHolonomicHeuristic> Build(const Pose2d& lb, const Pose2d& ub, const HybridParams& p)
	{
		// heuristics.cpp:36-76
		auto h = std::make_unique<NonHolonomicHeuristic>();
		h->spatialResolution = p.spatialResolution;
		h->angularResolution = p.angularResolution;
		h->minTurningRadius = p.minTurningRadius;
		h->reverseCostMultiplier = p.reverseCostMultiplier;
		h->forwardCostMultiplier = p.forwardCostMultiplier;
		h->directionSwitchingCost = p.directionSwitchingCost;
		const double spatialSizeX = ub.x - lb.x;
		const double spatialSizeY = ub.y - lb.y;
		unsigned int nx = std::ceil(spatialSizeX / p.spatialResolution);
		if (nx % 2 == 0)
			nx++;
		unsigned int ny = std::ceil(spatialSizeY / p.spatialResolution);
		if (ny % 2 == 0)
			ny++;
		h->numSpatialX = nx;
		h->numSpatialY = ny;
		h->numAngular = std::ceil(2 * M_PI / p.angularResolution); // heuristics.cpp:13
		h->offsetX = std::floor(nx / 2.0) * p.spatialResolution; // heuristics.cpp:14
		h->offsetY = std::floor(ny / 2.0) * p.spatialResolution;
		h->values.assign((size_t)nx * ny * h->numAngular, 0.0);
		const Pose2d goal0(0.0, 0.0, 0.0);
		for (int i = 0; i < (int)nx; i++)
			for (int j = 0; j < (int)ny; j++)
				for (int k = 0; k < (int)h->numAngular; k++) {
					Pose2d pose(i * p.spatialResolution - h->offsetX, j * p.spatialResolution - h->offsetY, k * p.angularResolution);
					auto path = rs::GetOptimalPath(pose, goal0, p.minTurningRadius, p.reverseCostMultiplier, p.forwardCostMultiplier, p.directionSwitchingCost);
					h->values[((size_t)i * ny + j) * h->numAngular + k] = path.ComputeCost(p.minTurningRadius, p.reverseCostMultiplier, p.forwardCostMultiplier, p.directionSwitchingCost);
				}
		return h;
	}

	/// m_values[i][j][k] including the out-of-bounds rows for k < 0 (Q7).  With
	/// glibc's sequential `new double[73]` chunks (592 B = 74 doubles):
	/// k == -1 reads the chunk-size word 0x251; k <= -2 reads row j-1 at 74+k;
	/// for j == 0 it reads the row-pointer array (heap addresses, i.e.
	/// denormals ~4.6e-310), modelled as +0.0 -- never observable because the
	/// combined heuristic takes max() with a Euclidean distance >= 0.
	double Lookup(int i, int j, int k) const
	{
		const size_t na = numAngular;
		if (k >= 0)
			return values[((size_t)i * numSpatialY + j) * na + k];
		if (!negativeKRead) // "fixed" mode: wrap the heading bin
			return values[((size_t)i * numSpatialY + j) * na + (k + (int)na)];
		if (k == -1) {
			uint64_t bits = (uint64_t)(na * 8 + 8 + 15) / 16 * 16 | 1; // chunk size | PREV_INUSE = 0x251 for na = 73
			double d;
			std::memcpy(&d, &bits, 8);
			return d;
		}
		if (j >= 1) {
			const int stride = (int)((na * 8 + 8 + 15) / 16 * 16 / 8); // 74 for na = 73
			int kk = stride + k;
			if (kk >= 0 && kk < (int)na)
				return values[((size_t)i * numSpatialY + (j - 1)) * na + kk];
			return 0.0;
		}
		return 0.0;
	}

	double GetHeuristicValue(const Pose2d& state) const
	{
		// heuristics.cpp:78-95
		Pose2d delta = Between(goal, state);
		int i = (int)std::round((delta.x + offsetX) / spatialResolution);
		int j = (int)std::round((delta.y + offsetY) / spatialResolution);
		int k = (int)std::round(delta.theta / angularResolution);
		if (k == (int)numAngular)
			k = 0;
		if (i < 0 || i >= (int)numSpatialX || j < 0 || j >= (int)numSpatialY) {
			double euclideanDistance = Norm(delta.x, delta.y);
			double distanceMultiplier = std::min(reverseCostMultiplier, forwardCostMultiplier);
			return distanceMultiplier * euclideanDistance;
		}
		return Lookup(i, j, k);
	}
};

struct ObstaclesHeuristic { // heuristics.cpp:97-166
	const World* world;
	float diagonalResolution; // heuristics.cpp:98
	float costMultiplier; // heuristics.cpp:99
	std::vector<float> cost;
	std::vector<uint8_t> explored;
	Pose2d goal;

	ObstaclesHeuristic(const World* w, double reverseCostMultiplier, double forwardCostMultiplier) :
		world(w), diagonalResolution(std::sqrt(2) * w->resolution), costMultiplier(std::min(reverseCostMultiplier, forwardCostMultiplier) * w->resolution)
	{
		cost.assign((size_t)w->rows * w->columns, std::numeric_limits<float>::infinity());
		explored.assign((size_t)w->rows * w->columns, 0);
	}

	struct FCell {
		Cell position;
		float value;
	};
	struct CompareCell { // heuristics.h:38-43: lhs > rhs on the value
		bool operator()(const FCell& a, const FCell& b) const { return a.value > b.value; }
	};

	/// heuristics.cpp:106-153 on the literal sorted-vector Frontier (slow; small maps).
	void UpdateLiteral(const Pose2d& goalPose)
	{
		const int R = world->rows, C = world->columns;
		std::fill(cost.begin(), cost.end(), std::numeric_limits<float>::infinity());
		std::fill(explored.begin(), explored.end(), 0);
		Cell start = world->WorldPositionToGridCell(goalPose.x, goalPose.y);
		if (!start.IsValid())
			return;
		SortedFrontier<FCell, CompareCell> frontier;
		std::vector<uint8_t> inFrontier((size_t)R * C, 0);
		frontier.Push({ start, 0.0f });
		inFrontier[(size_t)start.row * C + start.col] = 1;
		cost[(size_t)start.row * C + start.col] = 0.0f;
		Cell nb[8];
		while (!frontier.Empty()) {
			const Cell cell = frontier.Pop().position;
			inFrontier[(size_t)cell.row * C + cell.col] = 0;
			explored[(size_t)cell.row * C + cell.col] = 1;
			int nn = GetNeighbors(cell, R, C, nb);
			for (int q = 0; q < nn; q++) {
				const Cell& n = nb[q];
				if (world->IsOccupied(n))
					continue;
				if (n.IsDiagonalTo(cell))
					if (world->IsOccupied(Cell(n.row, cell.col)) && world->IsOccupied(Cell(cell.row, n.col)))
						continue;
				float transitionCost = cell.row == n.row || cell.col == n.col ? 1.0f : std::sqrt(2.0f);
				float pathCost = transitionCost + cost[(size_t)cell.row * C + cell.col];
				size_t ni = (size_t)n.row * C + n.col;
				bool inF = inFrontier[ni];
				bool inE = explored[ni];
				if (!inF && !inE) {
					frontier.Push({ n, pathCost });
					inFrontier[ni] = 1;
					cost[ni] = pathCost;
				}
				// else if (inFrontier): the reference compares the frontier value
				// with m_cost[n], which are always equal -> never relaxes (Q3).
			}
		}
	}

	/// Same semantics on the (cost, -seq) heap: O(N log N).
	void Update(const Pose2d& goalPose)
	{
		const int R = world->rows, C = world->columns;
		std::fill(cost.begin(), cost.end(), std::numeric_limits<float>::infinity());
		std::fill(explored.begin(), explored.end(), 0);
		Cell start = world->WorldPositionToGridCell(goalPose.x, goalPose.y);
		if (!start.IsValid())
			return;
		LifoHeap<float> heap;
		heap.Push(0.0f, (uint32_t)(start.row * C + start.col));
		cost[(size_t)start.row * C + start.col] = 0.0f;
		const float kDiag = std::sqrt(2.0f);
		Cell nb[8];
		const int* occ = world->occ->data.data();
		while (!heap.Empty()) {
			auto e = heap.Pop();
			const Cell cell((int)(e.id / C), (int)(e.id % C));
			explored[e.id] = 1;
			int nn = GetNeighbors(cell, R, C, nb);
			for (int q = 0; q < nn; q++) {
				const Cell& n = nb[q];
				size_t ni = (size_t)n.row * C + n.col;
				if (occ[ni] >= 0)
					continue;
				if (n.row != cell.row && n.col != cell.col)
					if (occ[(size_t)n.row * C + cell.col] >= 0 && occ[(size_t)cell.row * C + n.col] >= 0)
						continue;
				if (cost[ni] != std::numeric_limits<float>::infinity())
					continue; // already discovered (in frontier or explored)
				float transitionCost = cell.row == n.row || cell.col == n.col ? 1.0f : kDiag;
				float pathCost = transitionCost + cost[e.id];
				heap.Push(pathCost, (uint32_t)ni);
				cost[ni] = pathCost;
			}
		}
	}

	double GetHeuristicValue(const Pose2d& state) const
	{
		// heuristics.cpp:155-165
		double euclidean = Norm(goal.x - state.x, goal.y - state.y);
		Cell cell = world->WorldPositionToGridCell(state.x, state.y);
		if (!cell.IsValid())
			return euclidean;
		size_t ci = (size_t)cell.row * world->columns + cell.col;
		if (!explored[ci])
			return euclidean;
		double heuristic = cost[ci] * costMultiplier - diagonalResolution;
		return std::max(heuristic, euclidean);
	}
};

// ---------------------------------------------------------------------------
// utils/random.h -- one engine per search instead of the process-global one;
// the reference reseeded per query gives the same stream.
// ---------------------------------------------------------------------------
struct Rng {
	std::mt19937_64 engine;
	std::uniform_real_distribution<double> uniform { 0.0, std::nextafter(1.0, std::numeric_limits<double>::max()) }; // random.h:18
	uint64_t draws = 0;
	explicit Rng(uint64_t seed) :
		engine(seed) { }
	double SampleUniform(double lb, double ub)
	{
		// random.h:23-27
		draws++;
		double range = ub - lb;
		return lb + range * uniform(engine);
	}
};

// ---------------------------------------------------------------------------
// algo/hybrid_a_star.{h,cpp} + algo/a_star.h
// ---------------------------------------------------------------------------
inline int AliasHeadingBin(int theta)
{
	// Pose2<int>::WrapTheta instantiated from the generic template
	// (geometry/2dplane.h:36-45 with T = int; Appendix A Q6).
	int t = theta;
	while (t > M_PI)
		t -= 2 * M_PI;
	while (t < -M_PI)
		t += 2 * M_PI;
	return t;
}

struct DiscretePose {
	int x, y, theta;
	bool operator==(const DiscretePose& o) const { return x == o.x && y == o.y && theta == o.theta; }
};
struct DiscretePoseHash {
	size_t operator()(const DiscretePose& p) const
	{
		uint64_t h = (uint64_t)(uint32_t)p.x * 0x9E3779B97F4A7C15ull;
		h ^= ((uint64_t)(uint32_t)p.y + 0x9e3779b9 + (h << 6) + (h >> 2));
		h ^= ((uint64_t)(uint32_t)p.theta + 0x9e3779b9 + (h << 6) + (h >> 2));
		return (size_t)h;
	}
};

struct HybridNode {
	int parent = -1;
	Direction direction = Direction::NoMotion;
	DiscretePose discrete { 0, 0, 0 };
	Pose2d pose;
	double pathCost = 0, totalCost = 0;
	// action (the path from the parent): kind 0 = none (root), 1 = constant steer, 2 = Reeds-Shepp
	int kind = 0;
	double steering = 0, length = 0;
	Direction actionDirection = Direction::NoMotion;
	rs::PathSegment rsSegment;
	int rsWord = -1;
	bool dead = false; // removed from the frontier by ProcessPossibleShortcut
};

struct HybridResult {
	int status = -1; // 0 success, -1 failure (algo/path_planner.h:9-12)
	double cost = std::numeric_limits<double>::infinity();
	std::vector<DiscretePose> expanded; // cells in expansion order (incl. root)
	std::vector<int> expandedNode; // node index popped at each expansion
	std::vector<int> pathNodes; // node indices root..solution
	uint64_t nStateChecks = 0, nPathChecks = 0, nRngDraws = 0, nRsAttempts = 0, nChildren = 0;
	uint64_t nLatticeBoundary = 0; // poses discretised within 1e-9 cells of a lattice boundary (instrumentation, not the reference's)
};

struct HybridAStar {
	mutable uint64_t nLatticeBoundary = 0;
	const World* world;
	HybridParams param;
	Quirks quirks;
	KinematicBicycleModel model;
	std::vector<double> deltas;
	float voroFieldDiagResolution;
	std::unique_ptr<NonHolonomicHeuristic> nonHolo;
	std::unique_ptr<ObstaclesHeuristic> obstacle;
	std::vector<HybridNode> nodes;
	Pose2d goalPose;
	size_t maxExpansions = (size_t)-1;

	HybridAStar(const World* w, const HybridParams& p, const Quirks& q = Quirks()) :
		world(w), param(p), quirks(q)
	{
		// hybrid_a_star.cpp:13-29
		model.wheelbase = p.wheelbase;
		model.rearToCenter = 0.0;
		const double deltaMax = model.GetSteeringAngleFromTurningRadius(p.minTurningRadius);
		deltas.push_back(0.0);
		for (unsigned int i = 0; i < p.numGeneratedMotion / 2; i++) {
			double delta = (i + 1) / 2.0 * deltaMax;
			deltas.push_back(delta);
			deltas.push_back(-delta);
		}
		voroFieldDiagResolution = w->resolution * std::sqrt(2.0); // hybrid_a_star.cpp:38
	}

	/// hybrid_a_star.cpp:206-235 (heuristic part).  `table` lets callers reuse a built table.
	void Initialize(std::unique_ptr<NonHolonomicHeuristic> table = nullptr)
	{
		nonHolo = table ? std::move(table) : NonHolonomicHeuristic::Build(world->lb, world->ub, param);
		nonHolo->negativeKRead = quirks.negativeKRead;
		obstacle = std::make_unique<ObstaclesHeuristic>(world, param.reverseCostMultiplier, param.forwardCostMultiplier);
	}

	DiscretePose DiscretizePose(const Pose2d& pose) const
	{
		// hybrid_a_star.h:104-111, then the Pose2i constructor (2dplane.h:21-22)
		const double qx = pose.x / param.spatialResolution, qy = pose.y / param.spatialResolution, qt = pose.WrapTheta() / param.angularResolution;
		if (std::fabs(qx - std::rint(qx)) < 1e-9 || std::fabs(qy - std::rint(qy)) < 1e-9 || std::fabs(qt - std::rint(qt)) < 1e-9)
			nLatticeBoundary++;
		int t = static_cast<int>(pose.WrapTheta() / param.angularResolution);
		if (quirks.headingAlias)
			t = AliasHeadingBin(t);
		return { static_cast<int>(pose.x / param.spatialResolution), static_cast<int>(pose.y / param.spatialResolution), t };
	}

	double Heuristic(const Pose2d& pose) const
	{
		// a_star.h:102-109 (max over the two heuristics, seeded with -inf)
		double value = -std::numeric_limits<double>::infinity();
		value = std::max(value, nonHolo->GetHeuristicValue(pose));
		value = std::max(value, obstacle->GetHeuristicValue(pose));
		return value;
	}

	template <typename PathT>
	double GetVoronoiCost(const PathT& path) const
	{
		// hybrid_a_star.cpp:93-109 -- keeps only the LAST sample (Q8)
		float voronoiCost = 0.0;
		float interpLength = voroFieldDiagResolution;
		const double pathLength = path.length;
		for (double length = 0.0; length < pathLength; length += interpLength) {
			Pose2d p = path.Interpolate(length / pathLength);
			Cell cell = world->WorldPositionToGridCell(p.x, p.y, false);
			// The reference does not bounds-check here (PP_ASSERT is compiled
			// out); clamp so the oracle never reads out of the grid.
			int r = std::min(std::max(cell.row, 0), world->rows - 1);
			int c = std::min(std::max(cell.col, 0), world->columns - 1);
			voronoiCost = world->PathCostAt(r, c);
		}
		voronoiCost *= interpLength;
		return param.voronoiCostMultiplier * voronoiCost;
	}

	static bool IdenticalPoses(const Pose2d& a, const Pose2d& b, double tol = 1e-3)
	{
		// hybrid_a_star.h:208-211
		return Norm(a.x - b.x, a.y - b.y) < tol && std::abs(a.theta - b.theta) < tol * M_PI / 180.0;
	}

	struct Child {
		HybridNode node;
		double cost;
	};

	bool GetConstantSteerChild(const HybridNode& state, double delta, Direction direction, Child& out) const
	{
		// hybrid_a_star.cpp:111-147
		PathConstantSteer path(&model, state.pose, delta, param.spatialResolution * 1.5, direction);
		HybridNode& child = out.node;
		child = HybridNode();
		child.direction = path.GetDirection(1.0);
		child.pose = path.final;
		child.discrete = DiscretizePose(child.pose);
		float lastValidRatio;
		if (!world->IsPathValid(path, &lastValidRatio)) {
			path.Truncate(lastValidRatio);
			child.pose = path.final;
			child.discrete = DiscretizePose(child.pose);
			if (state.discrete == child.discrete)
				return false;
		}
		double pathCost;
		switch (direction) {
		case Direction::Forward: pathCost = param.forwardCostMultiplier * path.length; break;
		case Direction::Backward: pathCost = param.reverseCostMultiplier * path.length; break;
		default: pathCost = 0.0;
		}
		double switchingCost = 0.0; // hybrid_a_star.cpp:142 compares GetDirection(1.0) with itself (Q8)
		double voronoiCost = GetVoronoiCost(path);
		out.cost = pathCost + switchingCost + voronoiCost;
		child.kind = 1;
		child.steering = delta;
		child.length = path.length;
		child.actionDirection = direction;
		return true;
	}

	bool GetReedsSheppChild(const HybridNode& state, Child& out) const
	{
		// hybrid_a_star.cpp:149-173
		int word = -1;
		auto seg = rs::GetOptimalPath(state.pose, goalPose, param.minTurningRadius, param.reverseCostMultiplier, param.forwardCostMultiplier, param.directionSwitchingCost, &word);
		PathReedsShepp path(state.pose, seg, param.minTurningRadius);
		if (!world->IsPathValid(path))
			return false;
		double pathAndSwitchingCosts = path.ComputeCost(param.directionSwitchingCost, param.reverseCostMultiplier, param.forwardCostMultiplier);
		HybridNode& child = out.node;
		child = HybridNode();
		child.direction = path.GetDirection(1.0);
		child.pose = path.final;
		child.discrete = DiscretizePose(child.pose);
		double voronoiCost = GetVoronoiCost(path);
		out.cost = pathAndSwitchingCosts + voronoiCost;
		child.kind = 2;
		child.length = path.length;
		child.rsSegment = seg;
		child.rsWord = word;
		return true;
	}

	/// hybrid_a_star.cpp:237-257 (up to the graph search) + a_star.h:326-427.
	/// `skipObstacleUpdate` reuses the current obstacle field (same goal cell).
	HybridResult Search(const Pose2d& start, const Pose2d& goal, uint64_t seed, bool skipObstacleUpdate = false)
	{
		HybridResult res;
		Rng rng(seed);
		const uint64_t sc0 = world->nStateChecks, pc0 = world->nPathChecks;
		nLatticeBoundary = 0;
		goalPose = goal;
		if (!skipObstacleUpdate)
			obstacle->Update(goal); // hybrid_a_star.cpp:249
		// InitializeSearch, a_star.h:350-364
		nonHolo->goal = goal;
		obstacle->goal = goal;
		nodes.clear();
		LifoHeap<double> frontier;
		// cell -> state: absent = unseen; value >= 0 = node index in the frontier; -2 = explored
		std::unordered_map<DiscretePose, int, DiscretePoseHash> cellState;
		std::unordered_map<DiscretePose, uint8_t, DiscretePoseHash> exploredSet;
		HybridNode root;
		root.pose = start;
		root.discrete = DiscretizePose(start);
		nodes.push_back(root);
		frontier.Push(0.0, 0);
		cellState[root.discrete] = 0;
		exploredSet[root.discrete] = 1; // the root is inserted in explored at init (a_star.h:361)

		std::vector<Child> children;
		while (!frontier.Empty()) {
			auto top = frontier.Pop();
			int ni = (int)top.id;
			if (nodes[ni].dead)
				continue; // lazily removed entry
			{
				auto it = cellState.find(nodes[ni].discrete);
				if (it != cellState.end() && it->second == ni)
					cellState.erase(it); // Frontier::Pop erases from the membership set (frontier.h:89)
			}
			if (IdenticalPoses(nodes[ni].pose, goal)) { // hybrid_a_star.h:193-196
				res.status = 0;
				res.cost = nodes[ni].pathCost;
				for (int k = ni; k >= 0; k = nodes[k].parent)
					res.pathNodes.push_back(k);
				std::reverse(res.pathNodes.begin(), res.pathNodes.end());
				break;
			}
			if (res.expanded.size() >= maxExpansions)
				break;
			// Expand, a_star.h:377-409
			exploredSet[nodes[ni].discrete] = 1;
			res.expanded.push_back(nodes[ni].discrete);
			res.expandedNode.push_back(ni);
			const HybridNode parent = nodes[ni];
			children.clear();
			// GetNeighborStates, hybrid_a_star.cpp:59-91
			for (double delta : deltas) {
				Child c;
				if (GetConstantSteerChild(parent, delta, Direction::Forward, c))
					children.push_back(c);
				if (GetConstantSteerChild(parent, delta, Direction::Backward, c))
					children.push_back(c);
			}
			double hCost = Heuristic(parent.pose);
			if (hCost < 10.0 || rng.SampleUniform(0.0, 1.0) < 10.0 / (hCost * hCost)) {
				Child c;
				res.nRsAttempts++;
				if (GetReedsSheppChild(parent, c))
					children.push_back(c);
			}
			for (auto& ch : children) {
				res.nChildren++;
				HybridNode child = ch.node;
				child.parent = ni;
				child.pathCost = parent.pathCost + ch.cost;
				child.totalCost = child.pathCost + Heuristic(child.pose);
				auto fit = cellState.find(child.discrete);
				bool inFrontier = fit != cellState.end();
				bool inExplored = exploredSet.find(child.discrete) != exploredSet.end();
				if (!inFrontier && !inExplored) {
					int ci = (int)nodes.size();
					nodes.push_back(child);
					frontier.Push(child.totalCost, (uint32_t)ci);
					cellState[child.discrete] = ci;
				} else if (inFrontier) {
					// GraphSearch::ProcessPossibleShortcut, hybrid_a_star.h:199-205 + a_star.h:417-427
					HybridNode& fn = nodes[fit->second];
					if (IdenticalPoses(fn.pose, child.pose) && fn.totalCost > child.totalCost) {
						fn.dead = true;
						int ci = (int)nodes.size();
						nodes.push_back(child);
						frontier.Push(child.totalCost, (uint32_t)ci);
						cellState[child.discrete] = ci;
					}
				}
			}
		}
		res.nStateChecks = world->nStateChecks - sc0;
		res.nPathChecks = world->nPathChecks - pc0;
		res.nRngDraws = rng.draws;
		res.nLatticeBoundary = nLatticeBoundary;
		return res;
	}
};

// ---------------------------------------------------------------------------
// algo/a_star.h + a_star_n2.cpp -- 8-connected grid A* (unidirectional) and
// algo/bidirectional_a_star.h
// ---------------------------------------------------------------------------
using CellFn = std::function<double(const Cell&, const Cell&)>;

struct GridAStarResult {
	int status = -1;
	double cost = std::numeric_limits<double>::infinity();
	std::vector<Cell> path;
	std::vector<Cell> explored; // expansion order
};

struct GridSearchState { // one AStar<GridCellPosition> instance (a_star.h:213-441)
	struct N {
		Cell cell;
		int parent;
		double pathCost, totalCost;
		bool dead;
	};
	const World* world;
	CellFn costFn, heurFn; // heurFn(state, goal)
	Cell init, goal;
	std::vector<N> nodes;
	LifoHeap<double> frontier;
	std::vector<int> inFrontier; // node idx or -1
	std::vector<int> exploredNode; // node idx or -1 (ExploredMap semantics); root marked at init
	std::vector<Cell> exploredOrder;
	int solution = -1;

	size_t Idx(const Cell& c) const { return (size_t)c.row * world->columns + c.col; }

	void InitializeSearch()
	{
		// a_star.h:350-364
		nodes.clear();
		frontier.Clear();
		inFrontier.assign((size_t)world->rows * world->columns, -1);
		exploredNode.assign((size_t)world->rows * world->columns, -1);
		exploredOrder.clear();
		solution = -1;
		nodes.push_back({ init, -1, 0.0, 0.0, false });
		frontier.Push(0.0, 0);
		inFrontier[Idx(init)] = 0;
		exploredNode[Idx(init)] = 0;
	}
	bool FrontierEmpty()
	{
		while (!frontier.Empty() && nodes[frontier.Top().id].dead)
			frontier.Pop();
		return frontier.Empty();
	}
	int PopFrontier()
	{
		FrontierEmpty();
		int ni = (int)frontier.Pop().id;
		if (inFrontier[Idx(nodes[ni].cell)] == ni)
			inFrontier[Idx(nodes[ni].cell)] = -1;
		return ni;
	}
	double TopPathCost()
	{
		FrontierEmpty();
		return nodes[frontier.Top().id].pathCost;
	}
	void Expand(int ni)
	{
		// a_star.h:377-409 with AStarStatePropagatorFcnN2::GetNeighborStates (a_star_n2.cpp:12-28)
		exploredNode[Idx(nodes[ni].cell)] = ni;
		exploredOrder.push_back(nodes[ni].cell);
		const Cell cell = nodes[ni].cell;
		Cell nb[8];
		int nn = GetNeighbors(cell, world->rows, world->columns, nb);
		for (int q = 0; q < nn; q++) {
			const Cell& n = nb[q];
			if (world->IsOccupied(n))
				continue;
			if (n.IsDiagonalTo(cell))
				if (world->IsOccupied(Cell(n.row, cell.col)) && world->IsOccupied(Cell(cell.row, n.col)))
					continue;
			double transition = costFn(cell, n);
			double pathCost = nodes[ni].pathCost + transition;
			double totalCost = pathCost + heurFn(n, goal);
			size_t idx = Idx(n);
			bool inF = inFrontier[idx] >= 0;
			bool inE = exploredNode[idx] >= 0;
			if (!inF && !inE) {
				int ci = (int)nodes.size();
				nodes.push_back({ n, ni, pathCost, totalCost, false });
				frontier.Push(totalCost, (uint32_t)ci);
				inFrontier[idx] = ci;
			} else if (inF) {
				// a_star.h:417-427
				N& fn = nodes[inFrontier[idx]];
				if (fn.totalCost > totalCost) {
					fn.dead = true;
					int ci = (int)nodes.size();
					nodes.push_back({ n, ni, pathCost, totalCost, false });
					frontier.Push(totalCost, (uint32_t)ci);
					inFrontier[idx] = ci;
				}
			}
		}
	}
	std::vector<Cell> PathTo(int ni) const
	{
		std::vector<Cell> p;
		for (int k = ni; k >= 0; k = nodes[k].parent)
			p.push_back(nodes[k].cell);
		std::reverse(p.begin(), p.end());
		return p;
	}
};

inline GridAStarResult GridAStar(const World* w, const Cell& init, const Cell& goal, CellFn costFn, CellFn heurFn)
{
	// a_star.h:326-346
	GridAStarResult res;
	GridSearchState s { w, costFn, heurFn, init, goal };
	s.InitializeSearch();
	while (!s.FrontierEmpty()) {
		int ni = s.PopFrontier();
		if (s.nodes[ni].cell == goal) {
			res.status = 0;
			res.cost = s.nodes[ni].pathCost;
			res.path = s.PathTo(ni);
			break;
		}
		s.Expand(ni);
	}
	res.explored = s.exploredOrder;
	return res;
}

struct BidirResult {
	int status = -1;
	double cost = std::numeric_limits<double>::infinity();
	std::vector<Cell> path;
	std::vector<Cell> fExplored, rExplored;
};

/// bidirectional_a_star.h:130-196 with the AverageHeuristic pair (:10-39).
/// AverageHeuristic::SetGoal only stores its own m_goal (it derives from
/// AStarConcreteHeuristic, a_star.h:52-63) and never forwards the goal to the two
/// heuristics it wraps, so those keep whatever goal they were last given:
/// `innerGoalF` / `innerGoalR` are the goals held by the forward / reverse inner
/// heuristic objects (in interfaces/python/scripts/example_a_star_grid.py:103 both are
/// the same object, still holding the unidirectional run's goal).
inline BidirResult BidirectionalGridAStar(const World* w, const Cell& init, const Cell& goal, CellFn costFn, CellFn heurFn,
	const Cell& innerGoalF, const Cell& innerGoalR)
{
	BidirResult res;
	auto Hf = [&](const Cell& s) { return heurFn(s, innerGoalF); };
	auto Hr = [&](const Cell& s) { return heurFn(s, innerGoalR); };
	// AverageHeuristic::Update (bidirectional_a_star.h:29-34): constant = toInit(goal of that direction) / 2
	const double fConst = Hr(goal) / 2.0;
	const double rConst = Hf(init) / 2.0;
	CellFn fH = [&](const Cell& s, const Cell&) { return fConst + (Hf(s) - Hr(s)) / 2.0; };
	CellFn rH = [&](const Cell& s, const Cell&) { return rConst + (Hr(s) - Hf(s)) / 2.0; };
	GridSearchState f { w, costFn, fH, init, goal };
	GridSearchState r { w, costFn, rH, goal, init };
	f.InitializeSearch();
	r.InitializeSearch();
	// ExploredMap of the reference also holds the root -> exploredNode[...] = 0 at init (done above).
	const double costOffset = fH(goal, goal) + rH(goal, init);
	double bestCost = std::numeric_limits<double>::infinity();
	auto findIntersection = [&](int nodeA, GridSearchState& A, GridSearchState& B) {
		// bidirectional_a_star.h:180-196
		int nb = B.exploredNode[B.Idx(A.nodes[nodeA].cell)];
		if (nb >= 0) {
			double cand = A.nodes[nodeA].pathCost + B.nodes[nb].pathCost;
			if (cand < bestCost) {
				bestCost = cand;
				A.solution = nodeA;
				B.solution = nb;
			}
		}
	};
	while (!f.FrontierEmpty() && !r.FrontierEmpty()) {
		int fn = f.PopFrontier();
		f.Expand(fn);
		findIntersection(fn, f, r);
		int rn = r.PopFrontier();
		r.Expand(rn);
		findIntersection(rn, r, f);
		if (f.solution >= 0 && r.solution >= 0) {
			bool done = false;
			if (f.FrontierEmpty() || r.FrontierEmpty())
				done = true;
			else if (f.TopPathCost() + r.TopPathCost() >= bestCost + costOffset)
				done = true;
			if (done) {
				res.status = 0;
				break;
			}
		}
	}
	if (res.status == 0) {
		res.cost = f.nodes[f.solution].pathCost + r.nodes[r.solution].pathCost;
		auto fp = f.PathTo(f.solution);
		auto rp = r.PathTo(r.solution);
		res.path = fp;
		res.path.insert(res.path.end(), rp.rbegin(), rp.rend()); // duplicates the meeting cell (Q16)
	}
	res.fExplored = f.exploredOrder;
	res.rExplored = r.exploredOrder;
	return res;
}

// ---------------------------------------------------------------------------
// algo/rrt.h, algo/rrt_star.h, utils/tree.h (flann replaced by exact brute-force
// kNN: squared L2, ascending, ties by lower insertion index -- tie order is
// "parity unpinned", flann is absent from /root/reference).
// ---------------------------------------------------------------------------
struct RRTParams { // rrt.h:12-21 / rrt_star.h:12-21
	unsigned int maxIteration = 100;
	unsigned int maxNumberTreeNode = 10000;
	double maxConnectionDistance = 0.1;
	double goalBias = 0.05;
};

struct RRTResult {
	int status = -1;
	std::vector<Point2d> path;
	std::vector<Point2d> nodes;
	std::vector<int> parents;
	std::vector<double> costs;
	uint64_t iterations = 0, nKnnQueries = 0, nEdgeChecks = 0;
};

/// Validator for R2: `free` = StateValidatorFree (state_validator_free.h:9-31);
/// otherwise the occupancy test of state_validator_occupancy_map.cpp applied to
/// (x, y, 0) -- the R2 occupancy validator SURVEY 8(d) config 3 defines.
struct R2Problem {
	Point2d lb, ub;
	const World* world = nullptr; // nullptr => free space
	mutable uint64_t edgeChecks = 0;

	struct SegPath {
		Pose2d init;
		PathR2 seg;
		double length;
		Pose2d Interpolate(double ratio) const
		{
			Point2d p = seg.Interpolate(ratio);
			return Pose2d::Raw(p.x, p.y, 0.0);
		}
	};
	bool IsPathValid(const PathR2& path) const
	{
		edgeChecks++;
		if (!world)
			return true; // state_validator_free.h:24-29
		SegPath sp { Pose2d::Raw(path.init.x, path.init.y, 0.0), path, path.length };
		return world->IsPathValid(sp);
	}
};

struct PointTree { // utils/tree.h:34-177 without flann
	std::vector<Point2d> pts;
	std::vector<int> parent;
	std::vector<double> cost;
	struct PHash {
		size_t operator()(const std::pair<double, double>& p) const
		{
			size_t seed = 0;
			std::hash<double> h;
			seed ^= h(p.first) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
			seed ^= h(p.second) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
			return seed;
		}
	};
	std::unordered_map<std::pair<double, double>, int, PHash> index; // m_exploredNodeMap
	uint64_t knnQueries = 0;

	void Clear()
	{
		pts.clear();
		parent.clear();
		cost.clear();
		index.clear();
	}
	size_t Size() const { return index.size(); }
	int CreateRoot(const Point2d& s)
	{
		pts.push_back(s);
		parent.push_back(-1);
		cost.push_back(0.0);
		index[{ s.x, s.y }] = 0;
		return 0;
	}
	/// tree.h:73-95 -- k nearest, ascending squared distance.
	std::vector<int> Nearest(const Point2d& q, unsigned int k)
	{
		knnQueries++;
		std::vector<int> out;
		if (k == 0)
			return out;
		std::vector<std::pair<double, int>> d;
		d.reserve(pts.size());
		for (size_t i = 0; i < pts.size(); i++) {
			double dx = pts[i].x - q.x, dy = pts[i].y - q.y;
			d.push_back({ dx * dx + dy * dy, (int)i });
		}
		size_t kk = std::min<size_t>(k, d.size());
		std::partial_sort(d.begin(), d.begin() + kk, d.end());
		for (size_t i = 0; i < kk; i++)
			out.push_back(d[i].second);
		return out;
	}
	/// tree.h:124-146; returns the existing node if the state is present (Q15).
	int Extend(const Point2d& target, int source)
	{
		auto it = index.find({ target.x, target.y });
		if (it != index.end())
			return it->second;
		int id = (int)pts.size();
		pts.push_back(target);
		parent.push_back(source);
		cost.push_back(0.0);
		index[{ target.x, target.y }] = id;
		return id;
	}
};

inline Point2d SampleUniformR2(Rng& rng, const Point2d& lb, const Point2d& ub)
{
	// state_space_r2.cpp:25-35
	Point2d s;
	s.x = rng.SampleUniform(lb.x, ub.x);
	s.y = rng.SampleUniform(lb.y, ub.y);
	return s;
}

inline PathR2 SteerTowards(const Point2d& from, const Point2d& to, double distance)
{
	// rrt_star.h:143-151
	PathR2 path(from, to);
	if (path.length > 0) {
		double ratio = std::clamp(distance / path.length, 0.0, 1.0);
		path.Truncate(ratio);
	}
	return path;
}

inline void FillTree(RRTResult& res, const PointTree& tree, int solution)
{
	res.nodes = tree.pts;
	res.parents = tree.parent;
	res.costs = tree.cost;
	if (solution >= 0) {
		for (int k = solution; k >= 0; k = tree.parent[k])
			res.path.push_back(tree.pts[k]);
		std::reverse(res.path.begin(), res.path.end());
	}
}

/// rrt.h:55-95
inline RRTResult RRT(const R2Problem& prob, const RRTParams& p, const Point2d& init, const Point2d& goal, uint64_t seed)
{
	RRTResult res;
	Rng rng(seed);
	PointTree tree;
	tree.CreateRoot(init);
	int count = -1;
	int solution = -1;
	while (true) {
		count++;
		if (count > (int)p.maxIteration)
			break;
		if (tree.Size() > p.maxNumberTreeNode)
			break;
		res.iterations++;
		Point2d randomState = rng.SampleUniform(0, 1) < p.goalBias ? goal : SampleUniformR2(rng, prob.lb, prob.ub);
		int nearest = tree.Nearest(randomState, 1)[0];
		PathR2 pathNearToNew = SteerTowards(tree.pts[nearest], randomState, p.maxConnectionDistance);
		if (!prob.IsPathValid(pathNearToNew))
			continue;
		Point2d newState = pathNearToNew.final;
		int newNode = tree.Extend(newState, nearest);
		if (Norm(newState.x - goal.x, newState.y - goal.y) < 1) { // rrt.h:125-128
			solution = newNode;
			res.status = 0;
			break;
		}
	}
	res.nKnnQueries = tree.knnQueries;
	res.nEdgeChecks = prob.edgeChecks;
	FillTree(res, tree, solution);
	return res;
}

/// rrt_star.h:53-112 -- choose-parent only, no rewire (Q15): mode 1.
/// Modes 2 / 3 are NOT in the reference (its FIXME at rrt_star.h:83): the definition the product's extension is tested against --
/// after the new node is linked, every near node (in near-set order) that gets cheaper through it and whose connecting edge
/// is valid is re-parented to it (Node::Reparent, utils/node.h:203-225) and its subtree's costs follow (cost = parent cost +
/// stored edge length); mode 3 takes as near-set the <= 16 nearest nodes within gamma * sqrt(ln(n + 1) / (n + 1)).
inline RRTResult RRTStar(const R2Problem& prob, const RRTParams& p, const Point2d& init, const Point2d& goal, uint64_t seed, int mode = 1, double gamma = 0.0)
{
	std::vector<std::vector<int>> children; // modes 2 / 3
	std::vector<double> edgeLen;
	if (mode >= 2) {
		children.emplace_back();
		edgeLen.push_back(0.0);
	}
	RRTResult res;
	Rng rng(seed);
	PointTree tree;
	tree.CreateRoot(init);
	int count = -1;
	int solution = -1;
	while (true) {
		count++;
		if (count > (int)p.maxIteration)
			break;
		if (tree.Size() > p.maxNumberTreeNode)
			break;
		res.iterations++;
		Point2d randomState = rng.SampleUniform(0, 1) < p.goalBias ? goal : SampleUniformR2(rng, prob.lb, prob.ub);
		int nearest = tree.Nearest(randomState, 1)[0];
		PathR2 pathNearToNew = SteerTowards(tree.pts[nearest], randomState, p.maxConnectionDistance);
		if (!prob.IsPathValid(pathNearToNew))
			continue;
		Point2d newState = pathNearToNew.final;
		unsigned int nn = std::max<unsigned int>(1, std::log(tree.Size())); // rrt_star.h:84
		if (mode == 3)
			nn = (unsigned int)std::min<size_t>(16, tree.Size());
		std::vector<int> nearNodes = tree.Nearest(newState, nn);
		if (mode == 3) {
			const double np1 = (double)tree.Size() + 1.0;
			const double radius = gamma * std::sqrt(std::log(np1) / np1);
			size_t m = 0;
			while (m < nearNodes.size()) {
				double dx = tree.pts[nearNodes[m]].x - newState.x, dy = tree.pts[nearNodes[m]].y - newState.y;
				if (!(dx * dx + dy * dy <= radius * radius))
					break;
				m++;
			}
			nearNodes.resize(std::max<size_t>(1, m));
		}
		int bestParent = -1;
		double bestCost = std::numeric_limits<double>::infinity();
		for (int node : nearNodes) {
			PathR2 pathParentToNew(tree.pts[node], newState); // SteerExactly, rrt_star.h:156-160
			double cost = tree.cost[node] + pathParentToNew.length;
			if (cost < bestCost && prob.IsPathValid(pathParentToNew)) {
				bestParent = node;
				bestCost = cost;
			}
		}
		// rrt_star.h:100: Extend(newState, bestParentNode); a null parent falls
		// back to the nearest node inside Tree::Extend (tree.h:131).
		// tree.h:124-133: an existing state returns its node before any nearest-node query
		int src = bestParent;
		if (src < 0 && tree.index.find({ newState.x, newState.y }) == tree.index.end())
			src = tree.Nearest(newState, 1)[0];
		const size_t sizeBefore = tree.pts.size();
		int newNode = tree.Extend(newState, src);
		tree.cost[newNode] = bestCost; // rrt_star.h:101 (also overwrites an existing node's cost)
		if (mode >= 2 && tree.pts.size() > sizeBefore) {
			const int par = tree.parent[newNode];
			children.emplace_back();
			edgeLen.push_back(PathR2(tree.pts[par], newState).length);
			children[par].push_back(newNode);
			if (bestParent >= 0) {
				for (int node : nearNodes) {
					if (node == newNode || node == par)
						continue;
					PathR2 pathNewToNear(newState, tree.pts[node]);
					double through = tree.cost[newNode] + pathNewToNear.length;
					if (through < tree.cost[node] && prob.IsPathValid(pathNewToNear)) {
						auto& sib = children[tree.parent[node]];
						sib.erase(std::find(sib.begin(), sib.end(), node));
						tree.parent[node] = newNode;
						children[newNode].push_back(node);
						edgeLen[node] = pathNewToNear.length;
						tree.cost[node] = through;
						std::vector<int> stack { node };
						while (!stack.empty()) {
							int q = stack.back();
							stack.pop_back();
							for (int c : children[q]) {
								tree.cost[c] = tree.cost[q] + edgeLen[c];
								stack.push_back(c);
							}
						}
					}
				}
			}
		}
		if (newState.x == goal.x && newState.y == goal.y) { // rrt_star.h:136-139, exact equality
			solution = newNode;
			res.status = 0;
			break;
		}
	}
	res.nKnnQueries = tree.knnQueries;
	res.nEdgeChecks = prob.edgeChecks;
	FillTree(res, tree, solution);
	return res;
}

} // namespace ppo
