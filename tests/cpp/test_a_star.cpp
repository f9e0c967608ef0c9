// Mirror of the reference's planner/tests/test_a_star.cpp:15-34 on the host engine (pathplanning_amd/host/a_star.hpp):
// 8-connected infinite lattice (planner/tests/state_space/a_star_state_space_2d.h:7-23), (0,0) -> (10,5).  Asserts stay
// live.  Pure CPU: links libpphip.so only because planner_hip.hpp declares the GPU-backed classes next to the value types.
// Adds what the reference test leaves unchecked: optimal cost, tree search mode, the LIFO tie rule and the bidirectional
// stop rule.
#undef NDEBUG
#include <cassert>
#include <cmath>
#include <cstdio>

#include "../../pathplanning_amd/host/a_star.hpp"

using namespace Planner;

struct Point2i {
	int x = 0, y = 0;
	bool operator==(const Point2i& o) const { return x == o.x && y == o.y; }
};
namespace std {
template <>
struct hash<Point2i> {
	size_t operator()(const Point2i& p) const { return std::hash<long long>()(((long long)p.x << 32) ^ (unsigned)p.y); }
};
}

class Lattice : public AStarStatePropagator<Point2i, NullAction> {
public:
	std::vector<std::tuple<Point2i, NullAction, double>> GetNeighborStates(const Point2i& s) override
	{
		static const int d[8][2] = { { 1, 1 }, { 1, 0 }, { 1, -1 }, { 0, -1 }, { -1, -1 }, { -1, 0 }, { -1, 1 }, { 0, 1 } };
		std::vector<std::tuple<Point2i, NullAction, double>> out;
		for (auto& k : d)
			out.push_back({ Point2i { s.x + k[0], s.y + k[1] }, NullAction(), (k[0] && k[1]) ? std::sqrt(2.0) : 1.0 });
		expansions++;
		return out;
	}
	int expansions = 0;
};

class Euclid : public AStarConcreteHeuristic<Point2i> {
public:
	double GetHeuristicValue(const Point2i& s) override { return sqrtf(powf(s.x - m_goal.x, 2) + powf(s.y - m_goal.y, 2)); }
};

int main()
{
	auto lattice = makeRef<Lattice>();
	auto h = makeRef<Euclid>();
	AStar<Point2i, NullAction> aStar;
	aStar.SetInitState({ 0, 0 });
	aStar.SetGoalState({ 10, 5 });
	assert(aStar.SearchPath() == Status::Failure); // not initialised
	assert(aStar.Initialize(lattice, h));
	assert(aStar.SearchPath() == Status::Success);
	auto path = aStar.GetPath();
	assert(!path.empty());
	assert(path.front() == (Point2i { 0, 0 }) && path.back() == (Point2i { 10, 5 }));
	assert(std::fabs(aStar.GetOptimalCost() - (5 * std::sqrt(2.0) + 5)) < 1e-12);
	assert(aStar.GetActions().size() == path.size() - 1);
	const int graphExpansions = lattice->expansions;

	AStar<Point2i, NullAction, std::hash<Point2i>, std::equal_to<Point2i>, false> tree; // tree search: no explored set
	tree.Initialize(lattice, h);
	tree.SetInitState({ 0, 0 });
	tree.SetGoalState({ 10, 5 });
	assert(tree.SearchPath() == Status::Success);
	assert(std::fabs(tree.GetOptimalCost() - aStar.GetOptimalCost()) < 1e-12);

	// bidirectional with the averaged pair: same optimal cost, meeting state repeated once in the path
	auto hf = makeRef<Euclid>(), hr = makeRef<Euclid>();
	hf->SetGoal({ 10, 5 });
	hr->SetGoal({ 0, 0 });
	auto [af, ar] = BidirectionalAStar<Point2i, NullAction>::GetAverageHeuristicPair(hf, hr);
	BidirectionalAStar<Point2i, NullAction> bi;
	assert(bi.Initialize(lattice, lattice, af, ar));
	bi.SetInitState({ 0, 0 });
	bi.SetGoalState({ 10, 5 });
	assert(bi.SearchPath() == Status::Success);
	assert(std::fabs(bi.GetOptimalCost() - aStar.GetOptimalCost()) < 1e-9);
	auto bp = bi.GetPath();
	int dup = 0;
	for (size_t i = 1; i < bp.size(); i++)
		dup += bp[i] == bp[i - 1];
	assert(dup == 1 && bp.front() == (Point2i { 0, 0 }) && bp.back() == (Point2i { 10, 5 }));

	// zero heuristic on a unit-cost line: every open node of a wave ties, the last pushed is expanded first (Appendix A Q1)
	class Zero : public AStarConcreteHeuristic<Point2i> {
		double GetHeuristicValue(const Point2i&) override { return 0.0; }
	};
	class Fan : public AStarStatePropagator<Point2i, NullAction> {
		std::vector<std::tuple<Point2i, NullAction, double>> GetNeighborStates(const Point2i& s) override
		{
			if (s.x == 0)
				return { { Point2i { 1, 0 }, NullAction(), 1.0 }, { Point2i { 1, 1 }, NullAction(), 1.0 }, { Point2i { 1, 2 }, NullAction(), 1.0 } };
			return {};
		}
	};
	AStar<Point2i, NullAction> fan;
	fan.Initialize(makeRef<Fan>(), makeRef<Zero>());
	fan.SetInitState({ 0, 0 });
	fan.SetGoalState({ 9, 9 });
	assert(fan.SearchPath() == Status::Failure);
	auto order = fan.GetExpansionOrder();
	assert(order.size() == 4 && order[1] == (Point2i { 1, 2 }) && order[2] == (Point2i { 1, 1 }) && order[3] == (Point2i { 1, 0 }));
	std::printf("a*: %zu path states, cost %.6f, %d expansions; bidirectional %zu states\n", path.size(), aStar.GetOptimalCost(), graphExpansions, bp.size());
	return 0;
}
