// Mirrors the reference's own integration tests on the GPU-backed plugin classes:
//   planner/tests/test_hybrid_a_star.cpp:9-36   empty 200x200 map, (0,0,0) -> (8,8,0.78)
//   planner/tests/test_rrt.cpp:9-38, test_rrt_star.cpp:10-39   R2 free space, (0,0) -> (2,2)
// Asserts stay live (the reference's CI compiles them out).  Needs a GPU.
#undef NDEBUG
#include <cassert>
#include <cstdio>

#include "../../pathplanning_amd/host/planner_hip.hpp"

using namespace Planner;

static void TestHybridAStar()
{
	std::array<Pose2d, 2> bounds = { Pose2d(-10, -10, -M_PI), Pose2d(10, 10, M_PI) };
	Ref<StateSpaceSE2> stateSpace = makeRef<StateSpaceSE2>(bounds);
	Ref<OccupancyMap> map = makeRef<OccupancyMap>(0.1f);
	Ref<StateValidatorOccupancyMap> stateValidator = makeRef<StateValidatorOccupancyMap>(stateSpace, map);
	assert(map->Rows() == 200 && map->Columns() == 200);

	HybridAStar hybridAStar;
	assert(hybridAStar.SearchPath() == Status::Failure); // not initialised
	assert(hybridAStar.Initialize(stateValidator));

	Pose2d start = { 0.0, 0.0, 0.0 };
	Pose2d goal = { 8.0, 8.0, 0.78 };
	hybridAStar.SetInitState(start);
	hybridAStar.SetGoalState(goal);
	assert(hybridAStar.SearchPath() == Status::Success);
	std::vector<Pose2d> path = hybridAStar.GetPath();
	assert(!path.empty());
	double spatialTolerance = 1e-1;
	double angularTolerance = 5 * M_PI / 180;
	assert((path.front().position - start.position).norm() < spatialTolerance);
	assert(std::fabs(path.front().theta - start.theta) < angularTolerance);
	assert((path.back().position - goal.position).norm() < spatialTolerance);
	assert(std::fabs(path.back().theta - goal.theta) < angularTolerance);
	assert(stateValidator->IsStateValid(start));
	assert(!stateValidator->IsStateValid(Pose2d(10.5, 0.0, 0.0)));
	std::printf("hybrid a*: %zu path nodes, cost %.6f\n", path.size(), hybridAStar.GetGraphSearchOptimalCost());
}

static void TestRRT()
{
	RRTStarR2 rrtStar(Point2d(0, 0), Point2d(5, 5));
	RRTStarParameters parameters;
	rrtStar.SetParameters(parameters);
	Point2d start = { 0.0, 0.0 }, goal = { 2.0, 2.0 };
	rrtStar.SetInitState(start);
	rrtStar.SetGoalState(goal);
	int ok = 0;
	for (uint64_t seed = 0; seed < 4; seed++) {
		rrtStar.SetSeed(seed);
		if (rrtStar.SearchPath() == Status::Success) {
			auto path = rrtStar.GetPath();
			assert(!path.empty());
			assert((path.front() - start).norm() < 1);
			assert((path.back() - goal).norm() < 1);
			ok++;
		}
	}
	assert(ok >= 1);
	RRTR2 rrt(Point2d(0, 0), Point2d(5, 5));
	rrt.SetInitState(start);
	rrt.SetGoalState(goal);
	rrt.SearchPath(); // default maxIteration = 100 rarely reaches the goal; must not fail to run
	std::printf("rrt*: %d/4 seeds reached the goal\n", ok);
}

int main()
{
	TestHybridAStar();
	TestRRT();
	std::printf("plugin tests ok\n");
	return 0;
}
