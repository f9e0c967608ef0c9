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

// planner/tests/test_reeds_shepp.cpp:7-27 (the checker, on a few of its vectors) through PathConnectionReedsShepp, and the
// generic IsPathValid(const Path&, float*) of the validator interface over each path type of the reference.
static void TestPaths()
{
	std::array<Pose2d, 2> bounds = { Pose2d(-10, -10, -M_PI), Pose2d(10, 10, M_PI) };
	Ref<StateSpaceSE2> stateSpace = makeRef<StateSpaceSE2>(bounds);
	Ref<OccupancyMap> map = makeRef<OccupancyMap>(0.1f);
	Ref<StateValidatorOccupancyMap> validator = makeRef<StateValidatorOccupancyMap>(stateSpace, map);
	PathConnectionReedsShepp connection(1.0);
	const Pose2d starts[3] = { Pose2d(0, 0, 0), Pose2d(1, -2, 0.5), Pose2d(-3, 2, -2.0) };
	const Pose2d goals[3] = { Pose2d(4, 4, M_PI / 2), Pose2d(-2, 3, -1.0), Pose2d(0.5, 0.25, 3.0) };
	for (int i = 0; i < 3; i++) {
		Ref<PathSE2Base> path = connection.Connect(starts[i], goals[i]);
		const Pose2d& end = path->GetFinalState();
		assert((end.position - goals[i].position).norm() < 1e-6); // test_reeds_shepp.cpp:20-22
		assert(std::fabs(Pose2d(0, 0, end.theta - goals[i].theta).theta) < 1e-6);
		assert(path->GetLength() > 0);
		float last = -1.0f;
		assert(validator->IsPathValid(*path, &last) && last == 1.0f); // free map: valid all the way
		auto* rs = dynamic_cast<PathReedsShepp*>(path.get());
		assert(rs && rs->GetDirection(0.0) != Direction::NoMotion);
		const double before = rs->GetLength();
		rs->Truncate(0.5);
		assert(std::fabs(rs->GetLength() - 0.5 * before) < 1e-12);
	}
	// a path that leaves the bounds is cut where the last valid sample was
	PathSE2 out(Pose2d(0, 0, 0), Pose2d(20, 0, 0));
	float last = -1.0f;
	assert(!validator->IsPathValid(out, &last) && last > 0.3f && last <= 0.5f);
	Pose2d lastState;
	assert(!static_cast<StateValidatorSE2Base&>(*validator).IsPathValid(out, &lastState) && lastState.x() <= 10.0 && lastState.x() > 6.0);
	auto model = makeRef<KinematicBicycleModel>(2.6, 0.0);
	PathConstantSteer arc(model, Pose2d(0, 0, 0), 0.3, 3.0, Direction::Forward);
	assert(validator->IsPathValid(arc, &last) && last == 1.0f);
	StateValidatorSE2Free freeValidator(stateSpace);
	assert(freeValidator.IsPathValid(out, &last) && last == 1.0f);
	std::printf("paths: Reeds-Shepp connections reach their goals, IsPathValid over RS / SE2 / constant-steer paths\n");
}

// the streaming form of the same planner: 40 queries through a pipeline of 16 slots; every result must be what SearchPath gives
static void TestPipeline()
{
	std::array<Pose2d, 2> bounds = { Pose2d(-10, -10, -M_PI), Pose2d(10, 10, M_PI) };
	Ref<StateSpaceSE2> stateSpace = makeRef<StateSpaceSE2>(bounds);
	Ref<OccupancyMap> map = makeRef<OccupancyMap>(0.1f);
	Ref<StateValidatorOccupancyMap> validator = makeRef<StateValidatorOccupancyMap>(stateSpace, map);
	HybridAStar single;
	assert(single.Initialize(validator));
	HybridAStarPipeline pipe(HybridAStar::SearchParameters(), /*capacity=*/16, /*maxNodes=*/32768, /*searchRows=*/8);
	assert(pipe.Initialize(validator));
	const int n = 40;
	std::vector<Pose2d> starts, goals;
	std::vector<uint64_t> seeds;
	std::vector<double> cost((size_t)n);
	std::vector<char> ok((size_t)n);
	int solved = 0;
	for (int i = 0; i < n; i++) {
		starts.push_back(Pose2d(-8.0 + 0.37 * i, -7.5 + 0.21 * i, 0.1 * i));
		goals.push_back(Pose2d(8.0 - 0.29 * i, 7.0 - 0.33 * i, 0.78 - 0.05 * i));
		seeds.push_back(100 + i);
		single.SetInitState(starts.back());
		single.SetGoalState(goals.back());
		single.SetSeed(seeds.back());
		ok[(size_t)i] = single.SearchPath() == Status::Success;
		cost[(size_t)i] = single.GetGraphSearchOptimalCost();
		solved += ok[(size_t)i];
	}
	assert(solved >= n / 2);
	int submitted = 0, done = 0;
	std::vector<HybridAStarPipeline::Result> res;
	while (done < n) {
		if (submitted < n && pipe.FreeSlots() > 0) {
			std::vector<Pose2d> s(starts.begin() + submitted, starts.end()), g(goals.begin() + submitted, goals.end());
			std::vector<uint64_t> sd(seeds.begin() + submitted, seeds.end());
			submitted += pipe.Submit(s, g, sd);
		}
		pipe.Poll(res);
		for (const auto& r : res) { // tickets count submissions: ticket t is query t
			assert(r.ticket < (uint64_t)n && (r.status == Status::Success) == (bool)ok[(size_t)r.ticket] && (!ok[(size_t)r.ticket] || r.cost == cost[(size_t)r.ticket]));
			done++;
		}
	}
	assert(pipe.InFlight() == 0 && pipe.FreeSlots() == 16);
	std::printf("pipeline: %d queries through 16 slots (%d solved), statuses and costs equal to SearchPath's\n", n, solved);
}

// The pipeline in a process whose environment may not give the HIP runtime enough hardware queues (include/pp_hip.h, pp_pipeline_create): it
// must either run at full speed -- the 40-query test above passes -- or refuse loudly, naming GPU_MAX_HW_QUEUES; never limp along in silence.
static int TestQueuePrecondition()
{
	std::array<Pose2d, 2> bounds = { Pose2d(-10, -10, -M_PI), Pose2d(10, 10, M_PI) };
	Ref<StateSpaceSE2> stateSpace = makeRef<StateSpaceSE2>(bounds);
	Ref<OccupancyMap> map = makeRef<OccupancyMap>(0.1f);
	Ref<StateValidatorOccupancyMap> validator = makeRef<StateValidatorOccupancyMap>(stateSpace, map);
	{
		HybridAStarPipeline pipe(HybridAStar::SearchParameters(), /*capacity=*/16, /*maxNodes=*/32768, /*searchRows=*/8);
		if (!pipe.Initialize(validator)) {
			std::printf("pipeline refused: %s\n", pp_last_error());
			return 0;
		}
	} // (destroyed: the test below makes its own, and two pipelines at once need twice the hardware queues)
	TestPipeline();
	std::printf("pipeline accepted and ran\n");
	return 0;
}

int main(int argc, char** argv)
{
	if (argc > 1 && std::string(argv[1]) == "--queues")
		return TestQueuePrecondition();
	TestPipeline();
	TestHybridAStar();
	TestRRT();
	TestPaths();
	std::printf("plugin tests ok\n");
	return 0;
}
