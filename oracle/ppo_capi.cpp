// TEST INFRASTRUCTURE ONLY -- C entry points of the CPU oracle for ctypes.
// See ppo_geometry.hpp for the scope / parity-pinning statement.
#include <cstring>
#include "ppo_search.hpp"
#include "ppo_post.hpp"

#include <chrono>
#include <thread>

using namespace ppo;

namespace {
Pose2d P3(const double* p) { return Pose2d(p[0], p[1], p[2]); } // through the wrapping constructor, like user code
Pose2d P3raw(const double* p) { return Pose2d::Raw(p[0], p[1], p[2]); }

struct HybridHandle {
	std::unique_ptr<HybridAStar> algo;
	HybridResult last;
};
struct GridResultHandle {
	GridAStarResult uni;
	BidirResult bi;
	bool bidir = false;
};
}

extern "C" {

// ---------------------------------------------------------------- world ----
void* ppo_world_create(const double* lb, const double* ub, float resolution)
{
	try {
		return new World(P3(lb), P3(ub), resolution);
	} catch (...) {
		return nullptr;
	}
}
void ppo_world_destroy(void* w) { delete (World*)w; }
void ppo_world_dims(void* w, int* rows, int* cols)
{
	*rows = ((World*)w)->rows;
	*cols = ((World*)w)->columns;
}
void ppo_world_origin(void* w, double* out2)
{
	out2[0] = ((World*)w)->worldGridOrigin.x;
	out2[1] = ((World*)w)->worldGridOrigin.y;
}
void ppo_world_set_validator(void* w, float minSafeRadius, float minPathInterpolationDistance)
{
	((World*)w)->minSafeRadius = minSafeRadius;
	((World*)w)->minPathInterpolationDistance = minPathInterpolationDistance;
}
int ppo_world_add_polygon(void* wv, int nverts, const double* xy, const double* pose)
{
	World* w = (World*)wv;
	std::vector<Point2d> v;
	for (int i = 0; i < nverts; i++)
		v.push_back({ xy[2 * i], xy[2 * i + 1] });
	return (int)w->AddObstacleCells(w->PolygonCells(v, P3(pose)));
}
int ppo_world_add_rectangle(void* wv, double dx, double dy, const double* pose)
{
	World* w = (World*)wv;
	return (int)w->AddObstacleCells(w->PolygonCells(World::RectangleVertices(dx, dy), P3(pose)));
}
int ppo_world_add_circle(void* wv, double radius, int count, const double* pose)
{
	World* w = (World*)wv;
	return (int)w->AddObstacleCells(w->PolygonCells(World::CircleVertices(radius, count), P3(pose)));
}
int ppo_world_remove_rectangle(void* wv, int id, double dx, double dy, const double* pose)
{
	World* w = (World*)wv;
	w->RemoveObstacleCells((unsigned)id, w->PolygonCells(World::RectangleVertices(dx, dy), P3(pose)));
	return 0;
}
int ppo_world_polygon_cells(void* wv, int nverts, const double* xy, const double* pose, int cap, int* rc)
{
	World* w = (World*)wv;
	std::vector<Point2d> v;
	for (int i = 0; i < nverts; i++)
		v.push_back({ xy[2 * i], xy[2 * i + 1] });
	auto cells = w->PolygonCells(v, P3(pose));
	for (size_t i = 0; i < cells.size() && (int)i < cap; i++) {
		rc[2 * i] = cells[i].row;
		rc[2 * i + 1] = cells[i].col;
	}
	return (int)cells.size();
}
void ppo_world_update(void* w) { ((World*)w)->UpdateGVD(); }
void ppo_world_get_occ(void* w, int* out) { std::memcpy(out, ((World*)w)->occ->data.data(), ((World*)w)->occ->data.size() * sizeof(int)); }
void ppo_world_get_d2(void* wv, int* out)
{
	World* w = (World*)wv;
	for (int r = 0; r < w->rows; r++)
		for (int c = 0; c < w->columns; c++)
			out[(size_t)r * w->columns + c] = w->Dist2At(r, c);
}
void ppo_world_get_voro_d2(void* w, int* out) { std::memcpy(out, ((World*)w)->voronoiMap->distance.data.data(), ((World*)w)->voronoiMap->distance.data.size() * sizeof(int)); }
void ppo_world_get_pathcost(void* wv, float* out)
{
	World* w = (World*)wv;
	for (int r = 0; r < w->rows; r++)
		for (int c = 0; c < w->columns; c++)
			out[(size_t)r * w->columns + c] = w->PathCostAt(r, c);
}
void ppo_world_set_occ(void* w, const int* in) { std::memcpy(((World*)w)->occ->data.data(), in, ((World*)w)->occ->data.size() * sizeof(int)); }
void ppo_world_set_d2(void* wv, const int* in)
{
	World* w = (World*)wv;
	w->d2Override.assign(in, in + (size_t)w->rows * w->columns);
}
void ppo_world_set_pathcost(void* wv, const float* in)
{
	World* w = (World*)wv;
	w->pathCostOverride.assign(in, in + (size_t)w->rows * w->columns);
}
float ppo_world_distance_value(void* wv, int d2)
{
	// the expression of gvd.h:38 for one squared distance
	World* w = (World*)wv;
	return std::sqrt(d2) * w->resolution;
}

// ------------------------------------------------------------ validator ----
void ppo_is_state_valid(void* wv, int64_t n, const double* poses, uint8_t* out)
{
	World* w = (World*)wv;
	for (int64_t i = 0; i < n; i++)
		out[i] = w->IsStateValid(P3raw(poses + 3 * i)) ? 1 : 0;
}
void ppo_world_to_cell(void* wv, int64_t n, const double* xy, int bounded, int* rc)
{
	World* w = (World*)wv;
	for (int64_t i = 0; i < n; i++) {
		Cell c = w->WorldPositionToGridCell(xy[2 * i], xy[2 * i + 1], bounded != 0);
		rc[2 * i] = c.row;
		rc[2 * i + 1] = c.col;
	}
}
void ppo_constant_steer(int64_t n, const double* from, const double* steering, const double* dist, const int* dir, double wheelbase, double* out)
{
	KinematicBicycleModel m;
	m.wheelbase = wheelbase;
	for (int64_t i = 0; i < n; i++) {
		Pose2d p = m.ConstantSteer(P3raw(from + 3 * i), steering[i], dist[i], (Direction)dir[i]);
		out[3 * i] = p.x;
		out[3 * i + 1] = p.y;
		out[3 * i + 2] = p.theta;
	}
}
double ppo_steering_from_radius(double wheelbase, double radius)
{
	KinematicBicycleModel m;
	m.wheelbase = wheelbase;
	return m.GetSteeringAngleFromTurningRadius(radius);
}
/// d(theta)/d(dist) of ConstantSteer for rearToCenter = 0 (kinematic_bicycle_model.cpp:13-17)
double ppo_curvature_from_steering(double wheelbase, double steering)
{
	double tanSteering = std::tan(steering);
	double beta = std::atan(0.0 * tanSteering / wheelbase);
	double cosBeta = std::cos(beta);
	return cosBeta * tanSteering / wheelbase;
}
void ppo_is_path_valid_csteer(void* wv, int64_t n, const double* from, const double* steering, const double* length, const int* dir, double wheelbase,
	uint8_t* valid, float* last)
{
	World* w = (World*)wv;
	KinematicBicycleModel m;
	m.wheelbase = wheelbase;
	for (int64_t i = 0; i < n; i++) {
		PathConstantSteer path(&m, P3raw(from + 3 * i), steering[i], length[i], (Direction)dir[i]);
		float l = -1.0f;
		valid[i] = w->IsPathValid(path, &l) ? 1 : 0;
		last[i] = l;
	}
}
void ppo_is_path_valid_r2(void* wv, int64_t n, const double* from, const double* to, uint8_t* valid)
{
	World* w = (World*)wv;
	R2Problem prob;
	prob.world = w;
	for (int64_t i = 0; i < n; i++) {
		PathR2 path({ from[2 * i], from[2 * i + 1] }, { to[2 * i], to[2 * i + 1] });
		valid[i] = prob.IsPathValid(path) ? 1 : 0;
	}
}
void ppo_world_counters(void* wv, uint64_t* stateChecks, uint64_t* pathChecks)
{
	*stateChecks = ((World*)wv)->nStateChecks;
	*pathChecks = ((World*)wv)->nPathChecks;
}

// ---------------------------------------------------------- Reeds-Shepp ----
double ppo_rs_shortest(const double* start, const double* goal, double rmin, int* word, double* tuv)
{
	return rs::GetShortestDistance(P3(start), P3(goal), rmin, word, tuv);
}
static void FillSegment(const rs::PathSegment& seg, int* nmotions, int* steer, int* dir, double* len, double* segLength)
{
	if (nmotions)
		*nmotions = seg.GetNumMotions();
	for (int i = 0; i < rs::kNumMotion; i++) {
		if (steer)
			steer[i] = (int)seg.motions[i].steer;
		if (dir)
			dir[i] = (int)seg.motions[i].direction;
		if (len)
			len[i] = seg.motions[i].length;
	}
	if (segLength)
		*segLength = seg.length;
}
void ppo_rs_shortest_path(const double* start, const double* goal, double rmin, int* word, int* nmotions, int* steer, int* dir, double* len, double* segLength, double* finalPose)
{
	int w = -1;
	auto seg = rs::GetShortestPath(P3(start), P3(goal), rmin, &w);
	*word = w;
	FillSegment(seg, nmotions, steer, dir, len, segLength);
	PathReedsShepp path(P3(start), seg, rmin);
	finalPose[0] = path.final.x;
	finalPose[1] = path.final.y;
	finalPose[2] = path.final.theta;
}
void ppo_rs_optimal_batch(int64_t n, const double* starts, const double* goals, double rmin, float rev, float fwd, float sw, int* word, double* tuv, float* cost,
	double* segLength)
{
	for (int64_t i = 0; i < n; i++) {
		int w = -1;
		double t3[3] = { 0, 0, 0 };
		auto seg = rs::GetOptimalPath(P3raw(starts + 3 * i), P3raw(goals + 3 * i), rmin, rev, fwd, sw, &w, t3);
		word[i] = w;
		if (tuv) {
			tuv[3 * i] = t3[0];
			tuv[3 * i + 1] = t3[1];
			tuv[3 * i + 2] = t3[2];
		}
		if (cost)
			cost[i] = seg.ComputeCost(rmin, rev, fwd, sw);
		if (segLength)
			segLength[i] = seg.length;
	}
}
void ppo_rs_segment(int word, const double* tuv, int* nmotions, int* steer, int* dir, double* len, double* segLength)
{
	auto seg = rs::GetPath(word, tuv[0], tuv[1], tuv[2]);
	FillSegment(seg, nmotions, steer, dir, len, segLength);
}
void ppo_rs_interpolate(const double* start, int word, const double* tuv, double rmin, int64_t n, const double* ratios, double* out)
{
	auto seg = rs::GetPath(word, tuv[0], tuv[1], tuv[2]);
	PathReedsShepp path(P3raw(start), seg, rmin);
	for (int64_t i = 0; i < n; i++) {
		Pose2d p = path.Interpolate(ratios[i]);
		out[3 * i] = p.x;
		out[3 * i + 1] = p.y;
		out[3 * i + 2] = p.theta;
	}
}
int ppo_rs_path_valid(void* wv, const double* start, int word, const double* tuv, double rmin)
{
	auto seg = rs::GetPath(word, tuv[0], tuv[1], tuv[2]);
	PathReedsShepp path(P3raw(start), seg, rmin);
	return ((World*)wv)->IsPathValid(path) ? 1 : 0;
}

// --------------------------------------------- PathReedsShepp as a record ----
// Same 128-byte layout as the product ABI's pp_rs_path (include/pp_hip.h), restated here because the oracle shares no
// header with the product: m_init, m_final, the five Motion slots, m_minTurningRadius, m_length, cost / word of Connect.
struct PpoRsPath {
	double start[3];
	double finalPose[3];
	double motionLength[5];
	int8_t steer[5];
	int8_t direction[5];
	int8_t reserved[6];
	double rmin;
	double length;
	float cost;
	int32_t word;
};
static_assert(sizeof(PpoRsPath) == 128, "record layout");
static PathReedsShepp FromRecord(const PpoRsPath& r)
{
	rs::PathSegment seg;
	for (int i = 0; i < rs::kNumMotion; i++) {
		seg.motions[i].steer = (Steer)r.steer[i];
		seg.motions[i].direction = (Direction)r.direction[i];
		seg.motions[i].length = r.motionLength[i];
	}
	PathReedsShepp p(Pose2d::Raw(r.start[0], r.start[1], r.start[2]), seg, r.rmin);
	p.length = r.length; // a truncated path: m_length is no longer the sum of the slots
	p.final = Pose2d::Raw(r.finalPose[0], r.finalPose[1], r.finalPose[2]);
	return p;
}
static void ToRecord(const PathReedsShepp& p, PpoRsPath& r)
{
	r.start[0] = p.init.x, r.start[1] = p.init.y, r.start[2] = p.init.theta;
	r.finalPose[0] = p.final.x, r.finalPose[1] = p.final.y, r.finalPose[2] = p.final.theta;
	for (int i = 0; i < rs::kNumMotion; i++) {
		r.motionLength[i] = p.segment.motions[i].length;
		r.steer[i] = (int8_t)p.segment.motions[i].steer;
		r.direction[i] = (int8_t)p.segment.motions[i].direction;
	}
	r.rmin = p.minTurningRadius;
	r.length = p.length;
}
/// PathConnectionReedsShepp::Connect, paths/path_reeds_shepp.cpp:174-179
void ppo_rs_connect(int64_t n, const double* from, const double* to, double rmin, float rev, float fwd, float sw, PpoRsPath* out)
{
	for (int64_t i = 0; i < n; i++) {
		int w = -1;
		double t3[3] = { 0, 0, 0 };
		auto seg = rs::GetOptimalPath(P3(from + 3 * i), P3(to + 3 * i), rmin, rev, fwd, sw, &w, t3);
		PathReedsShepp p(P3(from + 3 * i), seg, rmin);
		std::memset(&out[i], 0, sizeof(PpoRsPath));
		ToRecord(p, out[i]);
		out[i].cost = seg.ComputeCost(rmin, rev, fwd, sw);
		out[i].word = w;
	}
}
void ppo_rs_path_interpolate(int64_t n, const PpoRsPath* paths, const double* ratio, double* pose, int32_t* direction)
{
	for (int64_t i = 0; i < n; i++) {
		PathReedsShepp p = FromRecord(paths[i]);
		Pose2d s = p.Interpolate(ratio[i]);
		pose[3 * i] = s.x, pose[3 * i + 1] = s.y, pose[3 * i + 2] = s.theta;
		direction[i] = (int32_t)p.GetDirection(ratio[i]);
	}
}
void ppo_rs_path_truncate(int64_t n, PpoRsPath* paths, const double* ratio)
{
	for (int64_t i = 0; i < n; i++) {
		PathReedsShepp p = FromRecord(paths[i]);
		p.Truncate(ratio[i]);
		ToRecord(p, paths[i]);
	}
}
void ppo_rs_path_cusps(int64_t n, const PpoRsPath* paths, double* ratios, int32_t* count)
{
	for (int64_t i = 0; i < n; i++) {
		PathReedsShepp p = FromRecord(paths[i]);
		auto c = p.GetCuspPointRatios();
		int k = 0;
		for (double r : c)
			if (k < 4)
				ratios[4 * i + k++] = r;
		count[i] = (int32_t)c.size();
		for (; k < 4; k++)
			ratios[4 * i + k] = 0.0;
	}
}
void ppo_rs_paths_valid(void* wv, int64_t n, const PpoRsPath* paths, uint8_t* valid, float* last)
{
	World* w = (World*)wv;
	for (int64_t i = 0; i < n; i++) {
		PathReedsShepp p = FromRecord(paths[i]);
		float l = -1.0f;
		valid[i] = w->IsPathValid(p, &l) ? 1 : 0;
		last[i] = l;
	}
}
/// PathSE2, paths/path_se2.cpp:5-22
struct PathSE2Line {
	Pose2d init, final;
	double length = 0.0;
	Pose2d Interpolate(double ratio) const
	{
		// members assigned directly (path_se2.cpp:13-15): theta is not wrapped
		return Pose2d::Raw((1 - ratio) * init.x + ratio * final.x, (1 - ratio) * init.y + ratio * final.y, (1 - ratio) * init.theta + ratio * final.theta);
	}
};
void ppo_se2_paths_valid(void* wv, int64_t n, const double* from, const double* to, uint8_t* valid, float* last)
{
	World* w = (World*)wv;
	for (int64_t i = 0; i < n; i++) {
		PathSE2Line p;
		p.init = P3(from + 3 * i);
		p.final = P3(to + 3 * i);
		p.length = Norm(p.final.x - p.init.x, p.final.y - p.init.y);
		float l = -1.0f;
		valid[i] = w->IsPathValid(p, &l) ? 1 : 0;
		last[i] = l;
	}
}

/// sensitivity probe of the smoother's curvature term (ppo_post.hpp: Smoother::LibmLastBit)
void ppo_smoother_libm_last_bit(int shift)
{
	Smoother::LibmLastBit() = shift;
	Smoother::ProbeCounter() = 0;
}

// --------------------------------------------- post-processing + smoother ----
struct PostHandle {
	std::vector<Pose2d> resampled, smoothed;
	std::vector<uint8_t> cusp;
	std::vector<double> ratios;
	int status = -1, iterations = 0;
	double length = 0.0;
};
/// hybrid_a_star.cpp:260-304 on a solution given as its nodes (root .. goal): hp = {wheelbase, minTurningRadius, reverseCostMultiplier,
/// forwardCostMultiplier, directionSwitchingCost}; sp = {stepTolerance, maxIterations, learningRate, pathWeight, smoothWeight,
/// voronoiWeight, collisionWeight, curvatureWeight, collisionRatio, maxCurvature}; nearestObstacle / nearestEdge: (row, col) per
/// cell, or NULL for the world's own brushfire results.
void* ppo_postprocess(void* wv, const double* hp, int nPath, const double* poses, const int* kind, const double* steering, const double* length, const int* direction,
	const double* goal, float pathInterpolation, const float* sp, const int* nearestObstacle, const int* nearestEdge)
{
	World* w = (World*)wv;
	auto* out = new PostHandle();
	KinematicBicycleModel model;
	model.wheelbase = hp[0];
	PathComposite comp;
	for (int i = 1; i < nPath; i++) {
		Pose2d from = Pose2d::Raw(poses[3 * (i - 1)], poses[3 * (i - 1) + 1], poses[3 * (i - 1) + 2]);
		if (kind[i] == 1) {
			comp.PushBack(std::make_shared<PathConstantSteer>(&model, from, steering[i], length[i], (Direction)direction[i]));
		} else {
			auto seg = rs::GetOptimalPath(from, P3raw(goal), hp[1], (float)hp[2], (float)hp[3], (float)hp[4], nullptr, nullptr);
			comp.PushBack(std::make_shared<PathReedsShepp>(from, seg, hp[1]));
		}
	}
	out->length = comp.length;
	std::unordered_set<int> cuspIndices;
	if (!comp.parts.empty())
		ResampleRatios(comp, pathInterpolation, out->ratios, cuspIndices);
	for (double r : out->ratios)
		out->resampled.push_back(comp.Interpolate(r));
	out->cusp.assign(out->ratios.size(), 0);
	for (int c : cuspIndices)
		if (c >= 0 && c < (int)out->cusp.size())
			out->cusp[c] = 1;
	Grid<Cell> obs(w->rows, w->columns, Cell(-1, -1)), edg(w->rows, w->columns, Cell(-1, -1));
	if (nearestObstacle && nearestEdge) {
		for (int r = 0; r < w->rows; r++)
			for (int c = 0; c < w->columns; c++) {
				size_t i = ((size_t)r * w->columns + c) * 2;
				obs.at(r, c) = Cell(nearestObstacle[i], nearestObstacle[i + 1]);
				edg.at(r, c) = Cell(nearestEdge[i], nearestEdge[i + 1]);
			}
	}
	Smoother sm;
	sm.world = w;
	sm.p.stepTolerance = sp[0];
	sm.p.maxIterations = (int)sp[1];
	sm.p.learningRate = sp[2];
	sm.p.pathWeight = sp[3];
	sm.p.smoothWeight = sp[4];
	sm.p.voronoiWeight = sp[5];
	sm.p.collisionWeight = sp[6];
	sm.p.curvatureWeight = sp[7];
	sm.p.collisionRatio = sp[8];
	sm.p.maxCurvature = sp[9];
	sm.nearestObstacle = nearestObstacle ? &obs : &w->obstacleMap->obstacle;
	sm.nearestEdge = nearestEdge ? &edg : &w->voronoiMap->edge;
	out->status = sm.Smooth(out->resampled, cuspIndices);
	out->iterations = sm.iterations;
	out->smoothed = sm.current;
	return out;
}
void ppo_post_info(void* hv, int* nPoints, int* status, int* iterations, double* length)
{
	auto* h = (PostHandle*)hv;
	*nPoints = (int)h->resampled.size();
	*status = h->status;
	*iterations = h->iterations;
	*length = h->length;
}
void ppo_post_get(void* hv, double* resampled, uint8_t* cusp, double* smoothed, double* ratios)
{
	auto* h = (PostHandle*)hv;
	for (size_t i = 0; i < h->resampled.size(); i++) {
		resampled[3 * i] = h->resampled[i].x, resampled[3 * i + 1] = h->resampled[i].y, resampled[3 * i + 2] = h->resampled[i].theta;
		cusp[i] = h->cusp[i];
		ratios[i] = h->ratios[i];
	}
	for (size_t i = 0; i < h->smoothed.size(); i++)
		smoothed[3 * i] = h->smoothed[i].x, smoothed[3 * i + 1] = h->smoothed[i].y, smoothed[3 * i + 2] = h->smoothed[i].theta;
}
void ppo_post_destroy(void* hv) { delete (PostHandle*)hv; }
void ppo_world_get_nearest(void* wv, int* obstacle, int* edge)
{
	World* w = (World*)wv;
	for (int r = 0; r < w->rows; r++)
		for (int c = 0; c < w->columns; c++) {
			size_t i = ((size_t)r * w->columns + c) * 2;
			obstacle[i] = w->obstacleMap->obstacle.at(r, c).row, obstacle[i + 1] = w->obstacleMap->obstacle.at(r, c).col;
			edge[i] = w->voronoiMap->edge.at(r, c).row, edge[i + 1] = w->voronoiMap->edge.at(r, c).col;
		}
}

// ----------------------------------------------------------- heuristics ----
void ppo_obstacle_heuristic(void* wv, const double* goalxy, double rev, double fwd, int literal, float* cost, uint8_t* explored)
{
	World* w = (World*)wv;
	ObstaclesHeuristic h(w, rev, fwd);
	Pose2d g = Pose2d::Raw(goalxy[0], goalxy[1], 0.0);
	if (literal)
		h.UpdateLiteral(g);
	else
		h.Update(g);
	std::memcpy(cost, h.cost.data(), h.cost.size() * sizeof(float));
	if (explored)
		std::memcpy(explored, h.explored.data(), h.explored.size());
}
static HybridParams ParamsFrom(const double* p)
{
	HybridParams hp;
	hp.wheelbase = p[0];
	hp.minTurningRadius = p[1];
	hp.directionSwitchingCost = p[2];
	hp.reverseCostMultiplier = p[3];
	hp.forwardCostMultiplier = p[4];
	hp.voronoiCostMultiplier = p[5];
	hp.numGeneratedMotion = (unsigned int)p[6];
	hp.spatialResolution = p[7];
	hp.angularResolution = p[8];
	return hp;
}
/// dims = {nX, nY, nAngular}; `out` may be null to query dims only.  `threads`
/// > 1 splits the i-loop (entries are independent; results identical).
void ppo_nonholo_build(const double* lb, const double* ub, const double* params, int* dims, double* offsets, double* out, int threads)
{
	HybridParams hp = ParamsFrom(params);
	Pose2d l = P3(lb), u = P3(ub);
	unsigned int nx = std::ceil((u.x - l.x) / hp.spatialResolution);
	if (nx % 2 == 0)
		nx++;
	unsigned int ny = std::ceil((u.y - l.y) / hp.spatialResolution);
	if (ny % 2 == 0)
		ny++;
	unsigned int na = std::ceil(2 * M_PI / hp.angularResolution);
	dims[0] = nx;
	dims[1] = ny;
	dims[2] = na;
	double offX = std::floor(nx / 2.0) * hp.spatialResolution, offY = std::floor(ny / 2.0) * hp.spatialResolution;
	if (offsets) {
		offsets[0] = offX;
		offsets[1] = offY;
	}
	if (!out)
		return;
	auto work = [&](int i0, int i1) {
		const Pose2d goal0(0.0, 0.0, 0.0);
		for (int i = i0; i < i1; i++)
			for (int j = 0; j < (int)ny; j++)
				for (int k = 0; k < (int)na; k++) {
					Pose2d pose(i * hp.spatialResolution - offX, j * hp.spatialResolution - offY, k * hp.angularResolution);
					auto path = rs::GetOptimalPath(pose, goal0, hp.minTurningRadius, hp.reverseCostMultiplier, hp.forwardCostMultiplier, hp.directionSwitchingCost);
					out[((size_t)i * ny + j) * na + k] = path.ComputeCost(hp.minTurningRadius, hp.reverseCostMultiplier, hp.forwardCostMultiplier, hp.directionSwitchingCost);
				}
	};
	if (threads <= 1) {
		work(0, nx);
	} else {
		std::vector<std::thread> pool;
		int chunk = (nx + threads - 1) / threads;
		for (int t = 0; t < threads; t++) {
			int i0 = t * chunk, i1 = std::min<int>(nx, i0 + chunk);
			if (i0 < i1)
				pool.emplace_back(work, i0, i1);
		}
		for (auto& th : pool)
			th.join();
	}
}

// ------------------------------------------------------------ Hybrid A* ----
void* ppo_hybrid_create(void* wv, const double* params, int headingAlias, int negativeKRead)
{
	auto* h = new HybridHandle();
	Quirks q;
	q.headingAlias = headingAlias != 0;
	q.negativeKRead = negativeKRead != 0;
	h->algo = std::make_unique<HybridAStar>((World*)wv, ParamsFrom(params), q);
	return h;
}
void ppo_hybrid_destroy(void* h) { delete (HybridHandle*)h; }
/// `table` (nX*nY*nA doubles) may be null: then the table is built here.
void ppo_hybrid_initialize(void* hv, const double* table)
{
	auto* h = (HybridHandle*)hv;
	if (!table) {
		h->algo->Initialize();
		return;
	}
	const World* w = h->algo->world;
	const HybridParams& hp = h->algo->param;
	auto nh = std::make_unique<NonHolonomicHeuristic>();
	nh->spatialResolution = hp.spatialResolution;
	nh->angularResolution = hp.angularResolution;
	nh->minTurningRadius = hp.minTurningRadius;
	nh->reverseCostMultiplier = hp.reverseCostMultiplier;
	nh->forwardCostMultiplier = hp.forwardCostMultiplier;
	nh->directionSwitchingCost = hp.directionSwitchingCost;
	unsigned int nx = std::ceil((w->ub.x - w->lb.x) / hp.spatialResolution);
	if (nx % 2 == 0)
		nx++;
	unsigned int ny = std::ceil((w->ub.y - w->lb.y) / hp.spatialResolution);
	if (ny % 2 == 0)
		ny++;
	nh->numSpatialX = nx;
	nh->numSpatialY = ny;
	nh->numAngular = std::ceil(2 * M_PI / hp.angularResolution);
	nh->offsetX = std::floor(nx / 2.0) * hp.spatialResolution;
	nh->offsetY = std::floor(ny / 2.0) * hp.spatialResolution;
	nh->values.assign(table, table + (size_t)nx * ny * nh->numAngular);
	h->algo->Initialize(std::move(nh));
}
int ppo_hybrid_num_primitives(void* hv) { return 2 * (int)((HybridHandle*)hv)->algo->deltas.size(); }
void ppo_hybrid_deltas(void* hv, double* out)
{
	auto& d = ((HybridHandle*)hv)->algo->deltas;
	for (size_t i = 0; i < d.size(); i++)
		out[i] = d[i];
}
/// Replaces m_deltas (hybrid_a_star.cpp:21-28 can only generate {0, +-0.5 dMax, +-1.0 dMax, ...}, i.e. 2 * odd primitives): any list of
/// steering angles, children in list order, forward then backward each (hybrid_a_star.cpp:65-77) -- SURVEY 8d config 2's "P = 72"
void ppo_hybrid_set_deltas(void* hv, int n, const double* deltas) { ((HybridHandle*)hv)->algo->deltas.assign(deltas, deltas + n); }
void ppo_hybrid_set_max_expansions(void* hv, int64_t n) { ((HybridHandle*)hv)->algo->maxExpansions = n < 0 ? (size_t)-1 : (size_t)n; }
/// Runs ObstaclesHeuristic::Update for `goal` and sets both heuristics' goal.
void ppo_hybrid_set_goal(void* hv, const double* goal)
{
	auto* h = (HybridHandle*)hv;
	Pose2d g = P3(goal);
	h->algo->goalPose = g;
	h->algo->obstacle->Update(g);
	h->algo->obstacle->goal = g;
	h->algo->nonHolo->goal = g;
}
void ppo_hybrid_heuristic(void* hv, int64_t n, const double* poses, double* nonholo, double* obst, double* combined)
{
	auto* h = (HybridHandle*)hv;
	for (int64_t i = 0; i < n; i++) {
		Pose2d p = P3raw(poses + 3 * i);
		if (nonholo)
			nonholo[i] = h->algo->nonHolo->GetHeuristicValue(p);
		if (obst)
			obst[i] = h->algo->obstacle->GetHeuristicValue(p);
		if (combined)
			combined[i] = h->algo->Heuristic(p);
	}
}
void ppo_hybrid_discretize(void* hv, int64_t n, const double* poses, int* out)
{
	auto* h = (HybridHandle*)hv;
	for (int64_t i = 0; i < n; i++) {
		DiscretePose d = h->algo->DiscretizePose(P3raw(poses + 3 * i));
		out[3 * i] = d.x;
		out[3 * i + 1] = d.y;
		out[3 * i + 2] = d.theta;
	}
}
/// Constant-steer children of each parent in reference order (delta-major,
/// Forward then Backward).  Output slot p*P + c; valid[] = 0 where the
/// reference drops the child.
void ppo_hybrid_children(void* hv, int64_t nParents, const double* parents, uint8_t* valid, double* poses, int* keys, double* cost, double* length)
{
	auto* h = (HybridHandle*)hv;
	const int P = 2 * (int)h->algo->deltas.size();
	for (int64_t p = 0; p < nParents; p++) {
		HybridNode parent;
		parent.pose = P3raw(parents + 3 * p);
		parent.discrete = h->algo->DiscretizePose(parent.pose);
		int c = 0;
		for (double delta : h->algo->deltas)
			for (int d = 0; d < 2; d++, c++) {
				HybridAStar::Child ch;
				bool ok = h->algo->GetConstantSteerChild(parent, delta, d == 0 ? Direction::Forward : Direction::Backward, ch);
				size_t o = (size_t)p * P + c;
				valid[o] = ok ? 1 : 0;
				poses[3 * o] = ch.node.pose.x;
				poses[3 * o + 1] = ch.node.pose.y;
				poses[3 * o + 2] = ch.node.pose.theta;
				keys[3 * o] = ch.node.discrete.x;
				keys[3 * o + 1] = ch.node.discrete.y;
				keys[3 * o + 2] = ch.node.discrete.theta;
				cost[o] = ok ? ch.cost : 0.0;
				length[o] = ok ? ch.node.length : 0.0;
			}
	}
}
int ppo_hybrid_search(void* hv, const double* start, const double* goal, uint64_t seed)
{
	auto* h = (HybridHandle*)hv;
	h->last = h->algo->Search(P3(start), P3(goal), seed);
	return h->last.status;
}
/// info = {status, nExpanded, nPathNodes, nNodes, nStateChecks, nPathChecks, nRngDraws, nRsAttempts, nChildren, nLatticeBoundary}
void ppo_hybrid_result_info(void* hv, int64_t* info, double* cost)
{
	auto* h = (HybridHandle*)hv;
	const HybridResult& r = h->last;
	info[0] = r.status;
	info[1] = (int64_t)r.expanded.size();
	info[2] = (int64_t)r.pathNodes.size();
	info[3] = (int64_t)h->algo->nodes.size();
	info[4] = (int64_t)r.nStateChecks;
	info[5] = (int64_t)r.nPathChecks;
	info[6] = (int64_t)r.nRngDraws;
	info[7] = (int64_t)r.nRsAttempts;
	info[8] = (int64_t)r.nChildren;
	info[9] = (int64_t)r.nLatticeBoundary;
	*cost = r.cost;
}
void ppo_hybrid_result_expanded(void* hv, int* cells3)
{
	auto* h = (HybridHandle*)hv;
	for (size_t i = 0; i < h->last.expanded.size(); i++) {
		cells3[3 * i] = h->last.expanded[i].x;
		cells3[3 * i + 1] = h->last.expanded[i].y;
		cells3[3 * i + 2] = h->last.expanded[i].theta;
	}
}
/// Design studies: records the open-list pushes / pops of the next searches on this thread.
static std::vector<FrontierEvent> g_trace;
void ppo_trace_begin()
{
	g_trace.clear();
	FrontierTraceSink() = &g_trace;
}
int64_t ppo_trace_end(int* kinds, double* costs, uint64_t* seqs, int64_t cap)
{
	FrontierTraceSink() = nullptr;
	const int64_t n = (int64_t)g_trace.size();
	for (int64_t i = 0; i < n && i < cap; i++) {
		kinds[i] = g_trace[i].kind;
		costs[i] = g_trace[i].cost;
		seqs[i] = g_trace[i].seq;
	}
	return n;
}
/// Every node of the search tree in creation order: parent index, pose (3), {pathCost, totalCost}, dead flag.
void ppo_hybrid_nodes(void* hv, int* parents, double* poses, double* costs, int* dead)
{
	auto* h = (HybridHandle*)hv;
	for (size_t i = 0; i < h->algo->nodes.size(); i++) {
		const HybridNode& n = h->algo->nodes[i];
		parents[i] = n.parent;
		poses[3 * i] = n.pose.x;
		poses[3 * i + 1] = n.pose.y;
		poses[3 * i + 2] = n.pose.theta;
		costs[2 * i] = n.pathCost;
		costs[2 * i + 1] = n.totalCost;
		dead[i] = n.dead ? 1 : 0;
	}
}
/// Per path node: pose (3), kind, steering, length, direction, rsWord, pathCost.
void ppo_hybrid_result_path(void* hv, double* poses, int* kind, double* steering, double* length, int* direction, int* rsWord, double* pathCost)
{
	auto* h = (HybridHandle*)hv;
	for (size_t i = 0; i < h->last.pathNodes.size(); i++) {
		const HybridNode& n = h->algo->nodes[h->last.pathNodes[i]];
		poses[3 * i] = n.pose.x;
		poses[3 * i + 1] = n.pose.y;
		poses[3 * i + 2] = n.pose.theta;
		kind[i] = n.kind;
		steering[i] = n.steering;
		length[i] = n.length;
		direction[i] = (int)n.actionDirection;
		rsWord[i] = n.rsWord;
		pathCost[i] = n.pathCost;
	}
}
/// Runs `n` queries back to back on `threads` threads (each thread owns a
/// HybridAStar sharing the read-only world + a copy of the table); returns
/// wall seconds.  The cpu_baseline leg of bench.py.
double ppo_hybrid_batch(void* wv, const double* params, int headingAlias, int negativeKRead, const double* table, int64_t n, const double* starts,
	const double* goals, const uint64_t* seeds, int threads, int* status, double* cost, int64_t* nExpanded)
{
	World* w = (World*)wv;
	if (threads < 1)
		threads = 1;
	std::vector<std::unique_ptr<HybridHandle>> hs;
	for (int t = 0; t < threads; t++) {
		hs.emplace_back((HybridHandle*)ppo_hybrid_create(w, params, headingAlias, negativeKRead));
		ppo_hybrid_initialize(hs.back().get(), table);
	}
	auto t0 = std::chrono::steady_clock::now();
	auto work = [&](int t) {
		for (int64_t i = t; i < n; i += threads) {
			HybridResult r = hs[t]->algo->Search(P3(starts + 3 * i), P3(goals + 3 * i), seeds[i]);
			if (status)
				status[i] = r.status;
			if (cost)
				cost[i] = r.cost;
			if (nExpanded)
				nExpanded[i] = (int64_t)r.expanded.size();
		}
	};
	if (threads == 1)
		work(0);
	else {
		std::vector<std::thread> pool;
		for (int t = 0; t < threads; t++)
			pool.emplace_back(work, t);
		for (auto& th : pool)
			th.join();
	}
	auto t1 = std::chrono::steady_clock::now();
	return std::chrono::duration<double>(t1 - t0).count();
}

/// ppo_hybrid_batch that also returns every solution path (GetGraphSearchPath, hybrid_a_star.h:223: the poses of the nodes root..solution):
/// poses [n][maxPoses][3], nPoses [n] (0: no solution; a longer path is cut at maxPoses).  bench.py's cpu_baseline leg checks the paths the
/// GPU pipeline delivered against these.
double ppo_hybrid_batch_paths(void* wv, const double* params, int headingAlias, int negativeKRead, const double* table, int64_t n, const double* starts,
	const double* goals, const uint64_t* seeds, int threads, int* status, double* cost, int64_t* nExpanded, int maxPoses, double* poses, int* nPoses)
{
	World* w = (World*)wv;
	if (threads < 1)
		threads = 1;
	std::vector<std::unique_ptr<HybridHandle>> hs;
	for (int t = 0; t < threads; t++) {
		hs.emplace_back((HybridHandle*)ppo_hybrid_create(w, params, headingAlias, negativeKRead));
		ppo_hybrid_initialize(hs.back().get(), table);
	}
	auto t0 = std::chrono::steady_clock::now();
	auto work = [&](int t) {
		for (int64_t i = t; i < n; i += threads) {
			HybridResult r = hs[t]->algo->Search(P3(starts + 3 * i), P3(goals + 3 * i), seeds[i]);
			status[i] = r.status;
			cost[i] = r.cost;
			nExpanded[i] = (int64_t)r.expanded.size();
			nPoses[i] = r.status == 0 ? (int)r.pathNodes.size() : 0;
			for (int k = 0; k < nPoses[i] && k < maxPoses; k++) {
				const HybridNode& nd = hs[t]->algo->nodes[r.pathNodes[(size_t)k]];
				double* o = poses + ((size_t)i * (size_t)maxPoses + (size_t)k) * 3;
				o[0] = nd.pose.x;
				o[1] = nd.pose.y;
				o[2] = nd.pose.theta;
			}
		}
	};
	std::vector<std::thread> pool;
	for (int t = 0; t < threads; t++)
		pool.emplace_back(work, t);
	for (auto& th : pool)
		th.join();
	auto t1 = std::chrono::steady_clock::now();
	return std::chrono::duration<double>(t1 - t0).count();
}

// -------------------------------------------------------------- grid A* ----
typedef double (*ppo_cell_fn)(int, int, int, int);
static double EuclidCells(const Cell& a, const Cell& b)
{
	// the cost / heuristic of interfaces/python/scripts/example_a_star_grid.py:46-52
	double dr = a.row - b.row, dc = a.col - b.col;
	return std::sqrt(dr * dr + dc * dc);
}
void* ppo_grid_astar(void* wv, const int* init, const int* goal, int bidirectional, const int* innerGoalF, const int* innerGoalR, ppo_cell_fn costFn,
	ppo_cell_fn heurFn)
{
	auto* res = new GridResultHandle();
	CellFn cf = costFn ? CellFn([costFn](const Cell& a, const Cell& b) { return costFn(a.row, a.col, b.row, b.col); }) : CellFn(EuclidCells);
	CellFn hf = heurFn ? CellFn([heurFn](const Cell& a, const Cell& b) { return heurFn(a.row, a.col, b.row, b.col); }) : CellFn(EuclidCells);
	Cell i(init[0], init[1]), g(goal[0], goal[1]);
	res->bidir = bidirectional != 0;
	if (!res->bidir)
		res->uni = GridAStar((World*)wv, i, g, cf, hf);
	else
		res->bi = BidirectionalGridAStar((World*)wv, i, g, cf, hf, Cell(innerGoalF[0], innerGoalF[1]), Cell(innerGoalR[0], innerGoalR[1]));
	return res;
}
void ppo_grid_result_info(void* rv, int* status, double* cost, int* nPath, int* nExplored, int* nExploredR)
{
	auto* r = (GridResultHandle*)rv;
	if (!r->bidir) {
		*status = r->uni.status;
		*cost = r->uni.cost;
		*nPath = (int)r->uni.path.size();
		*nExplored = (int)r->uni.explored.size();
		*nExploredR = 0;
	} else {
		*status = r->bi.status;
		*cost = r->bi.cost;
		*nPath = (int)r->bi.path.size();
		*nExplored = (int)r->bi.fExplored.size();
		*nExploredR = (int)r->bi.rExplored.size();
	}
}
void ppo_grid_result_get(void* rv, int* path, int* explored, int* exploredR)
{
	auto* r = (GridResultHandle*)rv;
	auto put = [](const std::vector<Cell>& v, int* out) {
		if (!out)
			return;
		for (size_t i = 0; i < v.size(); i++) {
			out[2 * i] = v[i].row;
			out[2 * i + 1] = v[i].col;
		}
	};
	if (!r->bidir) {
		put(r->uni.path, path);
		put(r->uni.explored, explored);
	} else {
		put(r->bi.path, path);
		put(r->bi.fExplored, explored);
		put(r->bi.rExplored, exploredR);
	}
}
void ppo_grid_result_destroy(void* rv) { delete (GridResultHandle*)rv; }

// ------------------------------------------------------------- Tree kNN ----
/// utils/tree.h:73-116 on the oracle's PointTree: points inserted in order (root, then Extend from the previous node),
/// k nearest of `query` in ascending squared distance; returns how many were found.
int ppo_tree_knn(int n, const double* points, const double* query, int k, int* outIdx)
{
	PointTree tree;
	for (int i = 0; i < n; i++) {
		Point2d p { points[2 * i], points[2 * i + 1] };
		if (i == 0)
			tree.CreateRoot(p);
		else
			tree.Extend(p, i - 1);
	}
	std::vector<int> nn = tree.Nearest(Point2d { query[0], query[1] }, (unsigned int)k);
	for (size_t i = 0; i < nn.size(); i++)
		outIdx[i] = nn[i];
	return (int)nn.size();
}

// ----------------------------------------------------------- RRT / RRT* ----
/// params = {maxIteration, maxNumberTreeNode, maxConnectionDistance, goalBias}
void* ppo_rrt(void* wv, const double* lb, const double* ub, const double* params, const double* init, const double* goal, uint64_t seed, int star)
{
	R2Problem prob;
	prob.lb = { lb[0], lb[1] };
	prob.ub = { ub[0], ub[1] };
	prob.world = (World*)wv;
	RRTParams p;
	p.maxIteration = (unsigned int)params[0];
	p.maxNumberTreeNode = (unsigned int)params[1];
	p.maxConnectionDistance = params[2];
	p.goalBias = params[3];
	// star: 0 RRT, 1 RRT* (reference), 2 / 3 the rewire extensions (gamma = params[4] for 3)
	auto* res = new RRTResult(star ? RRTStar(prob, p, { init[0], init[1] }, { goal[0], goal[1] }, seed, star, star == 3 ? params[4] : 0.0)
	                               : RRT(prob, p, { init[0], init[1] }, { goal[0], goal[1] }, seed));
	return res;
}
/// info = {status, nNodes, nPath, iterations, nKnnQueries, nEdgeChecks}
void ppo_rrt_result_info(void* rv, int64_t* info)
{
	auto* r = (RRTResult*)rv;
	info[0] = r->status;
	info[1] = (int64_t)r->nodes.size();
	info[2] = (int64_t)r->path.size();
	info[3] = (int64_t)r->iterations;
	info[4] = (int64_t)r->nKnnQueries;
	info[5] = (int64_t)r->nEdgeChecks;
}
void ppo_rrt_result_get(void* rv, double* nodes, int* parents, double* costs, double* path)
{
	auto* r = (RRTResult*)rv;
	for (size_t i = 0; i < r->nodes.size(); i++) {
		if (nodes) {
			nodes[2 * i] = r->nodes[i].x;
			nodes[2 * i + 1] = r->nodes[i].y;
		}
		if (parents)
			parents[i] = r->parents[i];
		if (costs)
			costs[i] = r->costs[i];
	}
	for (size_t i = 0; i < r->path.size() && path; i++) {
		path[2 * i] = r->path[i].x;
		path[2 * i + 1] = r->path[i].y;
	}
}
void ppo_rrt_result_destroy(void* rv) { delete (RRTResult*)rv; }

// ------------------------------------------------- open list / RNG pins ----
/// Pushes (cost[i], id = i) in order, interleaved with pops where ops[i] < 0
/// (ops[i] >= 0: push entry ops[i]; ops[i] == -1: pop).  Writes popped ids.
/// mode 0 = (cost,-seq) heap, 1 = literal sorted-vector Frontier.
int ppo_frontier_replay(int mode, int nops, const int* ops, const double* costs, int* popped)
{
	int np = 0;
	if (mode == 0) {
		LifoHeap<double> h;
		for (int i = 0; i < nops; i++) {
			if (ops[i] >= 0)
				h.Push(costs[ops[i]], (uint32_t)ops[i]);
			else if (!h.Empty())
				popped[np++] = (int)h.Pop().id;
		}
	} else {
		struct E {
			double cost;
			int id;
		};
		struct Cmp {
			bool operator()(const E& a, const E& b) const { return a.cost > b.cost; } // a_star.h:226-231 CompareNode
		};
		SortedFrontier<E, Cmp> f;
		for (int i = 0; i < nops; i++) {
			if (ops[i] >= 0)
				f.Push({ costs[ops[i]], ops[i] });
			else if (!f.Empty())
				popped[np++] = f.Pop().id;
		}
	}
	return np;
}
void ppo_neighbors(int row, int col, int rows, int cols, int* n, int* rc)
{
	Cell nb[8];
	*n = GetNeighbors(Cell(row, col), rows, cols, nb);
	for (int i = 0; i < *n; i++) {
		rc[2 * i] = nb[i].row;
		rc[2 * i + 1] = nb[i].col;
	}
}
void ppo_rng_uniform(uint64_t seed, int64_t n, double lb, double ub, double* out)
{
	Rng rng(seed);
	for (int64_t i = 0; i < n; i++)
		out[i] = rng.SampleUniform(lb, ub);
}
int ppo_alias_heading_bin(int k) { return AliasHeadingBin(k); }
void ppo_pose_between(const double* lhs, const double* rhs, double* out)
{
	Pose2d d = Between(P3raw(lhs), P3raw(rhs));
	out[0] = d.x;
	out[1] = d.y;
	out[2] = d.theta;
}
void ppo_pose_compose(const double* lhs, const double* rhs, double* out)
{
	Pose2d d = Compose(P3raw(lhs), P3raw(rhs));
	out[0] = d.x;
	out[1] = d.y;
	out[2] = d.theta;
}

} // extern "C"
