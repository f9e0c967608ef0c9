// Host-side C++ mirror of the reference's plugin surface for the hot path, over the C ABI
// (include/pp_hip.h).  Same namespace, class and method names as lfilipozzi/PathPlanning so that
// code written against PathPlanner<> / StateValidator<> keeps compiling:
//   algo/path_planner.h:9-45            Status, PathPlanner<Vertex>, PathPlannerSE2Base / R2Base
//   state_space/state_space_se2.h       StateSpaceSE2 (bounds, ValidateBounds, EnforceBounds)
//   state_validator/state_validator.h   StateValidator<State, Dim, T>
//   state_validator/occupancy_map.h     OccupancyMap (sizes, transforms, grids)
//   state_validator/state_validator_occupancy_map.h   StateValidatorOccupancyMap
//   algo/hybrid_a_star.h:27-264         HybridAStar (SearchParameters, Initialize, SearchPath, GetPath, ...)
//   algo/rrt.h, algo/rrt_star.h         RRTR2 / RRTStarR2 (+ parameters)
// No Eigen: Point2d / Pose2d are plain structs with the accessors the reference code uses.
// Everything computes on the GPU; constructing any of the HIP-backed classes without a device throws.
#pragma once

#include <random>
#include <limits>
#include <cstdio>

#include "types.hpp"
#include "paths.hpp"

namespace Planner {

/// algo/path_planner.h:19-41
template <typename Vertex>
class PathPlanner {
public:
	PathPlanner() { }
	virtual ~PathPlanner() = default;
	virtual Status SearchPath() = 0;
	virtual std::vector<Vertex> GetPath() const = 0;
	void SetInitState(const Vertex& init) { m_init = init; }
	void SetGoalState(const Vertex& goal) { m_goal = goal; }
	const Vertex& GetInitState() const { return m_init; }
	const Vertex& GetGoalState() const { return m_goal; }

protected:
	Vertex m_init;
	Vertex m_goal;
};
using PathPlannerR2Base = PathPlanner<Point2d>;
using PathPlannerSE2Base = PathPlanner<Pose2d>;

/// state_space/state_space_se2.{h,cpp}
/// utils/random.h:9-44: the process-wide std::mt19937_64 with a uniform [0, 1] and a standard normal distribution.  Same
/// standard-library objects as the reference, so the same seed gives the same stream.  (The searches do not draw from it: every
/// query carries its own engine on the device.)  Seed() is not in the reference, which seeds from std::random_device only.
template <typename T>
class Random {
public:
	static void Init()
	{
		if (!State().ready) {
			State().engine.seed(std::random_device {}());
			Reset();
		}
	}
	static void Seed(unsigned long long seed)
	{
		State().engine.seed(seed);
		Reset();
	}
	static T SampleUniform(T lb, T ub)
	{
		Init();
		const T range = ub - lb;
		return lb + range * State().uniform(State().engine);
	}
	static T SampleGaussian(const T& mean, const T& stdDev)
	{
		Init();
		return mean + stdDev * State().gaussian(State().engine);
	}

private:
	struct S {
		std::mt19937_64 engine;
		std::uniform_real_distribution<T> uniform;
		std::normal_distribution<T> gaussian;
		bool ready = false;
	};
	static S& State()
	{
		static S s;
		return s;
	}
	static void Reset()
	{
		State().uniform = std::uniform_real_distribution<T>(0.0, std::nextafter(1.0, std::numeric_limits<T>::max()));
		State().gaussian = std::normal_distribution<T>(0.0, 1.0);
		State().ready = true;
	}
};

class StateSpaceSE2 {
public:
	explicit StateSpaceSE2(const std::array<Pose2d, 2>& b) : bounds(b) { }
	StateSpaceSE2(const Pose2d& lb, const Pose2d& ub) : bounds({ lb, ub }) { }
	void EnforceBounds(Pose2d& s) const
	{
		s.x() = std::min(std::max(s.x(), bounds[0].x()), bounds[1].x());
		s.y() = std::min(std::max(s.y(), bounds[0].y()), bounds[1].y());
		s.theta = std::min(std::max(s.theta, bounds[0].theta), bounds[1].theta);
	}
	bool ValidateBounds(const Pose2d& s) const
	{
		if (s.x() < bounds[0].x() || s.x() > bounds[1].x()) return false;
		if (s.y() < bounds[0].y() || s.y() > bounds[1].y()) return false;
		if (s.theta < bounds[0].theta || s.theta > bounds[1].theta) return false;
		return true;
	}
	/// state_space_se2.cpp:27-52: x, y, theta drawn in this order from the global engine; the Gaussian sample is clamped to the bounds
	Pose2d SampleUniform() const
	{
		Pose2d s;
		s.x() = Random<double>::SampleUniform(bounds[0].x(), bounds[1].x());
		s.y() = Random<double>::SampleUniform(bounds[0].y(), bounds[1].y());
		s.theta = Random<double>::SampleUniform(bounds[0].theta, bounds[1].theta);
		return s;
	}
	Pose2d SampleGaussian(const Pose2d& mean, const Pose2d& stdDev) const
	{
		Pose2d s;
		s.x() = Random<double>::SampleGaussian(mean.x(), stdDev.x());
		s.y() = Random<double>::SampleGaussian(mean.y(), stdDev.y());
		s.theta = Random<double>::SampleGaussian(mean.theta, stdDev.theta);
		EnforceBounds(s);
		return s;
	}
	const std::array<Pose2d, 2> bounds;
};

/// state_validator/occupancy_map.{h,cpp}: sizes, transforms and the grids the path reads.  The map owns its device map set
/// (pp_map).  The grids come from either side: set by the caller on the host (SetGrids / SetDistances, e.g. copied out of the
/// reference's own GVD) and uploaded, or built on the device (obstacle outlines rasterised by ObstacleListOccupancyMap, fields by
/// GVD::Update, map_authoring.hpp) and fetched back when a host accessor asks.
class OccupancyMap {
public:
	explicit OccupancyMap(float res) : resolution(res) { }
	OccupancyMap(const OccupancyMap&) = delete;
	OccupancyMap& operator=(const OccupancyMap&) = delete;
	virtual ~OccupancyMap()
	{
		if (m_dev)
			pp_map_destroy(m_dev);
	}
	void InitializeSize(float width, float height)
	{
		// occupancy_map.cpp:6-14
		m_localGridOrigin = { -width / 2.0, -height / 2.0 };
		m_worldGridOrigin = m_localOrigin + m_localGridOrigin;
		m_rows = (int)std::ceil(width / resolution);
		m_columns = (int)std::ceil(height / resolution);
		if (!(m_rows > 0 && m_columns > 0))
			throw std::invalid_argument("Invalid grid size: received " + std::to_string(m_rows) + " x " + std::to_string(m_columns)); // utils/grid.h:69-72
		m_width = width;
		m_height = height;
		m_occupancy.assign((size_t)m_rows * m_columns, -1);
		m_dist2.assign((size_t)m_rows * m_columns, INT32_MAX);
		m_distance.assign((size_t)m_rows * m_columns, DistanceOf(INT32_MAX));
		m_pathCost.assign((size_t)m_rows * m_columns, 0.0f);
		m_hostVersion++;
		m_onDevice = false;
		ResetDevice();
	}
	int Rows() const { return m_rows; }
	int Columns() const { return m_columns; }
	/// occupancy_map.h:46 (the reference runs the obstacle brushfire here; the device fields are built by GVD::Update / BuildFields)
	virtual void Update() { }
	void SetPosition(const Point2d& p)
	{
		m_localOrigin = p;
		m_worldGridOrigin = m_localOrigin + m_localGridOrigin;
		ResetDevice();
	}
	const Point2d& GetPosition() const { return m_localOrigin; }
	virtual bool IsOccupied(const GridCellPosition& c) { return HostGrids().m_occupancy[(size_t)c.row * m_columns + c.col] >= 0; }
	int GetOccupancyValue(int row, int col) { return HostGrids().m_occupancy[(size_t)row * m_columns + col]; }
	int GetOccupancyValue(const GridCellPosition& c) { return GetOccupancyValue(c.row, c.col); }
	bool IsInsideMap(const GridCellPosition& c) const { return c.row >= 0 && c.row < m_rows && c.col >= 0 && c.col < m_columns; }
	bool IsInsideMap(const Point2d& p) const { return IsInsideMap(WorldPositionToGridCell(p, false)); }
	GridCellPosition LocalPositionToGridCell(const Point2d& p, bool bounded = true) const
	{
		// occupancy_map.h:106-117
		int row = (int)((p.x() - m_localGridOrigin.x()) / resolution);
		int col = (int)((p.y() - m_localGridOrigin.y()) / resolution);
		if (!bounded || IsInsideMap(GridCellPosition(row, col)))
			return { row, col };
		return { -1, -1 };
	}
	GridCellPosition WorldPositionToGridCell(const Point2d& p, bool bounded = true) const
	{
		// occupancy_map.h:175-183
		int row = (int)((p.x() - m_worldGridOrigin.x()) / resolution);
		int col = (int)((p.y() - m_worldGridOrigin.y()) / resolution);
		if (!bounded || IsInsideMap(GridCellPosition(row, col)))
			return { row, col };
		return { -1, -1 };
	}
	Point2d GridCellToLocalPosition(const GridCellPosition& c) const { return m_localGridOrigin + Point2d(c.row * resolution, c.col * resolution); }
	Point2d GridCellToWorldPosition(const GridCellPosition& c) const { return m_worldGridOrigin + Point2d(c.row * resolution, c.col * resolution); }
	Point2d LocalPositionToWorldPosition(const Point2d& p) const { return p + m_localOrigin; }
	Point2d WorldPositionToLocalPosition(const Point2d& p) const { return p - m_localOrigin; } // occupancy_map.h:185-188
	const Point2d& WorldGridOrigin() const { return m_worldGridOrigin; }
	/// Squared obstacle distance (GVD::ObstacleDistanceMap::m_distance), occupancy ids and GVD::PathCostMap, row-major.
	void SetGrids(const int32_t* occupancy, const int32_t* dist2, const float* pathCost)
	{
		HostGrids();
		const size_t n = (size_t)m_rows * m_columns;
		if (occupancy) m_occupancy.assign(occupancy, occupancy + n);
		if (dist2) {
			m_dist2.assign(dist2, dist2 + n);
			for (size_t i = 0; i < n; i++)
				m_distance[i] = DistanceOf(dist2[i]);
		}
		if (pathCost) m_pathCost.assign(pathCost, pathCost + n);
		m_hostVersion++;
		m_onDevice = false;
	}
	/// The distance grid as the reference's accessor returns it (float metres): ObstacleDistanceMap::GetDistanceToNearestObstacle
	/// for every cell, row-major.  Use this when only the accessor is reachable (m_distance is private in the reference).
	void SetDistances(const float* distance)
	{
		HostGrids();
		m_distance.assign(distance, distance + (size_t)m_rows * m_columns);
		m_hostVersion++;
		m_onDevice = false;
	}
	/// (row, col) per cell of GVD::GetNearestObstacleCell / GetNearestVoronoiEdgeCell, row-major: what the smoother reads
	/// (algo/smoother.cpp:113,123) when the fields were built outside this library
	void SetNearestCells(const int32_t* nearestObstacle, const int32_t* nearestEdge)
	{
		HostGrids();
		const size_t n = (size_t)m_rows * m_columns * 2;
		m_voronoi.nearestObstacle.assign(nearestObstacle, nearestObstacle + n);
		m_voronoi.nearestEdge.assign(nearestEdge, nearestEdge + n);
		m_nearestOnHost = true;
		m_hostVersion++;
		m_onDevice = false;
	}
	/// the post-processing of HybridAStar::SearchPath can run: the two label grids exist (built here or handed over)
	bool HasNearestCells() const { return (m_onDevice && m_fieldsBuilt) || (!m_onDevice && m_nearestOnHost); }
	/// gvd.h:38: `std::sqrt(m_distance[row][col]) * resolution` -- sqrt of the int in double, product in double, returned as float
	float DistanceOf(int32_t d2) const { return (float)(std::sqrt((double)d2) * (double)resolution); }
	float GetDistanceToNearestObstacle(int row, int col) { return HostGrids().m_distance[(size_t)row * m_columns + col]; }
	float GetPathCost(int row, int col) { return HostGrids().m_pathCost[(size_t)row * m_columns + col]; }
	const std::vector<float>& Distance() { return HostGrids().m_distance; }
	const std::vector<int32_t>& Occupancy() { return HostGrids().m_occupancy; }
	const std::vector<int32_t>& Dist2() { return HostGrids().m_dist2; }
	const std::vector<float>& PathCost() { return HostGrids().m_pathCost; }

	/// StateSpaceSE2 bounds the validator checks poses against (state_validator_occupancy_map.cpp:18-20); without a validator
	/// the map's own extent
	void SetStateBounds(const std::array<Pose2d, 2>& b)
	{
		m_bounds = b;
		m_hasBounds = true;
		ResetDevice();
	}
	/// the device map set; host-side grids are pushed when they are the newer side
	pp_map* Device()
	{
		if (m_rows <= 0)
			throw std::runtime_error("The size of the occupancy matrix has not been initialized"); // obstacle_list_occupancy_map.cpp:31-32
		if (!m_dev) {
			pp_map_desc d {};
			d.rows = m_rows;
			d.cols = m_columns;
			d.resolution = resolution;
			d.grid_origin[0] = m_worldGridOrigin.x();
			d.grid_origin[1] = m_worldGridOrigin.y();
			d.local_origin[0] = m_localOrigin.x();
			d.local_origin[1] = m_localOrigin.y();
			if (m_hasBounds) {
				d.lower[0] = m_bounds[0].x(), d.lower[1] = m_bounds[0].y(), d.lower[2] = m_bounds[0].theta;
				d.upper[0] = m_bounds[1].x(), d.upper[1] = m_bounds[1].y(), d.upper[2] = m_bounds[1].theta;
			} else {
				d.lower[0] = m_localGridOrigin.x(), d.lower[1] = m_localGridOrigin.y(), d.lower[2] = -M_PI;
				d.upper[0] = m_localGridOrigin.x() + m_width, d.upper[1] = m_localGridOrigin.y() + m_height, d.upper[2] = M_PI;
			}
			ppCheck(pp_map_create(HipContext::Get(), &d, &m_dev));
			m_uploaded = ~0ull;
			if (m_onDevice) { // the device grids went with the old handle: fall back to the host copies fetched before the reset
				m_onDevice = false;
				m_hostVersion++;
			}
		}
		if (!m_onDevice && m_uploaded != m_hostVersion) {
			ppCheck(pp_map_upload_distance(m_dev, m_distance.data()));
			ppCheck(pp_map_upload_occupancy(m_dev, m_occupancy.data()));
			ppCheck(pp_map_upload_path_cost(m_dev, m_pathCost.data()));
			if (m_nearestOnHost)
				ppCheck(pp_map_upload_nearest_cells(m_dev, m_voronoi.nearestObstacle.data(), m_voronoi.nearestEdge.data()));
			m_uploaded = m_hostVersion;
		}
		return m_dev;
	}
	/// How GVD::Update builds its two distance maps (pp_map_update_gvd_ex).  ReferenceOrder (the default of this drop-in): the
	/// reference's own brushfire, its grids bit for bit, incremental after the first build.  ExactTransform: the exact Euclidean
	/// transform on the device, milliseconds instead of seconds, NOT the reference's bits on a few tie cells in ten thousand --
	/// an opt-in for callers that rebuild large maps often.
	enum class FieldUpdateMode { ReferenceOrder = PP_GVD_REFERENCE_ORDER, ExactTransform = PP_GVD_EXACT_EDT };
	void SetFieldUpdateMode(FieldUpdateMode mode) { m_fieldMode = mode; }
	FieldUpdateMode GetFieldUpdateMode() const { return m_fieldMode; }
	/// GVD::Update (gvd.cpp:294-301) from the device occupancy grid
	void BuildFields(float alpha, float dMax)
	{
		pp_map* d = Device(); // (pushes a host-set occupancy first)
		ppCheck(pp_map_update_gvd_ex(d, alpha, dMax, (int32_t)m_fieldMode, nullptr));
		m_onDevice = true;
		m_hostStale = true;
		m_fieldsBuilt = true;
	}
	bool FieldsBuilt() const { return m_fieldsBuilt; }
	/// obstacles were written on the device since the fields were last built (HybridAStar::SearchPath calls GVD::Update, hybrid_a_star.cpp:250)
	bool FieldsOutdated() const { return m_onDevice && !m_fieldsBuilt; }
	/// ObstacleListOccupancyMap: boundary cells get `value` on the device
	void SetCellsOnDevice(const std::vector<GridCellPosition>& cells, int32_t value)
	{
		pp_map* d = Device();
		static_assert(sizeof(GridCellPosition) == 8, "GridCellPosition is two ints");
		ppCheck(pp_map_set_cells(d, (int64_t)cells.size(), cells.empty() ? nullptr : &cells[0].row, value));
		m_onDevice = true;
		m_hostStale = true;
		m_fieldsBuilt = false;
	}
	/// Voronoi data of the last BuildFields (host copies), row-major
	struct VoronoiGrids {
		std::vector<int32_t> d2, nearestEdge, nearestObstacle;
		std::vector<uint8_t> edge;
	};
	const VoronoiGrids& Voronoi()
	{
		HostGrids();
		return m_voronoi;
	}
	const float resolution;

protected:
	/// host copies, fetched from the device when it holds the newer grids
	OccupancyMap& HostGrids()
	{
		if (m_onDevice && m_hostStale && m_dev) {
			const size_t n = (size_t)m_rows * m_columns;
			ppCheck(pp_map_download_occupancy(m_dev, m_occupancy.data()));
			if (m_fieldsBuilt) {
				m_voronoi.d2.resize(n);
				m_voronoi.edge.resize(n);
				m_voronoi.nearestEdge.resize(2 * n);
				m_voronoi.nearestObstacle.resize(2 * n);
				ppCheck(pp_map_download_gvd(m_dev, m_dist2.data(), m_voronoi.nearestObstacle.data(), m_voronoi.edge.data(), m_voronoi.d2.data(), m_voronoi.nearestEdge.data(),
					m_pathCost.data()));
				for (size_t i = 0; i < n; i++)
					m_distance[i] = DistanceOf(m_dist2[i]);
				m_nearestOnHost = true;
			}
			m_hostStale = false;
		}
		return *this;
	}
	void ResetDevice()
	{
		if (m_dev) {
			HostGrids();
			pp_map_destroy(m_dev);
			m_dev = nullptr;
		}
	}
	int m_rows = -1, m_columns = -1;
	float m_width = 0, m_height = 0;
	Point2d m_localOrigin, m_localGridOrigin, m_worldGridOrigin;
	std::array<Pose2d, 2> m_bounds;
	bool m_hasBounds = false;
	std::vector<int32_t> m_occupancy, m_dist2;
	std::vector<float> m_pathCost, m_distance;
	VoronoiGrids m_voronoi;
	pp_map* m_dev = nullptr;
	uint64_t m_hostVersion = 0, m_uploaded = ~0ull;
	bool m_onDevice = false, m_hostStale = false, m_fieldsBuilt = false, m_nearestOnHost = false;
	FieldUpdateMode m_fieldMode = FieldUpdateMode::ReferenceOrder;
};

/// state_validator/state_validator.h:10-43 (SE2 instantiation)
class StateValidatorSE2Base {
public:
	explicit StateValidatorSE2Base(const Ref<StateSpaceSE2>& s) : m_stateSpace(s) { }
	virtual ~StateValidatorSE2Base() = default;
	virtual bool IsStateValid(const Pose2d& state) = 0;
	/// `last`: ratio of the last valid sample along the path
	virtual bool IsPathValid(const PathSE2Base& path, float* last = nullptr) = 0;
	/// state_validator.h:30-37 as it was meant (the reference passes the float by value there, Appendix A Q18)
	bool IsPathValid(const PathSE2Base& path, Pose2d* last)
	{
		float ratio = 0.0f;
		const bool ok = IsPathValid(path, &ratio);
		if (last)
			*last = path.Interpolate(ratio);
		return ok;
	}
	Ref<StateSpaceSE2>& GetStateSpace() { return m_stateSpace; }
protected:
	Ref<StateSpaceSE2> m_stateSpace;
};

/// state_validator/state_validator_free.h:9-31: bounds only, every path valid
class StateValidatorSE2Free : public StateValidatorSE2Base {
public:
	explicit StateValidatorSE2Free(const Ref<StateSpaceSE2>& s) : StateValidatorSE2Base(s) { }
	bool IsStateValid(const Pose2d& state) override { return m_stateSpace->ValidateBounds(state); }
	bool IsPathValid(const PathSE2Base& /*path*/, float* last = nullptr) override
	{
		if (last)
			*last = 1.0;
		return true;
	}
};

/// state_validator/state_validator_occupancy_map.{h,cpp}, GPU-backed.
class StateValidatorOccupancyMap : public StateValidatorSE2Base {
public:
	StateValidatorOccupancyMap(const Ref<StateSpaceSE2>& stateSpace, const Ref<OccupancyMap>& map) : StateValidatorSE2Base(stateSpace), m_map(map)
	{
		// state_validator_occupancy_map.cpp:6-13: the map is sized from the state-space bounds (float)
		float width = stateSpace->bounds[1].x() - stateSpace->bounds[0].x();
		float height = stateSpace->bounds[1].y() - stateSpace->bounds[0].y();
		m_map->InitializeSize(width, height);
		m_map->SetStateBounds(stateSpace->bounds);
	}
	bool IsStateValid(const Pose2d& state) override
	{
		uint8_t v = 0;
		ppCheck(pp_check_states(Device(), 1, &state.position.v[0], &v));
		return v != 0;
	}
	/// batched: n poses -> n flags
	std::vector<uint8_t> IsStateValid(const std::vector<Pose2d>& states)
	{
		std::vector<uint8_t> out(states.size());
		if (!states.empty())
			ppCheck(pp_check_states(Device(), (int64_t)states.size(), &states[0].position.v[0], out.data()));
		return out;
	}
	/// IsPathValid over constant-steer arcs given as (start, curvature, length, direction)
	bool IsArcValid(const Pose2d& from, double curvature, double length, Direction dir, float* last = nullptr)
	{
		uint8_t v = 0;
		float l = 0;
		int32_t d = dir == Direction::Backward ? 1 : 0;
		ppCheck(pp_check_arcs(Device(), 1, &from.position.v[0], &curvature, &length, &d, &v, &l));
		if (last)
			*last = l;
		return v != 0;
	}
	/// state_validator_occupancy_map.cpp:28-71.  The reference's path types are validated on the GPU, whole march in one kernel:
	/// PathConstantSteer -> pp_check_arcs, PathReedsShepp -> pp_check_rs_paths, PathSE2 -> pp_check_se2_paths.  A path type
	/// defined by the caller (a C++ or Python subclass of Path) can only be sampled through its own virtual Interpolate on the
	/// host, so for those the march runs here, sample by sample, over the host copy of the distance grid.
	bool IsPathValid(const PathSE2Base& path, float* last = nullptr) override
	{
		uint8_t v = 0;
		float l = 0.0f;
		if (auto* arc = dynamic_cast<const PathConstantSteer*>(&path)) {
			if (arc->GetModel()->RearToCenter() == 0.0) { // the device arc is the rear-axle model the planner uses (hybrid_a_star.cpp:19)
				const double kappa = arc->GetModel()->Curvature(arc->GetSteeringAngle());
				const double len = arc->GetLength();
				const int32_t d = arc->GetDirection(0.0) == Direction::Backward ? 1 : 0;
				ppCheck(pp_check_arcs(Device(), 1, &arc->GetInitialState().position.v[0], &kappa, &len, &d, &v, &l));
				if (last)
					*last = l;
				return v != 0;
			}
		} else if (auto* rs = dynamic_cast<const PathReedsShepp*>(&path)) {
			ppCheck(pp_check_rs_paths(Device(), 1, &rs->Record(), &v, &l));
			if (last)
				*last = l;
			return v != 0;
		} else if (auto* line = dynamic_cast<const PathSE2*>(&path)) {
			ppCheck(pp_check_se2_paths(Device(), 1, &line->GetInitialState().position.v[0], &line->GetFinalState().position.v[0], &v, &l));
			if (last)
				*last = l;
			return v != 0;
		}
		return MarchOnHost(path, last);
	}
	/// IsPathValid over many paths of one of the reference's types at once (one launch per type); `last` may be null
	std::vector<uint8_t> IsPathValid(const std::vector<Ref<PathSE2Base>>& paths, std::vector<float>* last = nullptr)
	{
		std::vector<uint8_t> out(paths.size());
		if (last)
			last->assign(paths.size(), 0.0f);
		std::vector<pp_rs_path> recs;
		std::vector<size_t> recIdx;
		for (size_t i = 0; i < paths.size(); i++) {
			if (auto* rs = dynamic_cast<const PathReedsShepp*>(paths[i].get())) {
				recs.push_back(rs->Record());
				recIdx.push_back(i);
			} else {
				float l = 0.0f;
				out[i] = IsPathValid(*paths[i], &l) ? 1 : 0;
				if (last)
					(*last)[i] = l;
			}
		}
		if (!recs.empty()) {
			std::vector<uint8_t> v(recs.size());
			std::vector<float> l(recs.size());
			ppCheck(pp_check_rs_paths(Device(), (int64_t)recs.size(), recs.data(), v.data(), l.data()));
			for (size_t k = 0; k < recs.size(); k++) {
				out[recIdx[k]] = v[k];
				if (last)
					(*last)[recIdx[k]] = l[k];
			}
		}
		return out;
	}
	Ref<OccupancyMap>& GetOccupancyMap() { return m_map; }
	/// the map's device map set with this validator's tunables pushed
	pp_map* Device()
	{
		pp_map* d = m_map->Device();
		ppCheck(pp_map_set_validator(d, minSafeRadius, minPathInterpolationDistance));
		return d;
	}
	float minPathInterpolationDistance = 0.1f; // state_validator_occupancy_map.h:27-28
	float minSafeRadius = 1.0f;

private:
	/// the reference's loop, verbatim in structure, for caller-defined path types (see IsPathValid)
	bool MarchOnHost(const PathSE2Base& path, float* last)
	{
		auto stateValid = [&](const Pose2d& state, float& distance) {
			const Pose2d local(m_map->WorldPositionToLocalPosition(state.position), state.theta);
			const GridCellPosition cell = m_map->WorldPositionToGridCell(state.position);
			if (!m_stateSpace->ValidateBounds(local) || !m_map->IsInsideMap(cell))
				return false;
			distance = m_map->GetDistanceToNearestObstacle(cell.row, cell.col);
			return distance >= minSafeRadius;
		};
		const auto& bounds = m_stateSpace->bounds;
		const double pathLength = path.GetLength();
		float distance = 0.0f;
		if (pathLength == 0.0) {
			if (last)
				*last = 1.0f;
			return stateValid(path.GetInitialState(), distance);
		}
		double lastValidLength = 0.0, length = 0.0;
		while (length < pathLength) {
			const Pose2d state = path.Interpolate(length / pathLength);
			if (!stateValid(state, distance)) {
				if (last)
					*last = lastValidLength / pathLength;
				return false;
			}
			lastValidLength = length;
			const float distToMapBorder = std::min({ state.x() - bounds[0].x(), bounds[1].x() - state.x(), state.y() - bounds[0].y(), bounds[1].y() - state.y() });
			float deltaLength = distance - minSafeRadius;
			deltaLength = std::min(deltaLength, distToMapBorder);
			length += std::max(deltaLength, minPathInterpolationDistance);
		}
		if (last)
			*last = 1.0f;
		return true;
	}
	Ref<OccupancyMap> m_map;
};

/// algo/smoother.h:18-60: the status codes and parameters of the path smoother (the descent itself runs in pp_planner_postprocess)
struct Smoother {
	enum Status { MaxIteration = 0, StepTolerance, PathSize, Failure = -1, Collision = -2 };
	struct Parameters {
		float stepTolerance = 1e-3;
		int maxIterations = 2000;
		float learningRate = 0.01f;
		float pathWeight = 0.0f;
		float smoothWeight = 0.4f;
		float voronoiWeight = 0.02f;
		float collisionWeight = 0.2f;
		float curvatureWeight = 0.4f;
		float collisionRatio = 0.2f;
		float maxCurvature;
		explicit Parameters(float maxCurvature) : maxCurvature(maxCurvature) { }
	};
};

/// algo/hybrid_a_star.h:27-264 -- graph search and post-processing on the GPU.
class HybridAStar : public PathPlannerSE2Base {
public:
	struct SearchParameters { // algo/hybrid_a_star.h:29-50
		const double wheelbase = 2.6;
		const double minTurningRadius = 2.0;
		const double directionSwitchingCost = 0.0;
		const double reverseCostMultiplier = 1.0;
		const double forwardCostMultiplier = 1.0;
		const double voronoiCostMultiplier = 1.0;
		const unsigned int numGeneratedMotion = 5;
		const double spatialResolution = 1.0;
		const double angularResolution = 0.0872;
		SearchParameters() = default;
		SearchParameters(double minTurningRadius, double directionSwitchingCost, double reverseCostMultiplier, double forwardCostMultiplier,
			double voronoiCostMultiplier, unsigned int numGeneratedMotion, double spatialResolution, double angularResolution) :
			minTurningRadius(minTurningRadius), directionSwitchingCost(directionSwitchingCost), reverseCostMultiplier(reverseCostMultiplier),
			forwardCostMultiplier(forwardCostMultiplier), voronoiCostMultiplier(voronoiCostMultiplier), numGeneratedMotion(numGeneratedMotion),
			spatialResolution(spatialResolution), angularResolution(angularResolution) { }
	};
	struct Stats { // algo/hybrid_a_star.h:52-55
		Status graphSearchStatus = Status::Failure;
		Smoother::Status smoothingStatus = Smoother::Status::Failure;
	};

	HybridAStar() : HybridAStar(SearchParameters()) { }
	explicit HybridAStar(const SearchParameters& p, int maxBatch = 1, int maxNodes = 81920) :
		m_param(p), m_maxBatch(maxBatch), m_maxNodes(maxNodes), m_smootherParam((float)(1.0 / p.minTurningRadius)) { } // hybrid_a_star.cpp:214
	~HybridAStar() override
	{
		if (m_planner)
			pp_planner_destroy(m_planner);
	}
	/// hybrid_a_star.cpp:206-235: builds the non-holonomic table (on the device) and the per-query workspaces
	bool Initialize(const Ref<StateValidatorOccupancyMap>& validator)
	{
		if (!validator || !validator->GetStateSpace())
			return isInitialized = false;
		m_validator = validator;
		if (validator->GetOccupancyMap()->FieldsOutdated())
			validator->GetOccupancyMap()->BuildFields(20.0f, 30.0f); // the GVD the reference builds here (hybrid_a_star.cpp:212)
		pp_hybrid_params hp { m_param.wheelbase, m_param.minTurningRadius, m_param.directionSwitchingCost, m_param.reverseCostMultiplier,
			m_param.forwardCostMultiplier, m_param.voronoiCostMultiplier, m_param.numGeneratedMotion, m_param.spatialResolution, m_param.angularResolution, 1, 1 };
		if (m_planner) {
			pp_planner_destroy(m_planner);
			m_planner = nullptr;
		}
		if (pp_planner_create(validator->Device(), &hp, m_maxBatch, m_maxNodes, &m_planner))
			return isInitialized = false;
		if (pp_planner_set_nonholo_table(m_planner, nullptr))
			return isInitialized = false;
		return isInitialized = true;
	}
	Status SearchPath() override
	{
		if (!isInitialized)
			return Status::Failure; // "The algorithm has not been initialized successfully." (hybrid_a_star.cpp:243-246)
		if (m_validator->GetOccupancyMap()->FieldsOutdated())
			m_validator->GetOccupancyMap()->BuildFields(20.0f, 30.0f); // m_gvd->Update(), hybrid_a_star.cpp:250 (GVD::alpha / dMax, gvd.h:181)
		m_validator->Device(); // pushes map edits / tunables
		pp_query_result r {};
		uint64_t seed = m_seed;
		if (pp_planner_search_batch(m_planner, 1, &m_init.position.v[0], &m_goal.position.v[0], &seed, &r))
			return m_stats.graphSearchStatus = Status::Failure;
		m_last = r;
		m_path.clear();
		m_smoothed.clear();
		m_stats.smoothingStatus = Smoother::Status::Failure;
		m_stats.graphSearchStatus = (r.status == 0 ? Status::Success : Status::Failure);
		if (m_stats.graphSearchStatus < 0)
			return m_stats.graphSearchStatus;
		// hybrid_a_star.cpp:260-303: sample the composite path, smooth it; the smoothed path when that worked, else the sampled one.
		// Needs the GVD's two nearest-cell grids (built by GVD::Update here, or handed over with OccupancyMap::SetNearestCells);
		// a map that only carries the three grids of the search keeps the graph-search nodes as its path.
		m_path = GetGraphSearchNodes();
		if (m_validator->GetOccupancyMap()->HasNearestCells() && r.n_path >= 2) {
			const pp_smoother_params sp { m_smootherParam.stepTolerance, m_smootherParam.maxIterations, m_smootherParam.learningRate, m_smootherParam.pathWeight,
				m_smootherParam.smoothWeight, m_smootherParam.voronoiWeight, m_smootherParam.collisionWeight, m_smootherParam.curvatureWeight, m_smootherParam.collisionRatio,
				m_smootherParam.maxCurvature };
			pp_post_result post {};
			const int postRc = pp_planner_postprocess(m_planner, 1, pathInterpolation, &sp, 2048, &post);
			m_postOverflow = postRc == 0 && post.smoothing_status == -4;
			if (m_postOverflow) // more samples than the device post-processing holds (2048): said, not hidden -- GetPath() keeps the graph-search nodes
				std::fprintf(stderr, "[pathplanning_amd] HybridAStar::SearchPath: the path (%.1f m at %.3f m spacing) has more than 2048 samples; it was neither "
					"sampled nor smoothed, GetPath() returns the graph-search nodes (raise pathInterpolation)\n", post.length, (double)pathInterpolation);
			if (postRc == 0 && post.n_points > 0) {
				std::vector<Pose2d> sampled((size_t)post.n_points), smoothed((size_t)post.n_points);
				ppCheck(pp_planner_get_processed_path(m_planner, 0, &sampled[0].position.v[0], nullptr, &smoothed[0].position.v[0]));
				m_stats.smoothingStatus = (Smoother::Status)post.smoothing_status;
				m_smoothed = smoothed;
				m_path = post.smoothing_status >= 0 ? smoothed : sampled;
			}
		}
		return Status::Success;
	}
	/// algo/hybrid_a_star.h:226: the sampled, and when the smoother succeeds smoothed, path (see SearchPath)
	std::vector<Pose2d> GetPath() const override { return m_path; }
	/// the last SearchPath's path had more samples than the device post-processing holds: GetPath() is the graph-search nodes
	bool PostProcessingOverflowed() const { return m_postOverflow; }
	/// algo/hybrid_a_star.h:237: what the smoother ended with (also when it failed)
	const std::vector<Pose2d>& GetSmoothedPath() const { return m_smoothed; }
	const Smoother::Parameters& GetSmootherParameters() const { return m_smootherParam; }
	void SetSmootherParameters(const Smoother::Parameters& p) { m_smootherParam = p; }
	/// nodes of the graph-search solution, root .. goal (the end points of GetGraphSearchPath()'s edges)
	std::vector<Pose2d> GetGraphSearchNodes() const
	{
		std::vector<Pose2d> out((size_t)m_last.n_path);
		if (m_last.n_path > 0)
			ppCheck(pp_planner_get_path(m_planner, 0, &out[0].position.v[0], nullptr, nullptr, nullptr, nullptr));
		return out;
	}
	/// algo/hybrid_a_star.h:231: the solution as path objects, one per edge of the search tree (constant-steer arcs, then
	/// possibly the Reeds-Shepp connection to the goal); empty when the search failed
	std::vector<Ref<PathNonHolonomicSE2Base>> GetGraphSearchPath() const
	{
		std::vector<Ref<PathNonHolonomicSE2Base>> out;
		const int n = m_last.n_path;
		if (n < 2)
			return out;
		std::vector<Pose2d> poses((size_t)n);
		std::vector<int32_t> kind((size_t)n), prim((size_t)n);
		std::vector<double> length((size_t)n);
		ppCheck(pp_planner_get_path(m_planner, 0, &poses[0].position.v[0], kind.data(), prim.data(), length.data(), nullptr));
		auto model = makeRef<KinematicBicycleModel>(m_param.wheelbase, 0.0); // hybrid_a_star.cpp:19
		// steering of primitive p = 2 * deltaIndex + direction, deltas {0, +d1, -d1, ...} (hybrid_a_star.cpp:21-28, 65-77)
		const double deltaMax = model->GetSteeringAngleFromTurningRadius(m_param.minTurningRadius);
		for (int i = 1; i < n; i++) {
			if (kind[i] == 1) {
				const int di = prim[i] / 2;
				const double delta = di == 0 ? 0.0 : ((di + 1) / 2) / 2.0 * deltaMax * ((di & 1) ? 1.0 : -1.0);
				out.push_back(makeRef<PathConstantSteer>(model, poses[i - 1], delta, length[i], (prim[i] & 1) ? Direction::Backward : Direction::Forward));
			} else { // the analytic expansion: GetOptimalPath(parent, goal) with the planner's costs (hybrid_a_star.cpp:155-157)
				pp_rs_path rec;
				ppCheck(pp_rs_connect(HipContext::Get(), 1, &poses[i - 1].position.v[0], &m_goal.position.v[0], m_param.minTurningRadius, (float)m_param.reverseCostMultiplier,
					(float)m_param.forwardCostMultiplier, (float)m_param.directionSwitchingCost, &rec));
				out.push_back(makeRef<PathReedsShepp>(rec));
			}
		}
		return out;
	}
	/// algo/hybrid_a_star.h:229: every edge of the search tree as a path object (also dead leaves, as the reference keeps them);
	/// single-query planners only (the throughput kernel keeps node records per row, not per query)
	std::vector<Ref<PathNonHolonomicSE2Base>> GetGraphSearchExploredPathSet() const
	{
		std::vector<Ref<PathNonHolonomicSE2Base>> out;
		const int n = m_last.n_nodes;
		if (n < 2 || pp_planner_search_rows(m_planner) != 0)
			return out;
		std::vector<int32_t> parents((size_t)n), actions((size_t)n);
		std::vector<Pose2d> poses((size_t)n);
		std::vector<double> lengths((size_t)n);
		ppCheck(pp_planner_debug_nodes(m_planner, 0, n, parents.data(), &poses[0].position.v[0], nullptr, nullptr));
		ppCheck(pp_planner_debug_node_actions(m_planner, 0, n, actions.data(), lengths.data()));
		auto model = makeRef<KinematicBicycleModel>(m_param.wheelbase, 0.0);
		const double deltaMax = model->GetSteeringAngleFromTurningRadius(m_param.minTurningRadius);
		for (int i = 1; i < n; i++) {
			if (parents[i] < 0)
				continue;
			const Pose2d& from = poses[(size_t)parents[i]];
			if (actions[i] >= 1000) {
				pp_rs_path rec;
				ppCheck(pp_rs_connect(HipContext::Get(), 1, &from.position.v[0], &m_goal.position.v[0], m_param.minTurningRadius, (float)m_param.reverseCostMultiplier,
					(float)m_param.forwardCostMultiplier, (float)m_param.directionSwitchingCost, &rec));
				out.push_back(makeRef<PathReedsShepp>(rec));
			} else if (actions[i] >= 0) {
				const int di = actions[i] / 2;
				const double delta = di == 0 ? 0.0 : ((di + 1) / 2) / 2.0 * deltaMax * ((di & 1) ? 1.0 : -1.0);
				out.push_back(makeRef<PathConstantSteer>(model, from, delta, lengths[i], (actions[i] & 1) ? Direction::Backward : Direction::Forward));
			}
		}
		return out;
	}
	/// algo/hybrid_a_star.h:247 / heuristics.cpp:167-205: PPM image of the obstacle heuristic of the current goal
	/// (black = unexplored, brighter = closer to the goal)
	void VisualizeObstacleHeuristic(const std::string& filename) const
	{
		OccupancyMap& map = *m_validator->GetOccupancyMap();
		const int rows = map.Rows(), cols = map.Columns();
		std::vector<float> cost((size_t)rows * cols);
		const double goal[2] = { m_goal.x(), m_goal.y() };
		ppCheck(pp_obstacle_heuristic(m_validator->Device(), 1, goal, cost.data()));
		FILE* F = std::fopen(filename.c_str(), "w");
		if (!F)
			return;
		float maxCost = -INFINITY;
		for (float c : cost)
			if (c != INFINITY)
				maxCost = std::max(maxCost, c);
		std::fprintf(F, "P6\n#\n%d %d\n255\n", rows, cols);
		for (int y = cols - 1; y >= 0; y--)
			for (int x = 0; x < rows; x++) {
				const float c = cost[(size_t)x * cols + y];
				unsigned char v = 0;
				if (c != INFINITY) // m_explored
					v = (unsigned char)std::max(0.0f, std::min((maxCost - c) / maxCost * 255, 255.0f));
				const unsigned char rgb[3] = { v, v, v };
				std::fwrite(rgb, 1, 3, F);
			}
		std::fclose(F);
	}
	double GetGraphSearchOptimalCost() const { return m_last.status == 0 ? m_last.cost : INFINITY; }
	const Stats& GetStats() const { return m_stats; }
	const SearchParameters& GetSearchParameters() const { return m_param; }
	Ref<StateValidatorOccupancyMap>& GetStateValidator() { return m_validator; }
	/// the reference's process-global RNG becomes one stream per query
	void SetSeed(uint64_t seed) { m_seed = seed; }
	/// batch of independent queries (n <= maxBatch)
	std::vector<pp_query_result> SearchBatch(const std::vector<Pose2d>& starts, const std::vector<Pose2d>& goals, const std::vector<uint64_t>& seeds)
	{
		if (!isInitialized || starts.size() != goals.size() || starts.size() != seeds.size())
			throw std::invalid_argument("HybridAStar::SearchBatch: not initialised or size mismatch");
		if (m_validator->GetOccupancyMap()->FieldsOutdated())
			m_validator->GetOccupancyMap()->BuildFields(20.0f, 30.0f);
		m_validator->Device();
		std::vector<pp_query_result> res(starts.size());
		if (!starts.empty())
			ppCheck(pp_planner_search_batch(m_planner, (int32_t)starts.size(), &starts[0].position.v[0], &goals[0].position.v[0], seeds.data(), res.data()));
		return res;
	}
	std::vector<Pose2d> GetPathOf(int q, int nPath) const
	{
		std::vector<Pose2d> out((size_t)nPath);
		if (nPath > 0)
			ppCheck(pp_planner_get_path(m_planner, q, &out[0].position.v[0], nullptr, nullptr, nullptr, nullptr));
		return out;
	}
	float pathInterpolation = 0.1f; // algo/hybrid_a_star.h:249: spacing of the sampled path

private:
	SearchParameters m_param;
	int m_maxBatch, m_maxNodes;
	bool isInitialized = false;
	Ref<StateValidatorOccupancyMap> m_validator;
	pp_planner* m_planner = nullptr;
	pp_query_result m_last {};
	Smoother::Parameters m_smootherParam;
	std::vector<Pose2d> m_path, m_smoothed;
	bool m_postOverflow = false;
	Stats m_stats;
	uint64_t m_seed = 0;
};

/// algo/rrt.h:12-21 / algo/rrt_star.h:12-21
struct RRTParameters {
	unsigned int maxIteration = 100;
	unsigned int maxNumberTreeNode = 1e4;
	double maxConnectionDistance = 0.1;
	double goalBias = 0.05;
};
struct RRTStarParameters {
	unsigned int maxIteration = 1e4;
	unsigned int maxNumberTreeNode = 1e4;
	double maxConnectionDistance = 0.1;
	double goalBias = 0.05;
	// beyond the reference (its RRT* has no rewire step, rrt_star.h:83): re-parent near nodes through the new node; with
	// radiusGamma > 0 the near-set is the <= 16 nearest nodes within radiusGamma * sqrt(ln(n + 1) / (n + 1))
	bool rewire = false;
	double radiusGamma = 0.0;
};

/// HybridAStar::SearchPath's search stage for a STREAM of queries (include/pp_hip.h: pp_pipeline_*): queries are submitted as they
/// come -- up to `capacity` in flight -- and results polled in completion order; per query they are what HybridAStar::SearchPath
/// finds (same device code).  The scheduling that keeps the GPU full lives in the library, not in the caller.
class HybridAStarPipeline {
public:
	struct Result {
		uint64_t ticket = 0;
		Status status = Status::Failure;
		double cost = 0.0;
		int numExpanded = 0, numPathNodes = 0, latticeBoundaryHits = 0;
	};
	explicit HybridAStarPipeline(const HybridAStar::SearchParameters& p, int capacity = 24576, int maxNodes = 81920, int searchRows = 0) :
		m_param(p), m_capacity(capacity), m_maxNodes(maxNodes), m_searchRows(searchRows) { }
	HybridAStarPipeline(const HybridAStarPipeline&) = delete;
	HybridAStarPipeline& operator=(const HybridAStarPipeline&) = delete;
	~HybridAStarPipeline()
	{
		if (m_pipe)
			pp_pipeline_destroy(m_pipe);
	}
	bool Initialize(const Ref<StateValidatorOccupancyMap>& validator)
	{
		if (!validator || !validator->GetStateSpace())
			return false;
		m_validator = validator;
		if (validator->GetOccupancyMap()->FieldsOutdated())
			validator->GetOccupancyMap()->BuildFields(20.0f, 30.0f);
		pp_hybrid_params hp { m_param.wheelbase, m_param.minTurningRadius, m_param.directionSwitchingCost, m_param.reverseCostMultiplier,
			m_param.forwardCostMultiplier, m_param.voronoiCostMultiplier, m_param.numGeneratedMotion, m_param.spatialResolution, m_param.angularResolution, 1, 1 };
		if (m_pipe) {
			pp_pipeline_destroy(m_pipe);
			m_pipe = nullptr;
		}
		if (pp_pipeline_create(validator->Device(), &hp, m_capacity, m_maxNodes, m_searchRows, 0, &m_pipe))
			return false;
		return pp_planner_set_nonholo_table(pp_pipeline_planner(m_pipe), nullptr) == 0;
	}
	/// takes a prefix of the queries (as many as there are free slots) and returns how many; `tickets` (optional) gets their ids
	int Submit(const std::vector<Pose2d>& starts, const std::vector<Pose2d>& goals, const std::vector<uint64_t>& seeds, std::vector<uint64_t>* tickets = nullptr)
	{
		static_assert(sizeof(Pose2d) == 24, "Pose2d is three contiguous doubles");
		const int n = (int)std::min(starts.size(), std::min(goals.size(), seeds.size()));
		if (!m_pipe || n == 0)
			return 0;
		m_validator->Device(); // pushes map edits / tunables
		std::vector<uint64_t> t((size_t)n);
		int32_t taken = 0;
		ppCheck(pp_pipeline_submit(m_pipe, n, &starts[0].position.v[0], &goals[0].position.v[0], seeds.data(), t.data(), &taken));
		if (tickets)
			tickets->assign(t.begin(), t.begin() + taken);
		return taken;
	}
	/// completed queries so far (never blocks); their slots are free again
	int Poll(std::vector<Result>& out, int maxResults = 4096)
	{
		out.clear();
		if (!m_pipe)
			return 0;
		std::vector<uint64_t> t((size_t)maxResults);
		std::vector<pp_query_result> r((size_t)maxResults);
		int32_t n = 0;
		ppCheck(pp_pipeline_poll(m_pipe, maxResults, t.data(), r.data(), 1, &n));
		for (int i = 0; i < n; i++) {
			Result x;
			x.ticket = t[(size_t)i];
			x.status = r[(size_t)i].status == 0 ? Status::Success : Status::Failure;
			x.cost = r[(size_t)i].cost;
			x.numExpanded = r[(size_t)i].n_expanded;
			x.numPathNodes = r[(size_t)i].n_path;
			x.latticeBoundaryHits = r[(size_t)i].n_lattice_boundary_hits;
			out.push_back(x);
		}
		return n;
	}
	int InFlight() const { return m_pipe ? pp_pipeline_in_flight(m_pipe) : 0; }
	int FreeSlots() const { return m_pipe ? pp_pipeline_free_slots(m_pipe) : 0; }

private:
	HybridAStar::SearchParameters m_param;
	int m_capacity, m_maxNodes, m_searchRows;
	Ref<StateValidatorOccupancyMap> m_validator;
	pp_pipeline* m_pipe = nullptr;
};

/// RRT<Point2d, 2> / RRTStar<Point2d, 2> with PathConnectionR2; validator == nullptr is StateValidatorFree.
template <bool kStar, typename Params>
class RRTR2Hip : public PathPlannerR2Base {
public:
	RRTR2Hip(const Point2d& lower, const Point2d& upper, const Ref<StateValidatorOccupancyMap>& validator = nullptr) : m_lb(lower), m_ub(upper), m_validator(validator) { }
	Params GetParameters() const { return m_parameters; }
	void SetParameters(const Params& p) { m_parameters = p; }
	void SetSeed(uint64_t s) { m_seed = s; }
	Status SearchPath() override
	{
		double params[5] = { (double)m_parameters.maxIteration, (double)m_parameters.maxNumberTreeNode, m_parameters.maxConnectionDistance, m_parameters.goalBias, 0.0 };
		int32_t mode = kStar ? 1 : 0;
		if constexpr (kStar) {
			if (m_parameters.radiusGamma > 0.0) {
				mode = 3;
				params[4] = m_parameters.radiusGamma;
			} else if (m_parameters.rewire) {
				mode = 2;
			}
		}
		pp_rrt* h = nullptr;
		pp_rrt_result r {};
		ppCheck(pp_rrt_run(HipContext::Get(), m_validator ? m_validator->Device() : nullptr, m_lb.v, m_ub.v, params, m_init.v, m_goal.v, m_seed, mode, &h, &r));
		m_path.assign((size_t)r.n_path, Point2d());
		if (r.n_path)
			ppCheck(pp_rrt_get(h, nullptr, nullptr, nullptr, &m_path[0].v[0]));
		pp_rrt_destroy(h);
		m_result = r;
		return r.status == 0 ? Status::Success : Status::Failure;
	}
	std::vector<Point2d> GetPath() const override { return m_path; }
	const pp_rrt_result& GetResult() const { return m_result; }
private:
	Point2d m_lb, m_ub;
	Ref<StateValidatorOccupancyMap> m_validator;
	Params m_parameters;
	std::vector<Point2d> m_path;
	pp_rrt_result m_result {};
	uint64_t m_seed = 0;
};
using RRTR2 = RRTR2Hip<false, RRTParameters>;
using RRTStarR2 = RRTR2Hip<true, RRTStarParameters>;

} // namespace Planner
