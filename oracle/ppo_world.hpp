// TEST INFRASTRUCTURE ONLY -- CPU oracle (see ppo_geometry.hpp header).
//
// World model restated from the reference: occupancy map + coordinate
// transforms, obstacle outline rasterisation, Lau-style dynamic brushfire
// (distance / Voronoi fields, Dolgov path-cost potential) and the
// occupancy-map state validator (the "collision check").
#pragma once

#include "ppo_geometry.hpp"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <queue>
#include <set>
#include <stdexcept>
#include <string>

namespace ppo {

// ---------------------------------------------------------------------------
// utils/grid.{h,cpp}
// ---------------------------------------------------------------------------
struct Cell { // utils/grid.h:8-22
	int row = -1, col = -1;
	Cell() = default;
	Cell(int r, int c) :
		row(r), col(c) { }
	bool IsValid() const { return row >= 0 && col >= 0; }
	bool operator==(const Cell& o) const { return row == o.row && col == o.col; }
	bool operator!=(const Cell& o) const { return !(*this == o); }
	bool IsDiagonalTo(const Cell& o) const { return row != o.row && col != o.col; } // utils/grid.cpp:64-69
	bool IsAdjacentTo(const Cell& o) const
	{
		// utils/grid.cpp:52-62
		if (*this == o)
			return false;
		int dr = row - o.row, dc = col - o.col;
		return (dr >= -1 && dr <= 1) && (dc >= -1 && dc <= 1);
	}
};

/// utils/grid.cpp:16-50 -- fixed enumeration order (Appendix A Q4):
/// (r,c-1),(r-1,c-1),(r+1,c-1),(r,c+1),(r-1,c+1),(r+1,c+1),(r-1,c),(r+1,c).
inline int GetNeighbors(const Cell& cell, int rows, int columns, Cell out[8])
{
	if (!cell.IsValid())
		return 0;
	int n = 0;
	const int c = cell.col, r = cell.row;
	bool left = c - 1 >= 0, right = c + 1 < columns, bottom = r - 1 >= 0, top = r + 1 < rows;
	if (left) {
		out[n++] = Cell(r, c - 1);
		if (bottom)
			out[n++] = Cell(r - 1, c - 1);
		if (top)
			out[n++] = Cell(r + 1, c - 1);
	}
	if (right) {
		out[n++] = Cell(r, c + 1);
		if (bottom)
			out[n++] = Cell(r - 1, c + 1);
		if (top)
			out[n++] = Cell(r + 1, c + 1);
	}
	if (bottom)
		out[n++] = Cell(r - 1, c);
	if (top)
		out[n++] = Cell(r + 1, c);
	return n;
}

template <typename T>
struct Grid { // utils/grid.h:64-99, row-major, index = row * columns + col
	int rows, columns;
	std::vector<T> data;
	Grid(int r, int c, const T& val) :
		rows(r), columns(c)
	{
		if (!(r > 0 && c > 0))
			throw std::invalid_argument("Invalid grid size: received " + std::to_string(r) + " x " + std::to_string(c));
		data.assign((size_t)r * c, val);
	}
	T& at(int r, int c) { return data[(size_t)r * columns + c]; }
	const T& at(int r, int c) const { return data[(size_t)r * columns + c]; }
	T& at(const Cell& c) { return at(c.row, c.col); }
	const T& at(const Cell& c) const { return at(c.row, c.col); }
};

// ---------------------------------------------------------------------------
// state_validator/gvd.{h,cpp} -- dynamic brushfire
// ---------------------------------------------------------------------------
struct QCell { // GridCell<int>, utils/grid.h:43-62
	Cell position;
	int value;
};
struct QGreater {
	bool operator()(const QCell& a, const QCell& b) const { return a.value > b.value; }
};
// gvd.h:19 -- std::priority_queue with std::greater: same libstdc++ heap
// algorithm as the reference build, so tie order among equal distances matches.
using BrushQueue = std::priority_queue<QCell, std::vector<QCell>, QGreater>;

inline int SquaredDistance(const Cell& a, const Cell& b)
{
	// gvd.cpp:12-17
	int dr = a.row - b.row, dc = a.col - b.col;
	return dr * dr + dc * dc;
}

struct VoronoiDistanceMap { // gvd.h:75-108, gvd.cpp:191-256
	int rows, columns;
	float resolution;
	Grid<int> distance;
	Grid<Cell> edge;
	Grid<uint8_t> toRaise, toProcess;
	BrushQueue open;

	VoronoiDistanceMap(int r, int c, float res) :
		rows(r), columns(c), resolution(res), distance(r, c, INT_MAX), edge(r, c, Cell(-1, -1)), toRaise(r, c, 0), toProcess(r, c, 0) { }

	bool IsOccupied(const Cell& s) const { return s.IsValid() && edge.at(s) == s; }

	void SetEdge(const Cell& s)
	{
		edge.at(s) = s;
		distance.at(s) = 0;
		open.push({ s, 0 });
		toProcess.at(s) = 1;
	}
	void UnsetEdge(const Cell& s)
	{
		distance.at(s) = INT_MAX;
		edge.at(s) = Cell(-1, -1);
		toRaise.at(s) = 1;
		open.push({ s, INT_MAX });
		toProcess.at(s) = 1;
	}
	void Update()
	{
		// gvd.cpp:200-237
		Cell nb[8];
		while (!open.empty()) {
			const Cell s = open.top().position;
			open.pop();
			if (!toProcess.at(s))
				continue;
			if (toRaise.at(s)) {
				int nn = GetNeighbors(s, rows, columns, nb);
				for (int q = 0; q < nn; q++) {
					const Cell& n = nb[q];
					if (edge.at(n).IsValid() && !toRaise.at(n)) {
						if (!IsOccupied(edge.at(n))) {
							distance.at(n) = INT_MAX;
							edge.at(n) = Cell(-1, -1);
							toRaise.at(n) = 1;
						}
						open.push({ n, distance.at(n) });
						toProcess.at(n) = 1;
					}
				}
				toRaise.at(s) = 0;
			} else if (IsOccupied(edge.at(s))) {
				toProcess.at(s) = 0;
				int nn = GetNeighbors(s, rows, columns, nb);
				for (int q = 0; q < nn; q++) {
					const Cell& n = nb[q];
					if (!toRaise.at(n)) {
						int d = SquaredDistance(edge.at(s), n);
						if (d < distance.at(n)) {
							distance.at(n) = d;
							edge.at(n) = edge.at(s);
							open.push({ n, d });
							toProcess.at(n) = 1;
						}
					}
				}
			}
		}
	}
	float GetDistanceToNearestVoronoiEdge(int r, int c) const { return std::sqrt(distance.at(r, c)) * resolution; } // gvd.h:85
};

struct ObstacleDistanceMap { // gvd.h:25-72, gvd.cpp:19-189
	int rows, columns;
	float resolution;
	Grid<int> distance;
	Grid<Cell> obstacle;
	Grid<uint8_t> toRaise, toProcess, voro;
	BrushQueue open;
	const Grid<int>* occupancy;
	VoronoiDistanceMap* voronoiMap = nullptr; // owned by World

	ObstacleDistanceMap(const Grid<int>* occ, float res) :
		rows(occ->rows), columns(occ->columns), resolution(res), distance(rows, columns, INT_MAX), obstacle(rows, columns, Cell(-1, -1)),
		toRaise(rows, columns, 0), toProcess(rows, columns, 0), voro(rows, columns, 1), occupancy(occ) { }

	bool IsOccupied(const Cell& s) const { return s.IsValid() && obstacle.at(s) == s; } // gvd.cpp:186-189

	void SetObstacle(const Cell& s)
	{
		// gvd.cpp:74-80
		obstacle.at(s) = s;
		distance.at(s) = 0;
		open.push({ s, 0 });
		toProcess.at(s) = 1;
	}
	void UnsetObstacle(const Cell& s)
	{
		// gvd.cpp:82-89
		distance.at(s) = INT_MAX;
		obstacle.at(s) = Cell(-1, -1);
		toRaise.at(s) = 1;
		open.push({ s, INT_MAX });
		toProcess.at(s) = 1;
	}

	void CheckVoro(const Cell& s, const Cell& n)
	{
		// gvd.cpp:105-131
		Cell obstS = obstacle.at(s), obstN = obstacle.at(n);
		// obstN may be (-1,-1) here; the reference then reads occupancy[-1][-1]
		// (one int before row -1) -- undefined.  We treat an invalid obstN as
		// "different obstacle" and let the IsValid() test below reject it,
		// which is the only outcome that does not depend on stray memory.
		if (obstN.IsValid() && occupancy->at(obstS) == occupancy->at(obstN))
			return;
		if ((distance.at(s) > 1 || distance.at(n) > 1) && obstN.IsValid()) {
			if (std::abs(obstS.row - obstN.row) > 1 || std::abs(obstS.col - obstN.col) > 1) {
				int sObstN = SquaredDistance(s, obstN);
				int nObstS = SquaredDistance(n, obstS);
				int sStability = sObstN - distance.at(s);
				int nStability = nObstS - distance.at(n);
				if (sStability < 0 || nStability < 0)
					return;
				if (sStability <= nStability) {
					voro.at(s) = 1;
					voronoiMap->SetEdge(s);
				}
				if (nStability <= sStability) {
					voro.at(n) = 1;
					voronoiMap->SetEdge(n);
				}
			}
		}
	}

	void Update()
	{
		// gvd.cpp:30-72
		Cell nb[8];
		while (!open.empty()) {
			const Cell s = open.top().position;
			open.pop();
			if (!toProcess.at(s))
				continue;
			if (toRaise.at(s)) {
				int nn = GetNeighbors(s, rows, columns, nb);
				for (int q = 0; q < nn; q++) {
					const Cell& n = nb[q];
					if (obstacle.at(n).IsValid() && !toRaise.at(n)) {
						if (!IsOccupied(obstacle.at(n))) {
							distance.at(n) = INT_MAX;
							obstacle.at(n) = Cell(-1, -1);
							toRaise.at(n) = 1;
						}
						open.push({ n, distance.at(n) });
						toProcess.at(n) = 1;
					}
				}
				toRaise.at(s) = 0;
			} else if (IsOccupied(obstacle.at(s))) {
				voro.at(s) = 0;
				if (voronoiMap)
					voronoiMap->UnsetEdge(s);
				toProcess.at(s) = 0;
				int nn = GetNeighbors(s, rows, columns, nb);
				for (int q = 0; q < nn; q++) {
					const Cell& n = nb[q];
					if (!toRaise.at(n)) {
						int d = SquaredDistance(obstacle.at(s), n);
						if (d < distance.at(n)) {
							distance.at(n) = d;
							obstacle.at(n) = obstacle.at(s);
							open.push({ n, d });
							toProcess.at(n) = 1;
						} else if (voronoiMap) {
							CheckVoro(s, n);
						}
					}
				}
			}
		}
	}

	/// gvd.h:38 -- std::sqrt(int) promotes to double; times float resolution
	/// (double multiply); returned as float.
	float GetDistanceToNearestObstacle(int r, int c) const { return std::sqrt(distance.at(r, c)) * resolution; }
};

// ---------------------------------------------------------------------------
// state_validator/occupancy_map.{h,cpp} + obstacle_list_occupancy_map.cpp +
// state_space/state_space_se2.cpp + state_validator_occupancy_map.cpp
// ---------------------------------------------------------------------------
struct World {
	// state space bounds (state_space/state_space.h:60-61)
	Pose2d lb, ub;
	// occupancy map (state_validator/occupancy_map.h:122-133)
	float resolution;
	int rows = -1, columns = -1;
	Point2d localOrigin; // always (0,0) unless SetPosition
	Point2d localGridOrigin, worldGridOrigin;
	Grid<int>* occ = nullptr;
	ObstacleDistanceMap* obstacleMap = nullptr;
	VoronoiDistanceMap* voronoiMap = nullptr;
	Grid<float>* pathCost = nullptr; // GVD::PathCostMap
	// Optional overrides: when set, validator / planner read these grids
	// instead of the brushfire results (lets tests feed arbitrary inputs).
	std::vector<int> d2Override;
	std::vector<float> pathCostOverride;
	// validator tunables (state_validator_occupancy_map.h:27-28)
	float minPathInterpolationDistance = 0.1f;
	float minSafeRadius = 1.0f;
	// GVD constants (gvd.h:181)
	float alpha = 20.0f, dMax = 30.0f;
	std::set<unsigned int> obstacleIDs;
	// counters
	mutable uint64_t nStateChecks = 0, nPathChecks = 0;

	World(const Pose2d& lb_, const Pose2d& ub_, float res) :
		lb(lb_), ub(ub_), resolution(res)
	{
		// state_validator_occupancy_map.cpp:6-13 (float width/height)
		float width = ub.x - lb.x;
		float height = ub.y - lb.y;
		// occupancy_map.cpp:6-14
		localGridOrigin = { -width / 2.0, -height / 2.0 };
		worldGridOrigin = { localOrigin.x + localGridOrigin.x, localOrigin.y + localGridOrigin.y };
		rows = std::ceil(width / resolution);
		columns = std::ceil(height / resolution);
		occ = new Grid<int>(rows, columns, -1);
		obstacleMap = new ObstacleDistanceMap(occ, resolution);
		// HybridAStar::Initialize -> GVD ctor -> GetVoronoiDistanceMap (gvd.cpp:285-292, 91-98)
		voronoiMap = new VoronoiDistanceMap(rows, columns, resolution);
		obstacleMap->voronoiMap = voronoiMap;
		pathCost = new Grid<float>(rows, columns, 0.0f);
	}
	~World()
	{
		delete pathCost;
		delete voronoiMap;
		delete obstacleMap;
		delete occ;
	}
	World(const World&) = delete;
	World& operator=(const World&) = delete;

	bool IsInsideMap(const Cell& c) const { return c.row >= 0 && c.row < rows && c.col >= 0 && c.col < columns; } // occupancy_map.cpp:27-30

	/// occupancy_map.h:106-117 -- truncation toward zero, double divide by the float resolution.
	Cell GridPositionToGridCell(double x, double y, bool bounded) const
	{
		int row = static_cast<int>(x / resolution);
		int col = static_cast<int>(y / resolution);
		if (!bounded)
			return Cell(row, col);
		else if (IsInsideMap(Cell(row, col)))
			return Cell(row, col);
		else
			return Cell(-1, -1);
	}
	/// occupancy_map.h:175-183
	Cell WorldPositionToGridCell(double x, double y, bool bounded = true) const
	{
		return GridPositionToGridCell(x - worldGridOrigin.x, y - worldGridOrigin.y, bounded);
	}
	/// occupancy_map.h:94-97,150-153: world position of a cell's corner.
	Point2d GridCellToWorldPosition(const Cell& c) const
	{
		return { worldGridOrigin.x + c.row * resolution, worldGridOrigin.y + c.col * resolution };
	}

	bool IsOccupied(const Cell& c) const { return occ->at(c) >= 0; } // obstacle_list_occupancy_map.cpp:63-69

	/// state_space_se2.cpp:15-25
	bool ValidateBounds(const Pose2d& s) const
	{
		if (s.x < lb.x || s.x > ub.x)
			return false;
		if (s.y < lb.y || s.y > ub.y)
			return false;
		if (s.theta < lb.theta || s.theta > ub.theta)
			return false;
		return true;
	}

	float DistanceAt(const Cell& c) const
	{
		if (!d2Override.empty())
			return std::sqrt(d2Override[(size_t)c.row * columns + c.col]) * resolution; // same expression as gvd.h:38
		return obstacleMap->GetDistanceToNearestObstacle(c.row, c.col);
	}
	int Dist2At(int r, int c) const
	{
		if (!d2Override.empty())
			return d2Override[(size_t)r * columns + c];
		return obstacleMap->distance.at(r, c);
	}
	float PathCostAt(int r, int c) const
	{
		if (!pathCostOverride.empty())
			return pathCostOverride[(size_t)r * columns + c];
		return pathCost->at(r, c);
	}

	/// state_validator_occupancy_map.cpp:15-26
	bool IsStateValid(const Pose2d& state) const
	{
		nStateChecks++;
		Pose2d localState = Pose2d(state.x - localOrigin.x, state.y - localOrigin.y, state.theta);
		Cell cell = WorldPositionToGridCell(state.x, state.y);
		if (!ValidateBounds(localState) || !IsInsideMap(cell))
			return false;
		float distance = DistanceAt(cell);
		return distance >= minSafeRadius;
	}

	/// state_validator_occupancy_map.cpp:28-71, generic over the path type.
	template <typename PathT>
	bool IsPathValid(const PathT& path, float* last = nullptr) const
	{
		nPathChecks++;
		const double pathLength = path.length;
		if (pathLength == 0.0) {
			if (last)
				*last = 1.0f;
			return IsStateValid(path.init);
		}
		double lastValidLength = 0.0;
		double length = 0.0;
		while (length < pathLength) {
			Pose2d state = path.Interpolate(length / pathLength);
			if (!IsStateValid(state)) {
				if (last)
					*last = lastValidLength / pathLength;
				return false;
			}
			lastValidLength = length;
			float distToMapBorder = std::min({ state.x - lb.x, ub.x - state.x, state.y - lb.y, ub.y - state.y });
			Cell cell = WorldPositionToGridCell(state.x, state.y);
			float distance = DistanceAt(cell);
			float deltaLength = distance - minSafeRadius;
			deltaLength = std::min(deltaLength, distToMapBorder);
			length += std::max(deltaLength, minPathInterpolationDistance);
		}
		if (last)
			*last = 1.0f;
		return true;
	}

	// -- obstacles: state_validator/obstacle.cpp -----------------------------
	/// obstacle.cpp:7-61 -- Bresenham between the cells of two world points.
	void RasterizeLine(const Point2d& p0, const Point2d& p1, std::vector<Cell>& line) const
	{
		Cell c0 = WorldPositionToGridCell(p0.x, p0.y, false);
		Cell c1 = WorldPositionToGridCell(p1.x, p1.y, false);
		int x0 = c0.row, y0 = c0.col, x1 = c1.row, y1 = c1.col;
		bool steep = std::abs(y1 - y0) > std::abs(x1 - x0);
		if (steep) {
			std::swap(x0, y0);
			std::swap(x1, y1);
		}
		if (x0 > x1) {
			std::swap(x0, x1);
			std::swap(y0, y1);
		}
		int dx = x1 - x0;
		int dy = std::abs(y1 - y0);
		int err = dx / 2;
		int ystep = y0 < y1 ? 1 : -1;
		int y = y0;
		for (int x = x0; x <= x1; x++) {
			Cell c = steep ? Cell(y, x) : Cell(x, y);
			if (c.row >= 0 && c.row < rows && c.col >= 0 && c.col < columns)
				line.push_back(c);
			err = err - dy;
			if (err < 0) {
				y += ystep;
				err += dx;
			}
		}
	}

	/// obstacle.cpp:77-95 -- polygon outline cells at `pose`.
	std::vector<Cell> PolygonCells(const std::vector<Point2d>& vertices, const Pose2d& pose) const
	{
		std::vector<Point2d> w;
		w.reserve(vertices.size());
		const double s = std::sin(pose.theta), c = std::cos(pose.theta);
		for (const auto& v : vertices)
			w.push_back({ c * v.x + (-s) * v.y + pose.x, s * v.x + c * v.y + pose.y });
		std::vector<Cell> cells;
		for (size_t i = 0; i < vertices.size(); i++)
			RasterizeLine(w[i % vertices.size()], w[(i + 1) % vertices.size()], cells);
		return cells;
	}

	static unsigned int FindSmallestIDAvailable(const std::set<unsigned int>& IDs)
	{
		// obstacle_list_occupancy_map.cpp:13-27
		if (IDs.empty())
			return 0;
		if (*IDs.begin() > 0)
			return 0;
		auto res = std::adjacent_find(IDs.begin(), IDs.end(), [](unsigned int a, unsigned int b) { return a + 1 != b; });
		if (res == IDs.end())
			return *IDs.rbegin() + 1;
		return *res + 1;
	}

	/// obstacle_list_occupancy_map.cpp:29-44; returns the id used.
	unsigned int AddObstacleCells(const std::vector<Cell>& cells)
	{
		unsigned int id = FindSmallestIDAvailable(obstacleIDs);
		obstacleIDs.insert(id);
		for (const auto& cell : cells) {
			occ->at(cell) = id;
			obstacleMap->SetObstacle(cell);
		}
		return id;
	}
	/// obstacle_list_occupancy_map.cpp:46-61
	void RemoveObstacleCells(unsigned int id, const std::vector<Cell>& cells)
	{
		obstacleIDs.erase(id);
		for (const auto& cell : cells) {
			occ->at(cell) = -1;
			obstacleMap->UnsetObstacle(cell);
		}
	}

	static std::vector<Point2d> RectangleVertices(double dx, double dy)
	{
		// obstacle.cpp:105-110
		double dx2 = dx / 2.0, dy2 = dy / 2.0;
		return { { dx2, dy2 }, { -dx2, dy2 }, { -dx2, -dy2 }, { dx2, -dy2 } };
	}
	static std::vector<Point2d> CircleVertices(double radius, int count)
	{
		// obstacle.cpp:112-122
		radius *= 1.0 / std::cos(M_PI / count);
		std::vector<Point2d> v;
		for (int i = 0; i < count; i++)
			v.push_back({ radius * std::cos(2 * M_PI * i / (float)count), radius * std::sin(2 * M_PI * i / (float)count) });
		return v;
	}
	static std::vector<Point2d> RegularPolygonVertices(double radius, int count)
	{
		// obstacle.cpp:97-103
		std::vector<Point2d> v;
		for (int i = 0; i < count; i++)
			v.push_back({ radius * std::cos(2 * M_PI * i / (float)count), radius * std::sin(2 * M_PI * i / (float)count) });
		return v;
	}

	/// gvd.cpp:266-283 (PathCostMap::Update) preceded by the two brushfires
	/// (gvd.cpp:294-301, GVD::Update).
	void UpdateGVD()
	{
		obstacleMap->Update();
		voronoiMap->Update();
		for (int r = 0; r < rows; r++) {
			for (int c = 0; c < columns; c++) {
				float obstDist = obstacleMap->GetDistanceToNearestObstacle(r, c);
				float voroDist = voronoiMap->GetDistanceToNearestVoronoiEdge(r, c);
				if (obstDist >= dMax || voroDist == std::numeric_limits<float>::infinity()) {
					pathCost->at(r, c) = 0.0f;
				} else {
					pathCost->at(r, c) = (alpha / (alpha + obstDist)) * (voroDist / (obstDist + voroDist)) * (std::pow(obstDist - dMax, 2) / std::pow(dMax, 2));
				}
			}
		}
	}
};

} // namespace ppo
