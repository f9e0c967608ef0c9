// Map authoring side of the plugin surface, same names as the reference:
//   state_validator/obstacle.{h,cpp}                   Shape, CompositeShape, PolygonShape, RegularPolygonShape, RectangleShape,
//                                                      CircleShape, Obstacle
//   state_validator/obstacle_list_occupancy_map.{h,cpp} ObstacleListOccupancyMap (AddObstacle / RemoveObstacle / ids)
//   state_validator/gvd.{h,cpp}                        GVD (Update, accessors, Visualize)
// Vertices are rotated and translated on the host (a handful of points per obstacle, the reference's own expression); the
// outline cells are walked by the device's Bresenham (pp_rasterize_cells), written into the device occupancy grid
// (pp_map_set_cells), and the distance / Voronoi / path-cost fields are built there (pp_map_update_gvd).  Host accessors read
// copies fetched on demand.
#pragma once

#include <algorithm>
#include <cstdio>
#include <map>
#include <set>
#include <tuple>

#include "planner_hip.hpp"

namespace Planner {

/// state_validator/obstacle.h:13-33
class Shape {
public:
	virtual ~Shape() = default;
	/// boundary cells on `map` of the shape placed at `pose`, appended to `cells`
	virtual void GetGridCellsPosition(OccupancyMap& map, const Pose2d& pose, std::vector<GridCellPosition>& cells) = 0;
	/// world positions of the vertices, appended to `points`
	virtual void GetVerticesPosition(const Pose2d& pose, std::vector<Point2d>& points) = 0;

protected:
	/// Shape::RasterizeLine (obstacle.cpp:7-61) for a closed or open chain of world points, on the device: segment k joins
	/// from[k] and to[k]; in-map cells in Bresenham order, segment after segment
	static void RasterizeLines(OccupancyMap& map, const std::vector<Point2d>& from, const std::vector<Point2d>& to, std::vector<GridCellPosition>& cells)
	{
		const int n = (int)from.size();
		if (n == 0)
			return;
		const int cap = map.Rows() + map.Columns() + 2; // a line holds at most max(rows, columns) in-map cells
		std::vector<int32_t> rc((size_t)n * cap * 2), count((size_t)n);
		ppCheck(pp_rasterize_cells(map.Device(), n, &from[0].v[0], &to[0].v[0], cap, rc.data(), count.data()));
		for (int k = 0; k < n; k++)
			for (int i = 0; i < count[k]; i++)
				cells.push_back({ rc[((size_t)k * cap + i) * 2], rc[((size_t)k * cap + i) * 2 + 1] });
	}
};

/// obstacle.h:35-48
class CompositeShape : public Shape {
public:
	void Add(const Ref<Shape>& shape) { m_children.push_back(shape); }
	void GetGridCellsPosition(OccupancyMap& map, const Pose2d& pose, std::vector<GridCellPosition>& cells) override
	{
		for (auto& child : m_children)
			child->GetGridCellsPosition(map, pose, cells);
	}
	void GetVerticesPosition(const Pose2d& pose, std::vector<Point2d>& points) override
	{
		for (auto& child : m_children)
			child->GetVerticesPosition(pose, points);
	}

private:
	std::vector<Ref<Shape>> m_children;
};

/// obstacle.h:50-66, obstacle.cpp:77-93: the outline only (edges i -> i+1, closed), interiors are not filled (Appendix A Q17)
class PolygonShape : public Shape {
public:
	explicit PolygonShape(const std::vector<Point2d>& vertices) : m_vertices(vertices) { }
	void GetGridCellsPosition(OccupancyMap& map, const Pose2d& pose, std::vector<GridCellPosition>& cells) override final
	{
		std::vector<Point2d> from;
		GetVerticesPosition(pose, from);
		std::vector<Point2d> to(from.size());
		for (size_t i = 0; i < from.size(); i++)
			to[i] = from[(i + 1) % from.size()];
		RasterizeLines(map, from, to, cells);
	}
	void GetVerticesPosition(const Pose2d& pose, std::vector<Point2d>& points) override
	{
		const double c = std::cos(pose.theta), s = std::sin(pose.theta);
		for (const auto& v : m_vertices) // Eigen::Rotation2D(theta) * vertex + position
			points.push_back(Point2d(c * v.x() - s * v.y(), s * v.x() + c * v.y()) + pose.position);
	}

protected:
	PolygonShape() = default;
	std::vector<Point2d> m_vertices;
};

/// obstacle.cpp:95-103 (the angle is divided by a float count, as there)
class RegularPolygonShape : public PolygonShape {
public:
	RegularPolygonShape(double radius, int count)
	{
		for (int i = 0; i < count; i++)
			m_vertices.push_back({ radius * std::cos(2 * M_PI * i / (float)count), radius * std::sin(2 * M_PI * i / (float)count) });
	}
};
/// obstacle.cpp:105-110
class RectangleShape : public PolygonShape {
public:
	RectangleShape(double dx, double dy)
	{
		const double dx2 = dx / 2.0, dy2 = dy / 2.0;
		m_vertices = { Point2d(dx2, dy2), Point2d(-dx2, dy2), Point2d(-dx2, -dy2), Point2d(dx2, -dy2) };
	}
};
/// obstacle.cpp:112-122: a `count`-gon inflated so that it contains the circle
class CircleShape : public PolygonShape {
public:
	CircleShape(double radius, int count)
	{
		radius *= 1.0 / std::cos(M_PI / count);
		for (int i = 0; i < count; i++)
			m_vertices.push_back({ radius * std::cos(2 * M_PI * i / (float)count), radius * std::sin(2 * M_PI * i / (float)count) });
	}
};

/// obstacle.h:82-98
class Obstacle {
public:
	void SetShape(const Ref<Shape>& shape) { m_shape = shape; }
	void SetPose(const Pose2d& pose) { m_pose = pose; }
	std::vector<GridCellPosition> GetBoundaryGridCellPosition(OccupancyMap& map)
	{
		std::vector<GridCellPosition> cells;
		if (m_shape)
			m_shape->GetGridCellsPosition(map, m_pose, cells);
		return cells;
	}
	std::vector<Point2d> GetBoundaryWorldPosition()
	{
		std::vector<Point2d> points;
		if (m_shape)
			m_shape->GetVerticesPosition(m_pose, points);
		return points;
	}

private:
	Ref<Shape> m_shape;
	Pose2d m_pose;
};

/// obstacle_list_occupancy_map.{h,cpp}
class ObstacleListOccupancyMap : public OccupancyMap {
public:
	explicit ObstacleListOccupancyMap(float resolution) : OccupancyMap(resolution) { }
	/// boundary cells take the smallest free id (:13-27, :29-44); false if the obstacle is already on the map
	bool AddObstacle(const Ref<Obstacle>& obstacle)
	{
		CheckSized();
		const unsigned int id = SmallestFreeId();
		if (!m_obstacles.insert({ obstacle, id }).second)
			return false;
		m_ids.insert(id);
		SetCellsOnDevice(obstacle->GetBoundaryGridCellPosition(*this), (int32_t)id);
		return true;
	}
	/// boundary cells go back to -1, also where another outline crossed them (:46-61)
	bool RemoveObstacle(const Ref<Obstacle>& obstacle)
	{
		CheckSized();
		auto it = m_obstacles.find(obstacle);
		if (it == m_obstacles.end())
			return false;
		m_ids.erase(it->second);
		m_obstacles.erase(it);
		SetCellsOnDevice(obstacle->GetBoundaryGridCellPosition(*this), -1);
		return true;
	}
	int GetNumObstacles() const { return (int)m_obstacles.size(); }
	bool IsOccupied(const GridCellPosition& cell) override
	{
		CheckSized();
		return OccupancyMap::IsOccupied(cell);
	}

private:
	void CheckSized() const
	{
		if (m_rows <= 0)
			throw std::runtime_error("The size of the occupancy matrix has not been initialized");
	}
	unsigned int SmallestFreeId() const
	{
		unsigned int id = 0;
		for (unsigned int used : m_ids) { // ascending
			if (used != id)
				break;
			id++;
		}
		return id;
	}
	std::map<Ref<Obstacle>, unsigned int> m_obstacles;
	std::set<unsigned int> m_ids;
};

/// state_validator/gvd.{h,cpp}: the three fields over an occupancy map.  By default the two distance maps are the reference's own
/// brushfire (OccupancyMap::FieldUpdateMode::ReferenceOrder: bit-identical grids); SetUpdateMode(ExactTransform) opts into the
/// device's exact Euclidean transform (see pp_gvd.hip for how the two relate)
class GVD {
public:
	explicit GVD(const Ref<OccupancyMap>& map) : rows(map->Rows()), columns(map->Columns()), resolution(map->resolution), m_map(map) { }
	/// gvd.cpp:294-301
	void Update() { m_map->BuildFields(alpha, dMax); }
	void SetUpdateMode(OccupancyMap::FieldUpdateMode mode) { m_map->SetFieldUpdateMode(mode); }
	GridCellPosition GetNearestObstacleCell(int row, int col) const { return Cell(m_map->Voronoi().nearestObstacle, row, col); }
	GridCellPosition GetNearestObstacleCell(const GridCellPosition& c) const { return GetNearestObstacleCell(c.row, c.col); }
	GridCellPosition GetNearestVoronoiEdgeCell(int row, int col) const { return Cell(m_map->Voronoi().nearestEdge, row, col); }
	GridCellPosition GetNearestVoronoiEdgeCell(const GridCellPosition& c) const { return GetNearestVoronoiEdgeCell(c.row, c.col); }
	float GetDistanceToNearestObstacle(int row, int col) const { return m_map->GetDistanceToNearestObstacle(row, col); }
	float GetDistanceToNearestObstacle(const GridCellPosition& c) const { return GetDistanceToNearestObstacle(c.row, c.col); }
	float GetDistanceToNearestVoronoiEdge(int row, int col) const { return m_map->DistanceOf(m_map->Voronoi().d2[(size_t)row * columns + col]); } // gvd.h:77
	float GetDistanceToNearestVoronoiEdge(const GridCellPosition& c) const { return GetDistanceToNearestVoronoiEdge(c.row, c.col); }
	float GetPathCost(int row, int col) const { return m_map->GetPathCost(row, col); }
	float GetPathCost(const GridCellPosition& c) const { return GetPathCost(c.row, c.col); }
	/// gvd.cpp:303-352: false outside the map
	bool GetNearestObstaclePosition(const Point2d& position, Point2d& obstacle) const
	{
		return At(position, [&](const GridCellPosition& c) { obstacle = m_map->GridCellToWorldPosition(GetNearestObstacleCell(c)); });
	}
	bool GetNearestVoronoiEdgePosition(const Point2d& position, Point2d& voronoi) const
	{
		return At(position, [&](const GridCellPosition& c) { voronoi = m_map->GridCellToWorldPosition(GetNearestVoronoiEdgeCell(c)); });
	}
	bool GetDistanceToNearestObstacle(const Point2d& position, float& distance) const
	{
		return At(position, [&](const GridCellPosition& c) { distance = GetDistanceToNearestObstacle(c); });
	}
	bool GetDistanceToNearestVoronoiEdge(const Point2d& position, float& distance) const
	{
		return At(position, [&](const GridCellPosition& c) { distance = GetDistanceToNearestVoronoiEdge(c); });
	}
	bool GetPathCost(const Point2d& position, float& cost) const
	{
		return At(position, [&](const GridCellPosition& c) { cost = GetPathCost(c); });
	}
	/// gvd.cpp:385-427: PPM image, hue = nearest obstacle's id, brightness = 1 - path cost
	void Visualize(const std::string& filename) const
	{
		FILE* F = std::fopen(filename.c_str(), "w");
		if (!F)
			return;
		int numObstacles = 0;
		for (int x = 0; x < rows; x++)
			for (int y = 0; y < columns; y++)
				numObstacles = std::max(numObstacles, m_map->GetOccupancyValue(x, y));
		std::fprintf(F, "P6\n#\n%d %d\n255\n", rows, columns);
		for (int y = columns - 1; y >= 0; y--) {
			for (int x = 0; x < rows; x++) {
				unsigned char rgb[3] = { 0, 0, 0 };
				const GridCellPosition o = m_map->FieldsBuilt() ? GetNearestObstacleCell(x, y) : GridCellPosition();
				if (!m_map->IsOccupied({ x, y }) && o.IsValid()) {
					const float h = m_map->GetOccupancyValue(o) / (float)(numObstacles + 1);
					const float l = std::max(0.0f, std::min(1.0f - GetPathCost(x, y), 1.0f));
					const float H = h * 360.0f, C = l, X = C * (1 - std::fabs(std::fmod(H / 60.0, 2) - 1));
					float r, g, b;
					if (H < 60) r = C, g = X, b = 0;
					else if (H < 120) r = X, g = C, b = 0;
					else if (H < 180) r = 0, g = C, b = X;
					else if (H < 240) r = 0, g = X, b = C;
					else if (H < 300) r = X, g = 0, b = C;
					else r = C, g = 0, b = X;
					rgb[0] = (unsigned char)(r * 255), rgb[1] = (unsigned char)(g * 255), rgb[2] = (unsigned char)(b * 255);
				}
				std::fwrite(rgb, 1, 3, F);
			}
		}
		std::fclose(F);
	}
	const int rows, columns;
	const float resolution;
	const float alpha = 20.0f, dMax = 30.0f; // gvd.h:181

private:
	GridCellPosition Cell(const std::vector<int32_t>& grid, int row, int col) const
	{
		if (grid.empty())
			return GridCellPosition();
		const size_t i = ((size_t)row * columns + col) * 2;
		return { grid[i], grid[i + 1] };
	}
	template <typename F>
	bool At(const Point2d& position, F f) const
	{
		const GridCellPosition c = m_map->WorldPositionToGridCell(position, false);
		if (!m_map->IsInsideMap(c))
			return false;
		f(c);
		return true;
	}
	Ref<OccupancyMap> m_map;
};

} // namespace Planner
