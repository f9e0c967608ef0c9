// Value types of the host mirror (no GPU work here): geometry/2dplane.h (Point2d, Pose2d), utils/grid.h (GridCellPosition),
// core/base.h (Ref, makeRef), algo/path_planner.h:9-12 (Status), paths/path.h:10-20 (Steer, Direction), and the process-wide GPU
// context the HIP-backed classes share.
#pragma once

#include <array>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pp_hip.h"

namespace Planner {

template <typename T>
using Ref = std::shared_ptr<T>;
template <typename T, typename... Args>
Ref<T> makeRef(Args&&... args) { return std::make_shared<T>(std::forward<Args>(args)...); }

enum Status { Success = 0, Failure = -1 }; // algo/path_planner.h:9-12
enum class Steer { Left, Straight, Right }; // paths/path.h:10-14
enum class Direction { Forward, Backward, NoMotion }; // paths/path.h:16-20

struct Point2d { // geometry/2dplane.h:11-14 (Eigen::Vector2d in the reference)
	double v[2] = { 0.0, 0.0 };
	Point2d() = default;
	Point2d(double x, double y) { v[0] = x; v[1] = y; }
	double& x() { return v[0]; }
	double& y() { return v[1]; }
	const double& x() const { return v[0]; }
	const double& y() const { return v[1]; }
	Point2d operator+(const Point2d& o) const { return { v[0] + o.v[0], v[1] + o.v[1] }; }
	Point2d operator-(const Point2d& o) const { return { v[0] - o.v[0], v[1] - o.v[1] }; }
	bool operator==(const Point2d& o) const { return v[0] == o.v[0] && v[1] == o.v[1]; }
	bool operator!=(const Point2d& o) const { return !(*this == o); }
	double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1]); }
};

struct Pose2d { // geometry/2dplane.h:17-45: 3 contiguous doubles, constructors wrap theta
	Point2d position;
	double theta = 0.0;
	Pose2d() = default;
	Pose2d(const Point2d& p, double t) : position(p), theta(t) { theta = WrapTheta(); }
	Pose2d(double x, double y, double t) : position(x, y), theta(t) { theta = WrapTheta(); }
	double& x() { return position.x(); }
	double& y() { return position.y(); }
	const double& x() const { return position.x(); }
	const double& y() const { return position.y(); }
	double WrapTheta() const
	{
		double t = theta;
		while (t > M_PI)
			t -= 2 * M_PI;
		while (t < -M_PI)
			t += 2 * M_PI;
		return t;
	}
	bool operator==(const Pose2d& o) const { return position == o.position && theta == o.theta; }
	bool operator!=(const Pose2d& o) const { return !(*this == o); }
	/// SE(2) composition: `rhs` expressed in this frame (geometry/2dplane.h:47-62; not commutative)
	Pose2d operator+(const Pose2d& rhs) const
	{
		const double c = std::cos(theta), s = std::sin(theta);
		Pose2d out(c * rhs.x() - s * rhs.y(), s * rhs.x() + c * rhs.y(), theta + rhs.theta);
		out.position = out.position + position;
		return out;
	}
	/// the pose p with rhs + p = *this (geometry/2dplane.h:64-79)
	Pose2d operator-(const Pose2d& rhs) const
	{
		const double dx = x() - rhs.x(), dy = y() - rhs.y();
		const double c = std::cos(-rhs.theta), s = std::sin(-rhs.theta);
		return Pose2d(c * dx - s * dy, s * dx + c * dy, theta - rhs.theta);
	}
};
static_assert(sizeof(Pose2d) == 24, "Pose2d must be 3 contiguous doubles (the ABI's pose layout)");

struct GridCellPosition { // utils/grid.h:8-22, utils/grid.cpp:6-66
	int row = -1, col = -1;
	GridCellPosition() = default;
	GridCellPosition(int r, int c) : row(r), col(c) { }
	bool IsValid() const { return row >= 0 && col >= 0; }
	bool operator==(const GridCellPosition& o) const { return row == o.row && col == o.col; }
	bool operator!=(const GridCellPosition& o) const { return !(*this == o); }
	bool IsAdjacentTo(const GridCellPosition& o) const { return !(*this == o) && std::abs(row - o.row) <= 1 && std::abs(col - o.col) <= 1; }
	bool IsDiagonalTo(const GridCellPosition& o) const { return row != o.row && col != o.col; }
	/// in-grid 8-neighbours in the reference's enumeration order (grid.cpp:29-47):
	/// the column-1 side (same row, row-1, row+1), the column+1 side (same order), then (row-1, col), (row+1, col)
	std::vector<GridCellPosition> GetNeighbors(int rows, int columns) const
	{
		std::vector<GridCellPosition> out;
		if (!IsValid())
			return out;
		out.reserve(8);
		static const int kStep[8][2] = { { 0, -1 }, { -1, -1 }, { 1, -1 }, { 0, 1 }, { -1, 1 }, { 1, 1 }, { -1, 0 }, { 1, 0 } };
		for (const auto& d : kStep) {
			const int r = row + d[0], c = col + d[1];
			if (r >= 0 && r < rows && c >= 0 && c < columns)
				out.push_back({ r, c });
		}
		return out;
	}
};

inline void ppCheck(int rc)
{
	if (rc != 0)
		throw std::runtime_error(std::string("libpphip: ") + pp_last_error());
}

/// One GPU context shared by the objects of a process (device 0 unless PP_DEVICE is set).
class HipContext {
public:
	static pp_ctx* Get()
	{
		static HipContext instance;
		return instance.m_ctx;
	}
private:
	HipContext()
	{
		int dev = 0;
		if (const char* e = std::getenv("PP_DEVICE"))
			dev = std::atoi(e);
		ppCheck(pp_ctx_create(dev, nullptr, &m_ctx));
	}
	~HipContext() { pp_ctx_destroy(m_ctx); }
	pp_ctx* m_ctx = nullptr;
};

} // namespace Planner
