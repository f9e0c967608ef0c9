// Device-side restatement of the per-pose / per-path arithmetic of the hot path.
// gfx950 only.  Built with -ffp-contract=off: expression order and the
// float/double mix follow the reference line by line because discrete outputs
// (cells, heading bins, validity, expansion order) must match it bit for bit.
//
// Reference files (relative to planner/src) are cited per function.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ppd {

#define PPD_INLINE __device__ __forceinline__

constexpr double kPi = 3.14159265358979323846;

// ---- obstacle-heuristic fields between the wavefront and the search kernel are stored in 8 x 8 tiles (one tile =
// 256 contiguous bytes): the wavefront writes each cell once, in ring order, and a ring crosses a tile in a few
// consecutive rounds, so the four lines of a tile fill up while they are still in L2 instead of one write-back
// per 4-byte store (row-major: a vertical or diagonal front touches a different line for every cell).
#ifndef PP_FIELD_TILE_LOG2
#define PP_FIELD_TILE_LOG2 3 // 8 x 8 cells; (experiment) 2 = 4 x 4 cells, one cache line per tile
#endif
constexpr int kFieldTileLog2 = PP_FIELD_TILE_LOG2, kFieldTile = 1 << kFieldTileLog2, kFieldTileMask = kFieldTile - 1;
__host__ __device__ inline size_t field_tiled_index(int cols, int row, int col)
{
	const int tpr = (cols + kFieldTileMask) >> kFieldTileLog2;
	return ((size_t)((row >> kFieldTileLog2) * tpr + (col >> kFieldTileLog2)) << (2 * kFieldTileLog2)) | (size_t)(((row & kFieldTileMask) << kFieldTileLog2) | (col & kFieldTileMask));
}
__host__ __device__ inline size_t field_tiled_elems(int rows, int cols)
{
	return (size_t)((rows + kFieldTileMask) >> kFieldTileLog2) * (size_t)((cols + kFieldTileMask) >> kFieldTileLog2) * (size_t)(kFieldTile * kFieldTile);
}
constexpr double kPi2 = 1.57079632679489661923;

struct Pose {
	double x, y, t;
};

/// Device view of one map set (filled by pp_map_*).
/// lattice resolutions with their correctly rounded reciprocals (see div_by)
struct Resolutions {
	double spatial, angular, invSpatial, invAngular;
	__host__ __device__ void set(double s, double a)
	{
		spatial = s;
		angular = a;
		invSpatial = 1.0 / s;
		invAngular = 1.0 / a;
	}
};

struct MapView {
	int rows, cols;
	float res;              // OccupancyMap::resolution
	double invRes;          // 1.0 / (double)res, correctly rounded on the host (div_by)
	double gx, gy;          // m_worldGridOrigin
	double lox, loy;        // m_localOrigin
	double lbx, lby, lbt;   // StateSpaceSE2 bounds
	double ubx, uby, ubt;
	float minSafeRadius;    // StateValidatorOccupancyMap::minSafeRadius
	float minInterp;        // ...::minPathInterpolationDistance
	const float* dist;      // (float)(sqrt((double)d2) * res): GetDistanceToNearestObstacle, gvd.h:38
	const float* pathcost;  // GVD::PathCostMap
	const uint8_t* occ8;    // 1 = occupied
	const uint32_t* validBits; // bit (row * cols + col): dist >= minSafeRadius, rebuilt when either changes (128 KiB at 1024^2)
};

/// a / b for a divisor whose correctly rounded reciprocal y = 1.0 / b is known (resolutions: wave-uniform kernel
/// arguments): product, exact residual (one fma), correction (one fma) -- Markstein's sequence, which returns the
/// correctly rounded quotient, i.e. the bits of `a / b`, for every finite a.  It replaces the ~12-instruction IEEE
/// division (quarter-rate v_rcp_f64 + Newton + v_div_fixup) on the per-pose paths.  Non-finite a gives NaN where a / b
/// gives +-inf; every caller converts with trunc_to_int, which maps both to INT_MIN.  Checked against `a / b` on the host for
/// 2.7e9 operands including +-4 ulp neighbourhoods of exact multiples (tests/cpp/test_reciprocal_division.c, run by the CPU suite).
PPD_INLINE double div_by(double a, double b, double y)
{
	const double q0 = a * y;
	const double e = fma(-q0, b, a);
	return fma(e, y, q0);
}

/// geometry/2dplane.h:36-45
PPD_INLINE double wrap_theta(double theta)
{
	double t = theta;
	// The reference loops forever on +-inf / astronomically large angles; a GPU wave must not.
	// Beyond 1e4 rad fold with fmod first (degenerate inputs only; NaN falls through unchanged).
	if (fabs(t) > 1.0e4)
		t = fmod(t, 2 * kPi);
	while (t > kPi)
		t -= 2 * kPi;
	while (t < -kPi)
		t += 2 * kPi;
	return t;
}

/// static_cast<int>(double) as x86 cvttsd2si does it for the reference build:
/// out-of-range and NaN give INT_MIN (the GPU conversion saturates / gives 0).
PPD_INLINE int trunc_to_int(double v)
{
	if (!(v > -2147483649.0 && v < 2147483648.0))
		return (int)0x80000000;
	return (int)v;
}

/// OccupancyMap::WorldPositionToGridCell(bounded = false), occupancy_map.h:106-117,180-183
PPD_INLINE void world_to_cell(const MapView& m, double x, double y, int& row, int& col)
{
	row = trunc_to_int(div_by(x - m.gx, (double)m.res, m.invRes));
	col = trunc_to_int(div_by(y - m.gy, (double)m.res, m.invRes));
}

PPD_INLINE bool inside_map(const MapView& m, int row, int col)
{
	return row >= 0 && row < m.rows && col >= 0 && col < m.cols; // occupancy_map.cpp:27-30
}

/// StateValidatorOccupancyMap::IsStateValid, state_validator_occupancy_map.cpp:15-26.
/// On success also returns the cell's obstacle distance (re-read by IsPathValid).
PPD_INLINE bool is_state_valid(const MapView& m, double x, double y, double theta, float& distance)
{
	const double lx = x - m.lox, ly = y - m.loy;
	const double lt = wrap_theta(theta); // Pose2d constructor of `localState`
	int row, col;
	world_to_cell(m, x, y, row, col);
	// StateSpaceSE2::ValidateBounds, state_space_se2.cpp:15-25
	if (lx < m.lbx || lx > m.ubx)
		return false;
	if (ly < m.lby || ly > m.uby)
		return false;
	if (lt < m.lbt || lt > m.ubt)
		return false;
	if (!inside_map(m, row, col))
		return false;
	distance = m.dist[(size_t)row * m.cols + col];
	return distance >= m.minSafeRadius;
}

/// Obstacle-distance window around an expanded node (one-query search kernel): every sample of the node's constant-steer children
/// lies within the arc length (1.5 x the spatial resolution) of its pose, so the (2 * kDistWinHalf + 1)^2 cells around the pose's
/// cell are fetched into LDS in ONE memory round trip when the node is popped; the children's adaptive marches -- a chain of
/// dependent distance reads, two to four per expansion -- then read LDS.  Same values, same arithmetic: only where they are read from.
constexpr int kDistWinHalf = 16, kDistWin = 2 * kDistWinHalf + 1, kDistWinElems = ((kDistWin * kDistWin + 63) / 64) * 64;
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(3))) float* LdsFloatPtr; // an LDS pointer the compiler KNOWS to be one: ds_read, and no
                                                                      // select between an LDS-derived and a global generic address
#else
typedef const float* LdsFloatPtr;
#endif
struct DistWindow {
	LdsFloatPtr win; // LDS, row-major kDistWin x kDistWin, cell (r0 + i, c0 + j) at i * kDistWin + j
	int r0, c0;
	PPD_INLINE float fetch(const MapView& m, int row, int col) const
	{
		const unsigned wr = (unsigned)(row - r0), wc = (unsigned)(col - c0);
		if (wr < (unsigned)kDistWin && wc < (unsigned)kDistWin)
			return win[wr * kDistWin + wc];
		return m.dist[(size_t)row * m.cols + col]; // (arcs longer than the window: never with the reference's 1.5 m arcs at 0.1 m cells)
	}
};
PPD_INLINE bool is_state_valid(const MapView& m, const DistWindow& w, double x, double y, double theta, float& distance)
{
	const double lx = x - m.lox, ly = y - m.loy;
	const double lt = wrap_theta(theta);
	int row, col;
	world_to_cell(m, x, y, row, col);
	if (lx < m.lbx || lx > m.ubx)
		return false;
	if (ly < m.lby || ly > m.uby)
		return false;
	if (lt < m.lbt || lt > m.ubt)
		return false;
	if (!inside_map(m, row, col))
		return false;
	distance = w.fetch(m, row, col);
	return distance >= m.minSafeRadius;
}

/// is_state_valid in two halves, so that the caller can put other work between the load and its first use (a wave
/// stalls at the first instruction that needs a loaded value): `issue` does the bounds tests and starts the distance
/// load, `finish` is the comparison.  Same result as is_state_valid.
PPD_INLINE bool is_state_valid_issue(const MapView& m, double x, double y, double theta, float& rawDistance)
{
	const double lx = x - m.lox, ly = y - m.loy;
	const double lt = wrap_theta(theta);
	int row, col;
	world_to_cell(m, x, y, row, col);
	rawDistance = 0.0f;
	if (lx < m.lbx || lx > m.ubx)
		return false;
	if (ly < m.lby || ly > m.uby)
		return false;
	if (lt < m.lbt || lt > m.ubt)
		return false;
	if (!inside_map(m, row, col))
		return false;
	rawDistance = m.dist[(size_t)row * m.cols + col];
	return true;
}
PPD_INLINE bool is_state_valid_finish(const MapView& m, bool inBounds, float rawDistance) { return inBounds && rawDistance >= m.minSafeRadius; }

/// Same test without the distance: the cell's answer comes from the validity bitmap (one bit per cell, the result of the
/// identical float comparison), so a streamed check moves 24 B in + 1 B out and the 128 KiB bitmap stays in cache
/// instead of a 4-byte gather that pulls a whole line of the 4 MiB distance grid per pose.
PPD_INLINE bool is_state_valid_bit(const MapView& m, double x, double y, double theta)
{
	const double lx = x - m.lox, ly = y - m.loy;
	const double lt = wrap_theta(theta);
	int row, col;
	world_to_cell(m, x, y, row, col);
	if (lx < m.lbx || lx > m.ubx)
		return false;
	if (ly < m.lby || ly > m.uby)
		return false;
	if (lt < m.lbt || lt > m.ubt)
		return false;
	if (!inside_map(m, row, col))
		return false;
	const size_t cell = (size_t)row * m.cols + col;
	return (m.validBits[cell >> 5] >> (cell & 31)) & 1u;
}

/// The same verdict without early exits (the streamed check kernel evaluates four poses per lane: eight nested exits and
/// four inlined wrap loops cost more issue slots than the arithmetic).  Every condition is computed and and-ed; the heading is
/// wrapped only when it lies outside [-pi, pi] (wrap_theta returns such values unchanged); the cell index is forced to 0 where
/// the pose is outside the grid so that the bitmap load is always in range.  NaN coordinates: the reference's (int) conversion
/// yields INT_MIN (outside), the device conversion yields 0 -- hence the explicit ordered tests.
template <typename Bits>
PPD_INLINE bool is_state_valid_bit_flat(const MapView& m, double x, double y, double theta, Bits validBits)
{
	const double lx = x - m.lox, ly = y - m.loy;
	double lt = theta;
	if (fabs(theta) > kPi)
		lt = wrap_theta(theta);
	const double qx = div_by(x - m.gx, (double)m.res, m.invRes), qy = div_by(y - m.gy, (double)m.res, m.invRes);
	const bool inRange = qx > -2147483649.0 && qx < 2147483648.0 && qy > -2147483649.0 && qy < 2147483648.0; // false for NaN
	const int row = (int)qx, col = (int)qy;
	const bool bounds = !(lx < m.lbx) & !(lx > m.ubx) & !(ly < m.lby) & !(ly > m.uby) & !(lt < m.lbt) & !(lt > m.ubt);
	const bool inside = inRange & ((unsigned)row < (unsigned)m.rows) & ((unsigned)col < (unsigned)m.cols);
#ifdef PP_CS_EXPERIMENT_NO_GATHER // measurement only: every lookup hits word 0
	const uint32_t cell = inside ? (uint32_t)(row & 1) : 0u;
#else
	const uint32_t cell = inside ? (uint32_t)row * (uint32_t)m.cols + (uint32_t)col : 0u;
#endif
	const uint32_t bit = (validBits[cell >> 5] >> (cell & 31)) & 1u; // the map's bitmap, or a copy of it in LDS
	return bounds & inside & (bit != 0u);
}

/// KinematicBicycleModel::ConstantSteer with rearToCenter = 0 (beta = 0, cos(beta) = 1),
/// models/kinematic_bicycle_model.cpp:5-32.  `kappa` = DthetaDdist (host libm),
/// `dist` already carries the direction sign.  theta is not wrapped.
PPD_INLINE Pose constant_steer(const Pose& from, double kappa, double dist)
{
	Pose to = from;
	if (fabs(kappa) > 1e-9) {
		to.t += dist * kappa;
		to.x += 1 / kappa * (sin(to.t) - sin(from.t));
		to.y += 1 / kappa * (-cos(to.t) + cos(from.t));
	} else {
		to.x += dist * cos(from.t);
		to.y += dist * sin(from.t);
	}
	return to;
}

/// A constant-steer arc: PathConstantSteer, paths/path_constant_steer.cpp:5-25
struct Arc {
	Pose init;
	double kappa;
	double length;
	int backward; // 1 = Direction::Backward
	PPD_INLINE Pose interpolate(double ratio) const
	{
		double d = length * ratio;
		if (backward)
			d = -d;
		return constant_steer(init, kappa, d);
	}
};

/// The same arc with sin/cos of the initial heading supplied (they are stored with the node that is
/// being expanded) and sin/cos of the interpolated heading returned: one sincos per sample instead of
/// four libm calls.  A zero travelled distance returns the initial pose (x + 1/k*(sin t - sin t) = x).
struct ArcSC {
	Pose init;
	double sinF, cosF;
	double kappa;
	double invKappa; // 1 / kappa, divided once per primitive on the host (PrimTable)
	double length;
	int backward;
	PPD_INLINE Pose interpolate_sc(double ratio, double& s, double& c) const
	{
		double d = length * ratio;
		if (backward)
			d = -d;
		Pose to = init;
		s = sinF;
		c = cosF;
		if (d == 0.0)
			return to;
		if (fabs(kappa) > 1e-9) {
			to.t += d * kappa;
			sincos(to.t, &s, &c);
			to.x += invKappa * (s - sinF);
			to.y += invKappa * (-c + cosF);
		} else {
			to.x += d * cosF;
			to.y += d * sinF;
		}
		return to;
	}
	PPD_INLINE Pose interpolate(double ratio) const
	{
		double s, c;
		return interpolate_sc(ratio, s, c);
	}
};

/// An R2 segment seen as an SE2 path with theta = 0 (paths/path_r2.cpp:11-16)
struct Segment {
	double x0, y0, x1, y1;
	double length;
	PPD_INLINE Pose interpolate(double ratio) const
	{
		Pose p;
		p.x = (1 - ratio) * x0 + ratio * x1;
		p.y = (1 - ratio) * y0 + ratio * y1;
		p.t = 0.0;
		return p;
	}
};

PPD_INLINE float fmin4(double a, double b, double c, double d)
{
	// std::min({a, b, c, d}) on doubles, converted to float (state_validator_occupancy_map.cpp:54-59)
	double r = a;
	if (b < r)
		r = b;
	if (c < r)
		r = c;
	if (d < r)
		r = d;
	return (float)r;
}

/// StateValidatorOccupancyMap::IsPathValid, state_validator_occupancy_map.cpp:28-71.
/// `checks` counts IsStateValid calls.
template <typename PathT>
PPD_INLINE bool is_path_valid(const MapView& m, const PathT& path, const Pose& init, float& last, int& checks)
{
	const double pathLength = path.length;
	float distance = 0.0f;
	if (pathLength == 0.0) {
		last = 1.0f;
		checks++;
		return is_state_valid(m, init.x, init.y, init.t, distance);
	}
	double lastValidLength = 0.0;
	double length = 0.0;
	while (length < pathLength) {
		// the reference spins forever when minPathInterpolationDistance <= 0 stalls the march;
		// every wave must drain, so give up (invalid) after 2^22 samples
		if (checks > (1 << 22)) {
			last = (float)(lastValidLength / pathLength);
			return false;
		}
		Pose s = path.interpolate(length / pathLength);
		checks++;
		if (!is_state_valid(m, s.x, s.y, s.t, distance)) {
			last = (float)(lastValidLength / pathLength);
			return false;
		}
		lastValidLength = length;
		float distToMapBorder = fmin4(s.x - m.lbx, m.ubx - s.x, s.y - m.lby, m.uby - s.y);
		float deltaLength = distance - m.minSafeRadius;
		deltaLength = fminf(deltaLength, distToMapBorder);
		length += (double)fmaxf(deltaLength, m.minInterp);
	}
	last = 1.0f;
	return true;
}

/// Same march when the validity / obstacle distance of the path's start pose is already known (firstDist < 0: the
/// start pose is invalid): the first sample of every child arc is the parent's pose, which was checked when the parent
/// node was created, so the march starts without waiting for a distance load.  Counts that sample like the reference.
struct DistGlobal { };
PPD_INLINE bool is_state_valid(const MapView& m, const DistGlobal&, double x, double y, double theta, float& distance) { return is_state_valid(m, x, y, theta, distance); }

template <typename PathT, typename DistSrc>
PPD_INLINE bool is_path_valid_from(const MapView& m, const DistSrc& src, const PathT& path, const Pose& init, float firstDist, float& last, int& checks);
template <typename PathT>
PPD_INLINE bool is_path_valid_from(const MapView& m, const PathT& path, const Pose& init, float firstDist, float& last, int& checks)
{
	return is_path_valid_from(m, DistGlobal(), path, init, firstDist, last, checks);
}
template <typename PathT, typename DistSrc>
PPD_INLINE bool is_path_valid_from(const MapView& m, const DistSrc& src, const PathT& path, const Pose& init, float firstDist, float& last, int& checks)
{
	const double pathLength = path.length;
	if (pathLength == 0.0) {
		last = 1.0f;
		checks++;
		return !(firstDist < 0.0f);
	}
	checks++;
	if (firstDist < 0.0f) {
		last = 0.0f; // (float)(0.0 / pathLength)
		return false;
	}
	float distance = firstDist;
	double lastValidLength = 0.0;
	double length = 0.0;
	{
		const float distToMapBorder = fmin4(init.x - m.lbx, m.ubx - init.x, init.y - m.lby, m.uby - init.y);
		float deltaLength = distance - m.minSafeRadius;
		deltaLength = fminf(deltaLength, distToMapBorder);
		length += (double)fmaxf(deltaLength, m.minInterp);
	}
	while (length < pathLength) {
		if (checks > (1 << 22)) {
			last = (float)(lastValidLength / pathLength);
			return false;
		}
		Pose s = path.interpolate(length / pathLength);
		checks++;
		if (!is_state_valid(m, src, s.x, s.y, s.t, distance)) {
			last = (float)(lastValidLength / pathLength);
			return false;
		}
		lastValidLength = length;
		float distToMapBorder = fmin4(s.x - m.lbx, m.ubx - s.x, s.y - m.lby, m.uby - s.y);
		float deltaLength = distance - m.minSafeRadius;
		deltaLength = fminf(deltaLength, distToMapBorder);
		length += (double)fmaxf(deltaLength, m.minInterp);
	}
	last = 1.0f;
	return true;
}

/// HybridAStar::StatePropagator::GetVoronoiCost, algo/hybrid_a_star.cpp:93-109.
/// The reference overwrites (`=`, not `+=`) the cost at every sample, so only the LAST
/// sample survives (Appendix A Q8); it is then multiplied by the diagonal resolution (float).
/// The sample lengths are reproduced by the same repeated double += float accumulation, but
/// only the last one is interpolated and looked up (the earlier reads have no effect).
template <typename PathT>
PPD_INLINE double voronoi_cost(const MapView& m, const PathT& path, float interpLength, double voronoiCostMultiplier)
{
	float voronoiCost = 0.0f;
	const double pathLength = path.length;
	if (0.0 < pathLength) {
		double lastLength = 0.0;
		if (interpLength > 0.0f)
			for (double length = 0.0; length < pathLength; length += (double)interpLength) {
				if (length == lastLength && length != 0.0)
					break; // increment below one ulp of length: the reference would never terminate
				lastLength = length;
			}
		Pose p = path.interpolate(lastLength / pathLength);
		int row, col;
		world_to_cell(m, p.x, p.y, row, col);
		// the reference does not bounds-check (assert compiled out); clamp for memory safety
		row = min(max(row, 0), m.rows - 1);
		col = min(max(col, 0), m.cols - 1);
		voronoiCost = m.pathcost[(size_t)row * m.cols + col];
	}
	voronoiCost *= interpLength;
	return voronoiCostMultiplier * (double)voronoiCost;
}

/// voronoi_cost in two halves (see is_state_valid_issue): the map read, then the arithmetic on the value.
template <typename PathT>
PPD_INLINE void voronoi_cost_issue(const MapView& m, const PathT& path, float interpLength, float& raw)
{
	raw = 0.0f;
	const double pathLength = path.length;
	if (0.0 < pathLength) {
		double lastLength = 0.0;
		if (interpLength > 0.0f)
			for (double length = 0.0; length < pathLength; length += (double)interpLength) {
				if (length == lastLength && length != 0.0)
					break;
				lastLength = length;
			}
		Pose p = path.interpolate(lastLength / pathLength);
		int row, col;
		world_to_cell(m, p.x, p.y, row, col);
		row = min(max(row, 0), m.rows - 1);
		col = min(max(col, 0), m.cols - 1);
		raw = m.pathcost[(size_t)row * m.cols + col];
	}
}
PPD_INLINE double voronoi_cost_finish(float raw, float interpLength, double voronoiCostMultiplier)
{
	float voronoiCost = raw;
	voronoiCost *= interpLength;
	return voronoiCostMultiplier * (double)voronoiCost;
}

/// Pose2<int>::WrapTheta instantiated from the generic template (geometry/2dplane.h:36-45,
/// T = int): SURVEY Appendix A Q6.
PPD_INLINE int alias_heading_bin(int theta)
{
	int t = theta;
	while ((double)t > kPi)
		t = (int)((double)t - 2 * kPi);
	while ((double)t < -kPi)
		t = (int)((double)t + 2 * kPi);
	return t;
}

/// SURVEY 7.3 H2 (guard band): did DiscretizePose truncate a coordinate that lies within 1e-9 cells of a lattice boundary?  There a
/// last-bit difference between this libm and the reference's could pick the other cell; everywhere else the truncation cannot
/// differ (observed libm differences are ~1e-16 relative).  The search kernels count such poses per query
/// (pp_query_result::n_lattice_boundary_hits): the discrete outputs of a query are bit-exact BY CONSTRUCTION when the count is 0.
PPD_INLINE bool near_integer(double q) { return fabs(q - rint(q)) < 1e-9; }

/// HybridAStar::StatePropagator::DiscretizePose, algo/hybrid_a_star.h:104-111; returns the guard-band flag above
PPD_INLINE bool discretize_pose(const Pose& p, const Resolutions& r, int headingAlias, int& ix, int& iy, int& it)
{
	const double qx = div_by(p.x, r.spatial, r.invSpatial), qy = div_by(p.y, r.spatial, r.invSpatial), qt = div_by(wrap_theta(p.t), r.angular, r.invAngular);
	ix = trunc_to_int(qx);
	iy = trunc_to_int(qy);
	it = trunc_to_int(qt);
	if (headingAlias)
		it = alias_heading_bin(it);
	return near_integer(qx) | near_integer(qy) | near_integer(qt);
}
/// the count rides in the upper bits of the per-lane path-check counter (one reduction, one SuspendRec field for both)
constexpr int kGuardShift = 44;
constexpr long long kGuardMask = (1ll << kGuardShift) - 1;

} // namespace ppd
