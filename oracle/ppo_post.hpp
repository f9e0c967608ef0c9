// TEST INFRASTRUCTURE ONLY -- CPU restatement of what the reference does AFTER the graph search:
//   paths/path_composite.{h,cpp}          PathComposite (PushBack, FindSegment, Interpolate), PathSE2CompositeNonHolonomic
//   algo/hybrid_a_star.cpp:175-184        GraphSearch::GetCompositePath
//   algo/hybrid_a_star.cpp:260-291        resampling at `pathInterpolation` with cusp snapping
//   algo/smoother.cpp:10-226              Smoother::Smooth, CalculateCurvatureTerm, IsPathSafe (Eigen 2-vector expressions
//                                         written out per coefficient; float / double mix as in the source)
// Parity unpinned beyond the source text: the reference has no test or fixture for this stage, and it cannot be built here.
#pragma once

#include <algorithm>
#include <cmath>
#include <memory>
#include <set>
#include <unordered_set>
#include <vector>

#include "ppo_geometry.hpp"
#include "ppo_world.hpp"

namespace ppo {

/// paths/path_composite.h:8-90 + path_composite.cpp
struct PathComposite {
	struct Part {
		double initLength, finalLength;
		std::shared_ptr<PathSE2Base> path;
	};
	std::vector<Part> parts;
	Pose2d final;
	double length = 0.0;

	void PushBack(const std::shared_ptr<PathSE2Base>& path)
	{
		double pathLength = path->length;
		parts.push_back({ length, length + pathLength, path });
		final = path->final;
		length += pathLength;
	}
	/// FindSegment, path_composite.h:66-82: first part whose finalLength exceeds ratio * length
	std::pair<size_t, double> FindSegment(double ratio) const
	{
		double len = ratio * length;
		size_t it = 0;
		{
			size_t lo = 0, hi = parts.size(); // std::upper_bound with comp(len, part) = len < part.finalLength
			while (lo < hi) {
				size_t mid = lo + (hi - lo) / 2;
				if (len < parts[mid].finalLength)
					hi = mid;
				else
					lo = mid + 1;
			}
			it = lo;
		}
		if (it == parts.size())
			return { it, 1.0 };
		double pathLength = parts[it].path->length;
		if (pathLength == 0.0)
			return { it, 0.0 };
		return { it, (len - parts[it].initLength) / pathLength };
	}
	Pose2d Interpolate(double ratio) const
	{
		auto [it, pathRatio] = FindSegment(ratio);
		if (it == parts.size())
			it--;
		return parts[it].path->Interpolate(pathRatio);
	}
	/// path_composite.cpp:4-27 -- note: the cusp ratios of a sub-path are inserted as they are, i.e. as ratios of THAT
	/// sub-path, not of the composite (the reference does not rescale them)
	std::set<double> GetCuspPointRatios() const
	{
		std::set<double> ratios;
		if (parts.empty())
			return ratios;
		double len = 0.0;
		Direction prevDirection = parts[0].path->GetDirection(0.0);
		for (const auto& p : parts) {
			if (p.path->GetDirection(0.0) != prevDirection)
				ratios.insert(len / length);
			auto cusps = p.path->GetCuspPointRatios();
			ratios.insert(cusps.begin(), cusps.end());
			prevDirection = p.path->GetDirection(1.0);
			len += p.path->length;
		}
		return ratios;
	}
};

/// hybrid_a_star.cpp:260-291: ratios at which the composite path is sampled, and which samples are cusp points
inline void ResampleRatios(const PathComposite& path, float pathInterpolation, std::vector<double>& ratios, std::unordered_set<int>& cuspIndices)
{
	const double pathLength = path.length;
	auto cuspRatios = path.GetCuspPointRatios();
	cuspRatios.insert(0.0);
	cuspRatios.insert(1.0);
	auto cuspRatioIt = cuspRatios.begin();
	for (double length = 0.0; length <= pathLength; length += pathInterpolation) {
		double cuspLength = *cuspRatioIt * pathLength;
		if (length >= cuspLength - pathInterpolation / 2 && length < cuspLength + pathInterpolation / 2) {
			cuspIndices.insert((int)ratios.size());
			ratios.push_back(*cuspRatioIt);
			cuspRatioIt++;
		} else {
			ratios.push_back(length / pathLength);
		}
	}
}

/// algo/smoother.h:20-60
struct SmootherParameters {
	float stepTolerance = 1e-3;
	int maxIterations = 2000;
	float learningRate = 0.01f;
	float pathWeight = 0.0f;
	float smoothWeight = 0.4f;
	float voronoiWeight = 0.02f;
	float collisionWeight = 0.2f;
	float curvatureWeight = 0.4f;
	float collisionRatio = 0.2f;
	float maxCurvature = 0.5f;
};
enum SmootherStatus { SmMaxIteration = 0, SmStepTolerance = 1, SmPathSize = 2, SmFailure = -1, SmCollision = -2 };

struct V2 { // Eigen::Vector2d
	double x = 0.0, y = 0.0;
	double dot(const V2& o) const { return x * o.x + y * o.y; }
	double squaredNorm() const { return x * x + y * y; }
	double norm() const { return std::sqrt(squaredNorm()); }
	V2 normalized() const
	{
		double n = norm();
		return { x / n, y / n };
	}
};

struct Smoother {
	const World* world;
	SmootherParameters p;
	// nearest obstacle cell / nearest Voronoi-edge cell per map cell (GVD::GetNearestObstacleCell / GetNearestVoronoiEdgeCell)
	const Grid<Cell>* nearestObstacle;
	const Grid<Cell>* nearestEdge;
	std::vector<Pose2d> current;
	int iterations = 0;

	static V2 OrthogonalComplement(const V2& a, const V2& b)
	{
		// smoother.cpp:10-13: a - a.dot(b) * b / b.squaredNorm()
		double d = a.dot(b), sq = b.squaredNorm();
		return { a.x - d * b.x / sq, a.y - d * b.y / sq };
	}
	bool IsPathSafe() const
	{
		for (const auto& pose : current)
			if (!world->IsStateValid(pose))
				return false;
		return true;
	}
	/// 0: the restatement as it is.  +1 / -1: every cosine of the curvature term is moved one ulp up / down (|k| >= 2: up or down per
	/// call, pseudo-randomly, a different sequence per k) -- what another libm
	/// build (glibc selects FMA or SSE2 variants of cos per CPU) may legitimately return.  A query whose smoothed path moves by
	/// more than the parity tolerance under this probe has no machine-independent reference result.
	static int& LibmLastBit()
	{
		static int v = 0;
		return v;
	}
	static uint64_t& ProbeCounter()
	{
		static uint64_t c = 0;
		return c;
	}
	void CurvatureTerm(const V2& xim1, const V2& xi, const V2& xip1, V2& gim1, V2& gi, V2& gip1) const
	{
		// smoother.cpp:160-214
		V2 deltaXi { xi.x - xim1.x, xi.y - xim1.y };
		V2 deltaXip1 { xip1.x - xi.x, xip1.y - xi.y };
		// The reference writes UNQUALIFIED acos / cos / sqrt on floats here.  Its translation unit sees <cmath> (through Eigen) but
		// neither <math.h> nor a using-directive, so only the C library's ::acos(double), ::cos(double), ::sqrt(double) are
		// visible unqualified (std::acos(float) is not; no ADL for a fundamental type): the float argument is promoted, the
		// function runs in DOUBLE and the result is narrowed on assignment.  (Checked with g++ 11 on a probe that includes what
		// Eigen/Core includes: decltype(acos(1.0f)) is double.)  So 1 - cos^2 below is formed in double -- no cancellation at
		// float level.  Rounds 1-2 restated these as std::acos / std::cos (the float overloads), which was wrong.
		float deltaPhi = ::acos((double)std::clamp<float>((deltaXi.normalized().dot(deltaXip1.normalized())), -1.0f, 1.0f));
		float kappa = deltaPhi / deltaXi.norm();
		if (kappa <= p.maxCurvature)
			return;
		float denominator = deltaXi.norm() * deltaXip1.norm();
		V2 oc1 = OrthogonalComplement(deltaXip1, deltaXi), oc2 = OrthogonalComplement(deltaXi, deltaXip1);
		V2 DcosDeltaPhi_DdeltaXi { oc1.x / denominator, oc1.y / denominator };
		V2 DcosDeltaPhi_DdeltaXip1 { oc2.x / denominator, oc2.y / denominator };
		double cosDeltaPhi = ::cos((double)deltaPhi);
		if (LibmLastBit() != 0) { // sensitivity probe (tests only): the cosine moved by one unit in the last place
			bool up = LibmLastBit() > 0;
			if (LibmLastBit() > 1 || LibmLastBit() < -1) { // |k| >= 2: up or down per call, a fixed pseudo-random sequence per k
				uint64_t z = (ProbeCounter()++ + (uint64_t)(int64_t)LibmLastBit()) * 0x9E3779B97F4A7C15ull;
				z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
				z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
				up = ((z ^ (z >> 31)) & 1ull) != 0;
			}
			cosDeltaPhi = std::nextafter(cosDeltaPhi, up ? 2.0 : -2.0);
		}
		float DdeltaPhi_DcosDeltaPhi = -1.0f / ::sqrt(1.0f - std::pow(cosDeltaPhi, 2));
		float coef1 = 1 / deltaXi.norm() * DdeltaPhi_DcosDeltaPhi;
		V2 nrm = deltaXi.normalized();
		double c2 = deltaPhi / deltaXi.squaredNorm();
		V2 coef2 { c2 * nrm.x, c2 * nrm.y };
		V2 Dxim1 { -DcosDeltaPhi_DdeltaXi.x, -DcosDeltaPhi_DdeltaXi.y };
		V2 Dxi { DcosDeltaPhi_DdeltaXi.x - DcosDeltaPhi_DdeltaXip1.x, DcosDeltaPhi_DdeltaXi.y - DcosDeltaPhi_DdeltaXip1.y };
		V2 Dxip1 = DcosDeltaPhi_DdeltaXip1;
		V2 Dk_im1 { coef1 * Dxim1.x + coef2.x, coef1 * Dxim1.y + coef2.y };
		V2 Dk_i { coef1 * Dxi.x - coef2.x, coef1 * Dxi.y - coef2.y };
		V2 Dk_ip1 { coef1 * Dxip1.x, coef1 * Dxip1.y };
		float w = p.curvatureWeight * (kappa - p.maxCurvature);
		gim1.x += w * Dk_im1.x, gim1.y += w * Dk_im1.y;
		gi.x += w * Dk_i.x, gi.y += w * Dk_i.y;
		gip1.x += w * Dk_ip1.x, gip1.y += w * Dk_ip1.y;
	}
	/// smoother.cpp:33-158
	int Smooth(const std::vector<Pose2d>& path, const std::unordered_set<int>& cuspIndices)
	{
		int status = SmFailure;
		iterations = 0;
		if (path.size() < 5) {
			current = path;
			if (!IsPathSafe())
				return SmFailure;
			return SmPathSize;
		}
		const int num = (int)path.size();
		std::vector<int> indices;
		for (int i = 0; i < num - 4; i++) {
			if (cuspIndices.count(i + 4)) {
				i += 3;
				continue;
			}
			if (cuspIndices.count(i + 3)) {
				i += 2;
				continue;
			}
			if (cuspIndices.count(i + 2)) {
				i += 1;
				continue;
			}
			if (cuspIndices.count(i + 1))
				continue;
			if (cuspIndices.count(i))
				continue;
			indices.push_back(i + 2);
		}
		current = path;
		if (LibmLastBit() > 1 || LibmLastBit() < -1) { // sensitivity probe: the sampled points come out of sin / cos (arc and Reeds-Shepp
			// interpolation), whose last bit is libm's: every coordinate one ulp up or down, pseudo-randomly
			for (auto& pose : current) {
				uint64_t z = (ProbeCounter()++ + 77u * (uint64_t)(int64_t)LibmLastBit()) * 0x9E3779B97F4A7C15ull;
				z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
				z ^= z >> 27;
				pose.x = std::nextafter(pose.x, (z & 1ull) ? 1e300 : -1e300);
				pose.y = std::nextafter(pose.y, (z & 2ull) ? 1e300 : -1e300);
			}
		}
		std::vector<V2> gradients((size_t)num);
		const float unsafeRadius = world->minSafeRadius * (1 + p.collisionRatio);
		int count = -1;
		float step = p.stepTolerance;
		while (true) {
			count++;
			if (count >= p.maxIterations) {
				status = SmMaxIteration;
				break;
			}
			if (step < p.stepTolerance) {
				status = SmStepTolerance;
				break;
			}
			std::fill(gradients.begin(), gradients.end(), V2 {});
			for (int i : indices) {
				const V2 curr { current[i].x, current[i].y };
				gradients[i].x += p.pathWeight * (path[i].x - curr.x);
				gradients[i].y += p.pathWeight * (path[i].y - curr.y);
				Cell cc = world->WorldPositionToGridCell(curr.x, curr.y, false); // OccupancyMap::IsInsideMap(const Point2d&), occupancy_map.h
				if (world->IsInsideMap(cc)) {
					Cell cell = world->WorldPositionToGridCell(curr.x, curr.y); // bounded
					Point2d ow = world->GridCellToWorldPosition(nearestObstacle->at(cell));
					V2 dirObs { curr.x - ow.x, curr.y - ow.y };
					float obstDist = dirObs.norm();
					if (obstDist < unsafeRadius) {
						float s = p.collisionWeight * (obstDist - unsafeRadius);
						gradients[i].x += s * dirObs.x / obstDist;
						gradients[i].y += s * dirObs.y / obstDist;
					}
					const float alpha = world->alpha, dMax = world->dMax;
					if (obstDist < dMax && p.voronoiWeight > 0.0f) {
						Point2d vw = world->GridCellToWorldPosition(nearestEdge->at(cell));
						V2 dirVoro { curr.x - vw.x, curr.y - vw.y };
						float voroDist = dirVoro.norm();
						if (voroDist > 0.0f) {
							float alphaPlusObstDist = alpha + obstDist;
							float obstDistMinusDMax = obstDist - dMax;
							float obstDistPlusVoroDist = obstDist + voroDist;
							float dMaxSquared = dMax * dMax;
							float pvdv = (alpha / alphaPlusObstDist) * (obstDistMinusDMax * obstDistMinusDMax / dMaxSquared) * (obstDist / (obstDistPlusVoroDist * obstDistPlusVoroDist));
							float pvdo = (alpha / alphaPlusObstDist) * (voroDist / obstDistPlusVoroDist) * (obstDistMinusDMax / dMaxSquared)
								* (-obstDistMinusDMax / alphaPlusObstDist - obstDistMinusDMax / obstDistPlusVoroDist + 2);
							gradients[i].x += p.voronoiWeight * (pvdo * dirObs.x / obstDist + pvdv * dirVoro.x / voroDist);
							gradients[i].y += p.voronoiWeight * (pvdo * dirObs.y / obstDist + pvdv * dirVoro.y / voroDist);
						}
					}
				}
				gradients[i].x += p.smoothWeight * (current[i - 2].x - 4 * current[i - 1].x + 6 * curr.x - 4 * current[i + 1].x + current[i + 2].x);
				gradients[i].y += p.smoothWeight * (current[i - 2].y - 4 * current[i - 1].y + 6 * curr.y - 4 * current[i + 1].y + current[i + 2].y);
				CurvatureTerm({ current[i - 1].x, current[i - 1].y }, { current[i].x, current[i].y }, { current[i + 1].x, current[i + 1].y }, gradients[i - 1], gradients[i],
					gradients[i + 1]);
			}
			step = 0.0;
			for (int i = 0; i < num; i++) {
				current[i].x = current[i].x - p.learningRate * gradients[i].x;
				current[i].y = current[i].y - p.learningRate * gradients[i].y;
				step = std::max<float>(step, gradients[i].norm());
			}
		}
		iterations = count;
		if (!IsPathSafe())
			status = SmFailure;
		return status;
	}
};

} // namespace ppo
