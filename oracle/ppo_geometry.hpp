// TEST INFRASTRUCTURE ONLY -- CPU oracle for the MI355X planner core.
//
// This header is a plain-C++17 restatement (no Eigen, no flann, no spdlog) of the
// geometry layer of lfilipozzi/PathPlanning that the hot path uses.  It exists
// so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
// check / time the HIP path against the reference's algorithm.  Nothing in
// pathplanning_amd/ may include, link or call anything in oracle/.
//
// Every function cites the reference file:line it follows (paths relative to
// /root/reference/planner/src).  Arithmetic is kept expression-for-expression
// identical (types, operation order, float/double mixing) because bit-exact
// expansion order depends on it.
//
// Parity pinning: the Reeds-Shepp solver below is pinned by the reference's 48
// known-answer vectors (planner/tests/test_reeds_shepp.cpp:38-300, extracted to
// tests/golden/reeds_shepp_vectors.json).  The reference itself is unbuildable
// in this image (needs Eigen3 + flann, both absent; stand-in headers are not
// allowed), so functions without reference-held vectors are "parity unpinned"
// beyond the reference's own smoke tests -- see DESIGN.md section 3.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <limits>
#include <set>
#include <vector>

namespace ppo {

// ---------------------------------------------------------------------------
// geometry/2dplane.h
// ---------------------------------------------------------------------------
struct Point2d {
	double x = 0.0, y = 0.0;
};

// Eigen 2-vector ops the reference uses (geometry/2dplane.h:11-14):
// norm() = sqrt(x*x + y*y); Rotation2D(a) * v = (c*x - s*y, s*x + c*y).
inline double Norm(double x, double y) { return std::sqrt(x * x + y * y); }

inline double WrapThetaD(double theta)
{
	// geometry/2dplane.h:36-45 (generic Pose2<T>::WrapTheta, T = double)
	double t = theta;
	while (t > M_PI)
		t -= 2 * M_PI;
	while (t < -M_PI)
		t += 2 * M_PI;
	return t;
}

struct Pose2d {
	double x = 0.0, y = 0.0, theta = 0.0;
	Pose2d() = default;
	// geometry/2dplane.h:19-22: the constructors wrap theta; copy/assign do not.
	Pose2d(double x_, double y_, double th) :
		x(x_), y(y_), theta(WrapThetaD(th)) { }
	static Pose2d Raw(double x_, double y_, double th)
	{
		Pose2d p;
		p.x = x_;
		p.y = y_;
		p.theta = th;
		return p;
	}
	double WrapTheta() const { return WrapThetaD(theta); }
};

/// geometry/2dplane.h:49-62 -- SE(2) compose, `lhs + rhs`.
inline Pose2d Compose(const Pose2d& lhs, const Pose2d& rhs)
{
	const double s = std::sin(lhs.theta), c = std::cos(lhs.theta);
	Pose2d out(c * rhs.x + (-s) * rhs.y, s * rhs.x + c * rhs.y, lhs.theta + rhs.theta);
	out.x += lhs.x;
	out.y += lhs.y;
	out.theta = out.WrapTheta();
	return out;
}

/// geometry/2dplane.h:65-79 -- the pose p with rhs + p = lhs, `lhs - rhs`.
inline Pose2d Between(const Pose2d& lhs, const Pose2d& rhs)
{
	const double dx = lhs.x - rhs.x, dy = lhs.y - rhs.y;
	const double a = -rhs.theta;
	const double s = std::sin(a), c = std::cos(a);
	Pose2d out(c * dx + (-s) * dy, s * dx + c * dy, lhs.theta - rhs.theta);
	out.theta = out.WrapTheta();
	return out;
}

/// utils/maths.h:9-16
inline double Modulo(double in, double mod)
{
	double out = std::fmod(in, mod);
	if (out < 0)
		out += mod;
	return out;
}

// paths/path.h:10-20
enum class Steer : int { Left = 0,
	Straight = 1,
	Right = 2 };
enum class Direction : int { Forward = 0,
	Backward = 1,
	NoMotion = 2 };

// ---------------------------------------------------------------------------
// geometry/reeds_shepp.{h,cpp}
// ---------------------------------------------------------------------------
namespace rs {
	constexpr int kNumWords = 48; // geometry/reeds_shepp.h:14-46
	constexpr int kNumMotion = 5; // geometry/reeds_shepp.h:74
	constexpr double kInf = std::numeric_limits<double>::infinity();

	struct Motion { // geometry/reeds_shepp.h:52-63
		Steer steer = Steer::Left;
		Direction direction = Direction::NoMotion;
		double length = kInf;
		bool IsValid() const { return length != kInf && direction != Direction::NoMotion; } // reeds_shepp.cpp:425-428
	};

	struct PathSegment { // geometry/reeds_shepp.h:69-107
		std::array<Motion, kNumMotion> motions;
		double length = 0.0; // normalised (unit turning radius)

		void AddMotion(Steer steer, Direction direction, double len)
		{
			// reeds_shepp.cpp:439-456
			int index = -1;
			for (int i = 0; i < kNumMotion; i++) {
				if (!motions[i].IsValid()) {
					index = i;
					break;
				}
			}
			if (index < 0)
				return; // PP_ASSERT compiled out in the reference
			motions[index].steer = steer;
			motions[index].direction = direction;
			motions[index].length = len;
			length += std::abs(len);
		}

		int GetNumMotions() const
		{
			// reeds_shepp.cpp:458-467
			int index = 0;
			while (motions[index].IsValid()) {
				index++;
				if (index == kNumMotion)
					break;
			}
			return index;
		}

		double GetLength(double minTurningRadius) const { return length * minTurningRadius; }

		/// reeds_shepp.cpp:469-497.  Returns float; note the fast path ignores
		/// forwardCostMultiplier.
		float ComputeCost(double minTurningRadius, float reverseCostMultiplier, float forwardCostMultiplier, float directionSwitchingCost) const
		{
			if (!motions[0].IsValid())
				return std::numeric_limits<float>::infinity();
			if (reverseCostMultiplier == 1.0f && directionSwitchingCost == 0.0f)
				return length * minTurningRadius;
			float cost = 0;
			Direction prevDirection = motions[0].direction;
			for (const auto& motion : motions) {
				if (!motion.IsValid())
					break;
				float motionCost = motion.length * minTurningRadius;
				if (motion.direction == Direction::Forward)
					motionCost *= forwardCostMultiplier;
				else if (motion.direction == Direction::Backward)
					motionCost *= reverseCostMultiplier;
				if (motion.direction != prevDirection)
					motionCost += directionSwitchingCost;
				prevDirection = motion.direction;
				cost += motionCost;
			}
			return cost;
		}

		void Timeflip()
		{
			// reeds_shepp.cpp:499-508
			for (auto& m : motions) {
				if (m.direction == Direction::Backward)
					m.direction = Direction::Forward;
				else if (m.direction == Direction::Forward)
					m.direction = Direction::Backward;
			}
		}
		void Reflect()
		{
			// reeds_shepp.cpp:510-519
			for (auto& m : motions) {
				if (m.steer == Steer::Left)
					m.steer = Steer::Right;
				else if (m.steer == Steer::Right)
					m.steer = Steer::Left;
			}
		}
	};

	inline bool IsAngleInvalid(double theta) { return theta < 0 || theta > M_PI; } // reeds_shepp.cpp:11-14
	inline double WrapAngle(double theta) { return Modulo(theta + M_PI, 2 * M_PI) - M_PI; } // reeds_shepp.cpp:16-19

	// The twelve base-word formulas, reeds_shepp.cpp:21-297.  `family` = word / 4.
	inline double BaseLengths(int family, const Pose2d& goal, double& t, double& u, double& v)
	{
		const double gx = goal.x, gy = goal.y, gt = goal.theta;
		switch (family) {
		case 0: { // LfSfLf, reeds_shepp.cpp:21-35
			double x = gx - std::sin(gt);
			double y = gy - 1 + std::cos(gt);
			u = std::sqrt(x * x + y * y);
			t = std::atan2(y, x);
			v = WrapAngle(gt - t);
			if (IsAngleInvalid(t) || IsAngleInvalid(v))
				return kInf;
			return t + u + v;
		}
		case 1: { // LfSfRf, reeds_shepp.cpp:37-58.  The `u1squared < 4` test has
			// no `return` in the reference: NaN flows on (Appendix A Q10).
			double x = gx + std::sin(gt);
			double y = gy - 1 - std::cos(gt);
			double u1squared = x * x + y * y;
			double t1 = std::atan2(y, x);
			u = std::sqrt(u1squared - 4);
			double phi = std::atan2(2, u);
			t = WrapAngle(t1 + phi);
			v = WrapAngle(t - gt);
			if (IsAngleInvalid(t) || IsAngleInvalid(v))
				return kInf;
			return t + u + v;
		}
		case 2: { // LfRbLf, reeds_shepp.cpp:60-82
			double xi = gx - std::sin(gt);
			double eta = gy - 1 + std::cos(gt);
			double u1 = std::sqrt(xi * xi + eta * eta);
			if (u1 > 4)
				return kInf;
			double phi = std::atan2(eta, xi);
			double alpha = std::acos(u1 / 4.0);
			t = Modulo(M_PI_2 + alpha + phi, 2 * M_PI);
			u = Modulo(M_PI - 2 * alpha, 2 * M_PI);
			v = Modulo(gt - t - u, 2 * M_PI);
			if (IsAngleInvalid(t) || IsAngleInvalid(u) || IsAngleInvalid(v))
				return kInf;
			return t + u + v;
		}
		case 3: { // LfRbLb, reeds_shepp.cpp:84-103
			double xi = gx - std::sin(gt);
			double eta = gy - 1 + std::cos(gt);
			double u1 = std::sqrt(xi * xi + eta * eta);
			if (u1 > 4)
				return kInf;
			double phi = std::atan2(eta, xi);
			double alpha = std::acos(u1 / 4.0);
			t = Modulo(M_PI_2 + alpha + phi, 2 * M_PI);
			u = Modulo(M_PI - 2 * alpha, 2 * M_PI);
			v = Modulo(t + u - gt, 2 * M_PI);
			return t + u + v;
		}
		case 4: { // LfRfLb, reeds_shepp.cpp:105-125
			double xi = gx - std::sin(gt);
			double eta = gy - 1 + std::cos(gt);
			double u1 = std::sqrt(xi * xi + eta * eta);
			if (u1 > 4)
				return kInf;
			double phi = std::atan2(eta, xi);
			u = std::acos((8 - u1 * u1) / 8.0);
			double va = std::sin(u);
			double alpha = std::asin(2 * va / u1);
			t = Modulo(M_PI_2 - alpha + phi, 2 * M_PI);
			v = Modulo(t - u - gt, 2 * M_PI);
			return t + u + v;
		}
		case 5: { // LfRufLubRb, reeds_shepp.cpp:127-155
			double xi = gx + std::sin(gt);
			double eta = gy - 1 - std::cos(gt);
			double u1 = std::sqrt(xi * xi + eta * eta);
			if (u1 > 4)
				return kInf;
			double phi = std::atan2(eta, xi);
			if (u1 > 2) {
				double alpha = std::acos(u1 / 4 - 0.5);
				t = Modulo(M_PI_2 + phi - alpha, 2 * M_PI);
				u = Modulo(M_PI - alpha, 2 * M_PI);
				v = Modulo(gt - t + 2 * u, 2 * M_PI);
			} else {
				double alpha = std::acos(u1 / 4 + 0.5);
				t = Modulo(M_PI_2 + phi + alpha, 2 * M_PI);
				u = Modulo(alpha, 2 * M_PI);
				v = Modulo(gt - t + 2 * u, 2 * M_PI);
			}
			return t + u + u + v;
		}
		case 6: { // LfRubLubRf, reeds_shepp.cpp:157-182 (note the float literal 1.25f)
			double xi = gx + std::sin(gt);
			double eta = gy - 1 - std::cos(gt);
			double u1 = std::sqrt(xi * xi + eta * eta);
			if (u1 > 6)
				return kInf;
			double phi = std::atan2(eta, xi);
			double va1 = 1.25f - u1 * u1 / 16;
			if (va1 < 0 || va1 > 1)
				return kInf;
			u = std::acos(va1);
			double va2 = std::sin(u);
			double alpha = std::asin(2 * va2 / u1);
			t = Modulo(M_PI_2 + phi + alpha, 2 * M_PI);
			v = Modulo(t - gt, 2 * M_PI);
			return t + u + u + v;
		}
		case 7: { // LfRbpi2SbLb, reeds_shepp.cpp:184-208
			double xi = gx - std::sin(gt);
			double eta = gy - 1 + std::cos(gt);
			double u1squared = xi * xi + eta * eta;
			if (u1squared < 4)
				return kInf;
			double phi = std::atan2(eta, xi);
			u = std::sqrt(u1squared - 4) - 2;
			if (u < 0)
				return kInf;
			double alpha = std::atan2(2, u + 2);
			t = Modulo(M_PI_2 + phi + alpha, 2 * M_PI);
			v = Modulo(t + M_PI_2 - gt, 2 * M_PI);
			return t + M_PI_2 + u + v;
		}
		case 8: { // LfRbpi2SbRb, reeds_shepp.cpp:210-230
			double xi = gx + std::sin(gt);
			double eta = gy - 1 - std::cos(gt);
			double u1 = std::sqrt(xi * xi + eta * eta);
			if (u1 < 2)
				return kInf;
			double phi = std::atan2(eta, xi);
			t = Modulo(M_PI_2 + phi, 2 * M_PI);
			u = u1 - 2;
			v = Modulo(gt - t - M_PI_2, 2 * M_PI);
			return t + M_PI_2 + u + v;
		}
		case 9: { // LfSfRfpi2Lb, reeds_shepp.cpp:232-256
			double xi = gx - std::sin(gt);
			double eta = gy - 1 + std::cos(gt);
			double u1squared = xi * xi + eta * eta;
			if (u1squared < 4)
				return kInf;
			double phi = std::atan2(eta, xi);
			u = std::sqrt(u1squared - 4) - 2;
			if (u < 0)
				return kInf;
			double alpha = std::atan2(u + 2, 2);
			t = Modulo(M_PI_2 + phi - alpha, 2 * M_PI);
			v = Modulo(t - M_PI_2 - gt, 2 * M_PI);
			return t + u + M_PI_2 + v;
		}
		case 10: { // LfSfLfpi2Rb, reeds_shepp.cpp:258-278
			double xi = gx + std::sin(gt);
			double eta = gy - 1 - std::cos(gt);
			double u1 = std::sqrt(xi * xi + eta * eta);
			if (u1 < 2)
				return kInf;
			double phi = std::atan2(eta, xi);
			t = Modulo(phi, 2 * M_PI);
			u = u1 - 2;
			v = Modulo(-t - M_PI_2 + gt, 2 * M_PI);
			return t + u + M_PI_2 + v;
		}
		case 11: { // LfRbpi2SbLbpi2Rf, reeds_shepp.cpp:280-304
			double xi = gx + std::sin(gt);
			double eta = gy - 1 - std::cos(gt);
			double u1squared = xi * xi + eta * eta;
			if (u1squared < 16)
				return kInf;
			double phi = std::atan2(eta, xi);
			u = std::sqrt(u1squared - 4) - 4;
			if (u < 0)
				return kInf;
			double alpha = std::atan2(2, u + 4);
			t = Modulo(M_PI_2 + phi + alpha, 2 * M_PI);
			v = Modulo(t - gt, 2 * M_PI);
			return t + u + v + M_PI;
		}
		default:
			t = u = v = 0;
			return kInf;
		}
	}

	/// reeds_shepp.cpp:306-414 (the twelve *Path builders) + 566-606 (GetPath).
	inline PathSegment GetPath(int word, double t, double u, double v)
	{
		using S = Steer;
		using D = Direction;
		PathSegment p;
		switch (word / 4) {
		case 0: p.AddMotion(S::Left, D::Forward, t); p.AddMotion(S::Straight, D::Forward, u); p.AddMotion(S::Left, D::Forward, v); break;
		case 1: p.AddMotion(S::Left, D::Forward, t); p.AddMotion(S::Straight, D::Forward, u); p.AddMotion(S::Right, D::Forward, v); break;
		case 2: p.AddMotion(S::Left, D::Forward, t); p.AddMotion(S::Right, D::Backward, u); p.AddMotion(S::Left, D::Forward, v); break;
		case 3: p.AddMotion(S::Left, D::Forward, t); p.AddMotion(S::Right, D::Backward, u); p.AddMotion(S::Left, D::Backward, v); break;
		case 4: p.AddMotion(S::Left, D::Forward, t); p.AddMotion(S::Right, D::Forward, u); p.AddMotion(S::Left, D::Backward, v); break;
		case 5: p.AddMotion(S::Left, D::Forward, t); p.AddMotion(S::Right, D::Forward, u); p.AddMotion(S::Left, D::Backward, u); p.AddMotion(S::Right, D::Backward, v); break;
		case 6: p.AddMotion(S::Left, D::Forward, t); p.AddMotion(S::Right, D::Backward, u); p.AddMotion(S::Left, D::Backward, u); p.AddMotion(S::Right, D::Forward, v); break;
		case 7: p.AddMotion(S::Left, D::Forward, t); p.AddMotion(S::Right, D::Backward, M_PI_2); p.AddMotion(S::Straight, D::Backward, u); p.AddMotion(S::Left, D::Backward, v); break;
		case 8: p.AddMotion(S::Left, D::Forward, t); p.AddMotion(S::Right, D::Backward, M_PI_2); p.AddMotion(S::Straight, D::Backward, u); p.AddMotion(S::Right, D::Backward, v); break;
		case 9: p.AddMotion(S::Left, D::Forward, t); p.AddMotion(S::Straight, D::Forward, u); p.AddMotion(S::Right, D::Forward, M_PI_2); p.AddMotion(S::Left, D::Backward, v); break;
		case 10: p.AddMotion(S::Left, D::Forward, t); p.AddMotion(S::Straight, D::Forward, u); p.AddMotion(S::Left, D::Forward, M_PI_2); p.AddMotion(S::Right, D::Backward, v); break;
		case 11: p.AddMotion(S::Left, D::Forward, t); p.AddMotion(S::Right, D::Backward, M_PI_2); p.AddMotion(S::Straight, D::Backward, u); p.AddMotion(S::Left, D::Backward, M_PI_2); p.AddMotion(S::Right, D::Forward, v); break;
		default: return PathSegment();
		}
		switch (word % 4) { // reeds_shepp.cpp:592-604
		case 1: p.Timeflip(); break;
		case 2: p.Reflect(); break;
		case 3: p.Timeflip(); p.Reflect(); break;
		default: break;
		}
		return p;
	}

	/// reeds_shepp.cpp:521-535
	inline std::array<Pose2d, 4> GetGoalArray(const Pose2d& start, const Pose2d& goal, double minTurningRadius)
	{
		Pose2d ng = Between(goal, start);
		ng.x = ng.x / minTurningRadius;
		ng.y = ng.y / minTurningRadius;
		return { Pose2d(ng.x, ng.y, ng.theta), Pose2d(-ng.x, ng.y, -ng.theta), Pose2d(ng.x, -ng.y, -ng.theta), Pose2d(-ng.x, -ng.y, ng.theta) };
	}

	/// reeds_shepp.cpp:537-564
	inline double GetMotionLengths(int word, const std::array<Pose2d, 4>& goals, double& t, double& u, double& v)
	{
		return BaseLengths(word / 4, goals[word % 4], t, u, v);
	}

	/// reeds_shepp.cpp:608-637.  Returns the length; word = -1 if none.
	inline double GetShortestDistance(const Pose2d& start, const Pose2d& goal, double minTurningRadius, int* wordOut, double* tuvOut)
	{
		const auto goals = GetGoalArray(start, goal, minTurningRadius);
		double smallestLength = kInf;
		int smallestWord = -1;
		double t = 0, u = 0, v = 0;
		for (int w = 0; w < kNumWords; w++) {
			double tt, uu, vv;
			double length = GetMotionLengths(w, goals, tt, uu, vv);
			if (length < smallestLength) {
				smallestLength = length;
				smallestWord = w;
				t = tt;
				u = uu;
				v = vv;
			}
		}
		if (wordOut)
			*wordOut = smallestWord;
		if (tuvOut) {
			tuvOut[0] = t;
			tuvOut[1] = u;
			tuvOut[2] = v;
		}
		return smallestLength;
	}

	/// reeds_shepp.cpp:639-652
	inline PathSegment GetShortestPath(const Pose2d& start, const Pose2d& goal, double minTurningRadius, int* wordOut)
	{
		double tuv[3];
		int w = -1;
		GetShortestDistance(start, goal, minTurningRadius, &w, tuv);
		if (w < 0 || w >= kNumWords)
			return PathSegment();
		if (wordOut)
			*wordOut = w;
		return GetPath(w, tuv[0], tuv[1], tuv[2]);
	}

	/// reeds_shepp.cpp:654-683.  Costs compared in float, first strictly lower wins.
	inline PathSegment GetOptimalPath(const Pose2d& start, const Pose2d& goal, double minTurningRadius, float reverseCostMultiplier,
		float forwardCostMultiplier, float directionSwitchingCost, int* wordOut = nullptr, double* tuvOut = nullptr)
	{
		const auto goals = GetGoalArray(start, goal, minTurningRadius);
		float optimalCost = std::numeric_limits<float>::infinity();
		int optimalWord = -1;
		PathSegment optimalPath;
		for (int w = 0; w < kNumWords; w++) {
			double t, u, v;
			double length = GetMotionLengths(w, goals, t, u, v);
			if (length == kInf)
				continue;
			PathSegment path = GetPath(w, t, u, v);
			float cost = path.ComputeCost(minTurningRadius, reverseCostMultiplier, forwardCostMultiplier, directionSwitchingCost);
			if (cost < optimalCost) {
				optimalCost = cost;
				optimalWord = w;
				optimalPath = path;
				if (tuvOut) {
					tuvOut[0] = t;
					tuvOut[1] = u;
					tuvOut[2] = v;
				}
			}
		}
		if (wordOut)
			*wordOut = optimalWord;
		if (optimalWord < 0 || optimalWord >= kNumWords)
			return PathSegment();
		return optimalPath;
	}
} // namespace rs

// ---------------------------------------------------------------------------
// models/kinematic_bicycle_model.{h,cpp}
// ---------------------------------------------------------------------------
struct KinematicBicycleModel {
	double wheelbase = 2.6, rearToCenter = 0.0;

	/// models/kinematic_bicycle_model.cpp:5-32.  theta is NOT wrapped (Q12).
	Pose2d ConstantSteer(const Pose2d& from, double steering, double dist, Direction direction) const
	{
		if (direction == Direction::Backward)
			dist = -dist;
		Pose2d to = from;
		double tanSteering = std::tan(steering);
		double beta = std::atan(rearToCenter * tanSteering / wheelbase);
		double cosBeta = std::cos(beta);
		double DthetaDdist = cosBeta * tanSteering / wheelbase;
		dist = dist / cosBeta;
		if (std::abs(DthetaDdist) > 1e-9) {
			to.theta += dist * DthetaDdist;
			to.x += 1 / DthetaDdist * (std::sin(beta + to.theta) - std::sin(beta + from.theta));
			to.y += 1 / DthetaDdist * (-std::cos(beta + to.theta) + std::cos(beta + from.theta));
		} else {
			to.x += dist * std::cos(from.theta);
			to.y += dist * std::sin(from.theta);
		}
		return to;
	}

	/// models/kinematic_bicycle_model.cpp:34-41
	double GetSteeringAngleFromTurningRadius(double radius) const
	{
		if (radius < rearToCenter)
			return M_PI_2;
		return std::atan(wheelbase / std::sqrt(std::pow(radius, 2) - std::pow(rearToCenter, 2)));
	}
};

// ---------------------------------------------------------------------------
// paths/*.{h,cpp}
// ---------------------------------------------------------------------------
struct PathSE2Base { // paths/path.h:22-52 + 73-90 (PathNonHolonomic)
	Pose2d init, final;
	double length = 0.0;
	virtual ~PathSE2Base() = default;
	virtual Pose2d Interpolate(double ratio) const = 0;
	virtual void Truncate(double ratio) = 0;
	virtual Direction GetDirection(double ratio) const = 0;
	virtual std::set<double> GetCuspPointRatios() const { return {}; }
};

struct PathConstantSteer : PathSE2Base { // paths/path_constant_steer.{h,cpp}
	const KinematicBicycleModel* model;
	double steering;
	Direction direction;
	PathConstantSteer(const KinematicBicycleModel* m, const Pose2d& from, double steer, double len, Direction dir) :
		model(m), steering(steer), direction(dir)
	{
		init = from;
		length = len;
		final = Interpolate(1.0); // path_constant_steer.cpp:8
	}
	Pose2d Interpolate(double ratio) const override { return model->ConstantSteer(init, steering, length * ratio, direction); } // :11-14
	void Truncate(double ratio) override
	{
		// :16-20
		final = Interpolate(ratio);
		length *= ratio;
	}
	Direction GetDirection(double) const override { return direction; }
};

struct PathReedsShepp : PathSE2Base { // paths/path_reeds_shepp.{h,cpp}
	rs::PathSegment segment;
	double minTurningRadius = 1;

	PathReedsShepp(const Pose2d& from, const rs::PathSegment& seg, double rmin) :
		segment(seg), minTurningRadius(rmin)
	{
		init = from;
		length = seg.GetLength(rmin);
		final = Interpolate(1.0); // path_reeds_shepp.cpp:9
	}

	Pose2d Straight(const Pose2d& start, Direction direction, double len) const
	{
		// path_reeds_shepp.cpp:123-135
		if (direction == Direction::Backward)
			len = -len;
		len *= minTurningRadius;
		return Pose2d(start.x + len * std::cos(start.theta), start.y + len * std::sin(start.theta), start.theta);
	}

	Pose2d Turn(const Pose2d& start, Direction direction, Steer steer, double turnAngle) const
	{
		// path_reeds_shepp.cpp:137-153
		if (direction == Direction::Backward)
			turnAngle = -turnAngle;
		double phi = turnAngle / 2;
		double cosPhi = std::cos(phi);
		double sinPhi = std::sin(phi);
		double L = 2 * sinPhi * minTurningRadius;
		double x = L * cosPhi;
		double y = L * sinPhi;
		if (steer == Steer::Right) {
			y *= -1;
			turnAngle *= -1;
		}
		return Compose(start, Pose2d(x, y, turnAngle));
	}

	Pose2d Interpolate(double ratio) const override
	{
		// path_reeds_shepp.cpp:12-47
		const double totalLength = length;
		if (totalLength == 0)
			return init;
		Pose2d interp = init;
		double len = 0;
		for (const auto& motion : segment.motions) {
			if (!motion.IsValid())
				break;
			double motionLength = motion.length * minTurningRadius;
			if (motionLength == 0)
				continue;
			double motionRatio = (ratio * totalLength - len) / motionLength;
			motionRatio = std::min(motionRatio, 1.0);
			switch (motion.steer) {
			case Steer::Straight: interp = Straight(interp, motion.direction, motion.length * motionRatio); break;
			case Steer::Left: interp = Turn(interp, motion.direction, motion.steer, motion.length * motionRatio); break;
			case Steer::Right: interp = Turn(interp, motion.direction, motion.steer, motion.length * motionRatio); break;
			}
			len += motionLength;
			if (len >= ratio * totalLength)
				break;
		}
		return interp;
	}

	void Truncate(double ratio) override
	{
		// path_reeds_shepp.cpp:49-93, including the wrong-index reset (Q11):
		// motions[i] (not [ii]) is invalidated when i < 4.
		const double totalLength = length;
		if (totalLength == 0)
			final = init;
		else {
			Pose2d interp = init;
			double len = 0;
			for (int i = 0; i < rs::kNumMotion; i++) {
				const auto& motion = segment.motions[i];
				if (!motion.IsValid())
					break;
				double motionLength = motion.length * minTurningRadius;
				if (motionLength == 0)
					continue;
				double motionRatio = (ratio * totalLength - len) / motionLength;
				motionRatio = std::min(motionRatio, 1.0);
				switch (motion.steer) {
				case Steer::Straight: interp = Straight(interp, motion.direction, motion.length * motionRatio); break;
				case Steer::Left: interp = Turn(interp, motion.direction, motion.steer, motion.length * motionRatio); break;
				case Steer::Right: interp = Turn(interp, motion.direction, motion.steer, motion.length * motionRatio); break;
				}
				len += motionLength;
				if (len >= ratio * totalLength) {
					segment.motions[i].length *= motionRatio;
					for (int ii = i + 1; ii < rs::kNumMotion; ii++)
						segment.motions[i] = rs::Motion();
					break;
				}
			}
			final = interp;
		}
		length *= ratio;
	}

	std::set<double> GetCuspPointRatios() const override
	{
		// path_reeds_shepp.cpp:95-121
		std::set<double> ratios;
		if (length == 0.0)
			return ratios;
		double len = segment.motions[0].length * minTurningRadius;
		for (int i = 1; i < segment.GetNumMotions(); i++) {
			const auto& curr = segment.motions[i];
			const auto& prev = segment.motions[i - 1];
			double ratio = len / length;
			if (ratio > 1.0)
				break;
			if (curr.direction != prev.direction)
				ratios.insert(ratio);
			len += curr.length * minTurningRadius;
		}
		return ratios;
	}

	Direction GetDirection(double ratio) const override
	{
		// path_reeds_shepp.cpp:155-167
		if (length == 0)
			return Direction::NoMotion;
		double len = 0;
		for (const auto& motion : segment.motions) {
			if (!motion.IsValid())
				break;
			len += motion.length * minTurningRadius;
			if (ratio * length <= len)
				return motion.direction;
		}
		return Direction::NoMotion;
	}

	/// path_reeds_shepp.cpp:169-172
	double ComputeCost(double directionSwitchingCost, double reverseCostMultiplier, double forwardCostMultiplier) const
	{
		return segment.ComputeCost(minTurningRadius, reverseCostMultiplier, forwardCostMultiplier, directionSwitchingCost);
	}
};

struct PathR2 { // paths/path_r2.{h,cpp}
	Point2d init, final;
	double length = 0.0;
	PathR2(const Point2d& from, const Point2d& to)
	{
		init = from;
		length = Norm(to.x - from.x, to.y - from.y); // path_r2.cpp:5-9
		final = to;
	}
	Point2d Interpolate(double ratio) const
	{
		// path_r2.cpp:11-16: (1 - ratio) * init + ratio * final
		Point2d p;
		p.x = (1 - ratio) * init.x + ratio * final.x;
		p.y = (1 - ratio) * init.y + ratio * final.y;
		return p;
	}
	void Truncate(double ratio)
	{
		// path_r2.cpp:18-22
		final = Interpolate(ratio);
		length *= ratio;
	}
};

} // namespace ppo
