// Path value types of the plugin surface (the reference's paths/ directory and models/kinematic_bicycle_model.*):
//   paths/path.h:22-90                    Path<State>, PathConnection<State>, PathNonHolonomic<State>
//   paths/path_r2.cpp, paths/path_se2.cpp PathR2, PathSE2 (+ connections): linear, plain arithmetic
//   paths/path_constant_steer.cpp         PathConstantSteer over KinematicBicycleModel::ConstantSteer (closed form)
//   paths/path_reeds_shepp.{h,cpp}        PathReedsShepp, PathConnectionReedsShepp
// The Reeds-Shepp path is held as the ABI's pp_rs_path record and every operation on it (Connect, Interpolate, Truncate with
// the reference's Q11 slot reset, GetDirection, GetCuspPointRatios) runs on the GPU through include/pp_hip.h -- one source of
// that arithmetic, the one the search kernels inline.  Validation of any path is the validator's job (planner_hip.hpp).
#pragma once

#include <limits>
#include <set>

#include "types.hpp"

namespace Planner {

/// paths/path.h:22-52
template <typename State>
class Path {
public:
	Path() = default;
	Path(State init, double length = 0.0) : m_init(init), m_length(length) { }
	virtual ~Path() = default;
	const State& GetInitialState() const { return m_init; }
	const State& GetFinalState() const { return m_final; }
	virtual State Interpolate(double ratio) const = 0;
	std::vector<State> Interpolate(const std::vector<double>& ratios) const
	{
		std::vector<State> states;
		states.reserve(ratios.size());
		for (double r : ratios)
			states.push_back(Interpolate(r));
		return states;
	}
	virtual void Truncate(double ratio) = 0;
	double GetLength() const { return m_length; }

protected:
	State m_init, m_final;
	double m_length = 0.0;
};

/// paths/path.h:54-62
template <typename State>
class PathConnection {
public:
	virtual ~PathConnection() = default;
	virtual Ref<Path<State>> Connect(const State& from, const State& to) = 0;
};

using PathR2Base = Path<Point2d>;
using PathSE2Base = Path<Pose2d>;
using PathConnectionR2Base = PathConnection<Point2d>;
using PathConnectionSE2Base = PathConnection<Pose2d>;

/// paths/path.h:73-90
template <typename State>
class PathNonHolonomic : public Path<State> {
public:
	PathNonHolonomic() = default;
	PathNonHolonomic(Pose2d init, double length = 0.0) : Path<State>(init, length) { }
	virtual Direction GetDirection(double ratio) const = 0;
	virtual std::set<double> GetCuspPointRatios() const { return {}; }
};
using PathNonHolonomicSE2Base = PathNonHolonomic<Pose2d>;

/// paths/path_r2.cpp
class PathR2 : public PathR2Base {
public:
	PathR2(const Point2d& from, const Point2d& to) : PathR2Base(from, (to - from).norm()) { m_final = to; }
	Point2d Interpolate(double ratio) const override
	{
		return { (1 - ratio) * m_init.x() + ratio * m_final.x(), (1 - ratio) * m_init.y() + ratio * m_final.y() };
	}
	void Truncate(double ratio) override
	{
		m_final = Interpolate(ratio);
		m_length *= ratio;
	}
};
class PathConnectionR2 : public PathConnectionR2Base {
public:
	Ref<PathR2Base> Connect(const Point2d& from, const Point2d& to) override { return makeRef<PathR2>(from, to); }
};

/// paths/path_se2.cpp: heading interpolated linearly and assigned to the member, i.e. not wrapped
class PathSE2 : public PathSE2Base {
public:
	PathSE2(const Pose2d& from, const Pose2d& to) : PathSE2Base(from, (to.position - from.position).norm()) { m_final = to; }
	Pose2d Interpolate(double ratio) const override
	{
		Pose2d interp;
		interp.position = { (1 - ratio) * m_init.x() + ratio * m_final.x(), (1 - ratio) * m_init.y() + ratio * m_final.y() };
		interp.theta = (1 - ratio) * m_init.theta + ratio * m_final.theta;
		return interp;
	}
	void Truncate(double ratio) override
	{
		m_final = Interpolate(ratio);
		m_length *= ratio;
	}
};
class PathConnectionSE2 : public PathConnectionSE2Base {
public:
	Ref<PathSE2Base> Connect(const Pose2d& from, const Pose2d& to) override { return makeRef<PathSE2>(from, to); }
};

/// models/kinematic_bicycle_model.{h,cpp}
class KinematicBicycleModel {
public:
	KinematicBicycleModel(double wheelbase = 2.6, double rearToCenter = 0.0) : m_wheelbase(wheelbase), m_rearToCenter(rearToCenter) { }
	/// d(heading) / d(distance of the reference point) for a steering angle, and cos(beta) (kinematic_bicycle_model.cpp:13-19)
	double Curvature(double steering, double* cosBetaOut = nullptr) const
	{
		const double tanSteering = std::tan(steering);
		const double beta = std::atan(m_rearToCenter * tanSteering / m_wheelbase);
		const double cosBeta = std::cos(beta);
		if (cosBetaOut)
			*cosBetaOut = cosBeta;
		return cosBeta * tanSteering / m_wheelbase;
	}
	Pose2d ConstantSteer(const Pose2d& from, double steering, double dist, Direction direction = Direction::Forward) const
	{
		if (direction == Direction::Backward)
			dist = -dist;
		Pose2d to = from;
		const double tanSteering = std::tan(steering);
		const double beta = std::atan(m_rearToCenter * tanSteering / m_wheelbase);
		const double cosBeta = std::cos(beta);
		const double DthetaDdist = cosBeta * tanSteering / m_wheelbase;
		dist = dist / cosBeta;
		if (std::abs(DthetaDdist) > 1e-9) {
			to.theta += dist * DthetaDdist; // not wrapped
			to.x() += 1 / DthetaDdist * (std::sin(beta + to.theta) - std::sin(beta + from.theta));
			to.y() += 1 / DthetaDdist * (-std::cos(beta + to.theta) + std::cos(beta + from.theta));
		} else {
			to.x() += dist * std::cos(from.theta);
			to.y() += dist * std::sin(from.theta);
		}
		return to;
	}
	double GetSteeringAngleFromTurningRadius(double radius) const
	{
		if (radius < m_rearToCenter)
			return M_PI_2;
		return std::atan(m_wheelbase / std::sqrt(std::pow(radius, 2) - std::pow(m_rearToCenter, 2)));
	}
	double Wheelbase() const { return m_wheelbase; }
	double RearToCenter() const { return m_rearToCenter; }

private:
	double m_wheelbase, m_rearToCenter;
};

/// paths/path_constant_steer.{h,cpp}
class PathConstantSteer : public PathNonHolonomicSE2Base {
public:
	PathConstantSteer(const Ref<KinematicBicycleModel>& model, const Pose2d& init, double steering, double length, Direction direction) :
		PathNonHolonomicSE2Base(init, length), m_model(model), m_steering(steering), m_direction(direction)
	{
		m_final = Interpolate(1.0);
	}
	Pose2d Interpolate(double ratio) const override { return m_model->ConstantSteer(m_init, m_steering, m_length * ratio, m_direction); }
	using PathNonHolonomicSE2Base::Interpolate;
	void Truncate(double ratio) override
	{
		m_final = Interpolate(ratio);
		m_length *= ratio;
	}
	Direction GetDirection(double /*ratio*/) const override { return m_direction; }
	double GetSteeringAngle() const { return m_steering; }
	const Ref<KinematicBicycleModel>& GetModel() const { return m_model; }

private:
	Ref<KinematicBicycleModel> m_model;
	double m_steering;
	Direction m_direction;
};

/// paths/path_reeds_shepp.{h,cpp} over the ABI's record; `word` / `cost` are those of the connection that made it.
class PathReedsShepp : public PathNonHolonomicSE2Base {
public:
	explicit PathReedsShepp(const pp_rs_path& rec) : PathNonHolonomicSE2Base(Pose2d(rec.start[0], rec.start[1], rec.start[2]), rec.length), m_rec(rec)
	{
		m_final = Raw(rec.final_pose);
	}
	Pose2d Interpolate(double ratio) const override
	{
		double pose[3];
		ppCheck(pp_rs_path_interpolate(HipContext::Get(), 1, &m_rec, &ratio, pose, nullptr));
		return Raw(pose);
	}
	/// all ratios in one launch
	std::vector<Pose2d> Interpolate(const std::vector<double>& ratios) const
	{
		std::vector<pp_rs_path> recs(ratios.size(), m_rec);
		std::vector<Pose2d> out(ratios.size());
		if (!ratios.empty())
			ppCheck(pp_rs_path_interpolate(HipContext::Get(), (int64_t)ratios.size(), recs.data(), ratios.data(), &out[0].position.v[0], nullptr));
		return out;
	}
	void Truncate(double ratio) override
	{
		ppCheck(pp_rs_path_truncate(HipContext::Get(), 1, &m_rec, &ratio, /*q11=*/1));
		m_final = Raw(m_rec.final_pose);
		m_length = m_rec.length;
	}
	Direction GetDirection(double ratio) const override
	{
		int32_t d = 2;
		ppCheck(pp_rs_path_interpolate(HipContext::Get(), 1, &m_rec, &ratio, nullptr, &d));
		return (Direction)d;
	}
	std::set<double> GetCuspPointRatios() const override
	{
		double r[4];
		int32_t n = 0;
		ppCheck(pp_rs_path_cusps(HipContext::Get(), 1, &m_rec, r, &n));
		return std::set<double>(r, r + n);
	}
	double GetMinTurningRadius() const { return m_rec.min_turning_radius; }
	const pp_rs_path& Record() const { return m_rec; }

private:
	static Pose2d Raw(const double* p)
	{
		Pose2d s;
		s.position = { p[0], p[1] };
		s.theta = p[2];
		return s;
	}
	pp_rs_path m_rec;
};

class PathConnectionReedsShepp : public PathConnectionSE2Base {
public:
	PathConnectionReedsShepp(double minTurningRadius = 1.0, double directionSwitchingCost = 0.0, double reverseCostMultiplier = 1.0, double forwardCostMultiplier = 1.0) :
		m_minTurningRadius(minTurningRadius), m_directionSwitchingCost(directionSwitchingCost), m_reverseCostMultiplier(reverseCostMultiplier),
		m_forwardCostMultiplier(forwardCostMultiplier) { }
	Ref<PathSE2Base> Connect(const Pose2d& from, const Pose2d& to) override
	{
		pp_rs_path rec;
		ppCheck(pp_rs_connect(HipContext::Get(), 1, &from.position.v[0], &to.position.v[0], m_minTurningRadius, (float)m_reverseCostMultiplier, (float)m_forwardCostMultiplier,
			(float)m_directionSwitchingCost, &rec));
		return makeRef<PathReedsShepp>(rec);
	}

private:
	double m_minTurningRadius, m_directionSwitchingCost, m_reverseCostMultiplier, m_forwardCostMultiplier;
};

} // namespace Planner
