// Device-side Reeds-Shepp solver and path (geometry/reeds_shepp.cpp,
// paths/path_reeds_shepp.cpp of the reference), gfx950, FP64, -ffp-contract=off.
#pragma once

#include "pp_device.hpp"

namespace ppd {
namespace rs {

	constexpr int kNumWords = 48;
	constexpr int kNumMotion = 5;
	// enum values of the reference: Steer {Left, Straight, Right}, Direction {Forward, Backward, NoMotion}
	constexpr int kLeft = 0, kStraight = 1, kRight = 2;
	constexpr int kFwd = 0, kBwd = 1, kNoMotion = 2;

	PPD_INLINE double inf() { return __builtin_huge_val(); }
	PPD_INLINE double modulo(double in, double mod)
	{
		// utils/maths.h:9-16
		double out = fmod(in, mod);
		if (out < 0)
			out += mod;
		return out;
	}
	PPD_INLINE bool angle_invalid(double th) { return th < 0 || th > kPi; }          // reeds_shepp.cpp:11-14
	PPD_INLINE double wrap_angle(double th) { return modulo(th + kPi, 2 * kPi) - kPi; } // reeds_shepp.cpp:16-19

	/// SE(2) `lhs - rhs`, geometry/2dplane.h:65-79
	PPD_INLINE Pose between(const Pose& lhs, const Pose& rhs)
	{
		const double dx = lhs.x - rhs.x, dy = lhs.y - rhs.y;
		const double a = -rhs.t;
		const double s = sin(a), c = cos(a);
		Pose out;
		out.x = c * dx + (-s) * dy;
		out.y = s * dx + c * dy;
		out.t = wrap_theta(wrap_theta(lhs.t - rhs.t));
		return out;
	}
	/// SE(2) `lhs + rhs`, geometry/2dplane.h:49-62 (rhs.t already wrapped by its constructor)
	PPD_INLINE Pose compose(const Pose& lhs, const Pose& rhs)
	{
		const double s = sin(lhs.t), c = cos(lhs.t);
		Pose out;
		out.x = c * rhs.x + (-s) * rhs.y;
		out.y = s * rhs.x + c * rhs.y;
		out.t = wrap_theta(lhs.t + rhs.t);
		out.x += lhs.x;
		out.y += lhs.y;
		out.t = wrap_theta(out.t);
		return out;
	}

	/// The twelve base-word formulas, reeds_shepp.cpp:21-304.  family = word / 4.
	PPD_INLINE double base_lengths(int family, double gx, double gy, double gt, double& t, double& u, double& v)
	{
		const double sg = sin(gt), cg = cos(gt);
		// families 0,2,3,4,7,9 use (x - sin, y - 1 + cos); the others (x + sin, y - 1 - cos)
		const bool minusForm = (family == 0 || family == 2 || family == 3 || family == 4 || family == 7 || family == 9);
		const double xi = minusForm ? gx - sg : gx + sg;
		const double eta = minusForm ? gy - 1 + cg : gy - 1 - cg;
		switch (family) {
		case 0: { // LfSfLf
			u = sqrt(xi * xi + eta * eta);
			t = atan2(eta, xi);
			v = wrap_angle(gt - t);
			if (angle_invalid(t) || angle_invalid(v))
				return inf();
			return t + u + v;
		}
		case 1: { // LfSfRf -- the `u1squared < 4` test has no return in the reference (NaN flows on)
			double u1squared = xi * xi + eta * eta;
			double t1 = atan2(eta, xi);
			u = sqrt(u1squared - 4);
			double phi = atan2(2.0, u);
			t = wrap_angle(t1 + phi);
			v = wrap_angle(t - gt);
			if (angle_invalid(t) || angle_invalid(v))
				return inf();
			return t + u + v;
		}
		case 2: { // LfRbLf
			double u1 = sqrt(xi * xi + eta * eta);
			if (u1 > 4)
				return inf();
			double phi = atan2(eta, xi);
			double alpha = acos(u1 / 4.0);
			t = modulo(kPi2 + alpha + phi, 2 * kPi);
			u = modulo(kPi - 2 * alpha, 2 * kPi);
			v = modulo(gt - t - u, 2 * kPi);
			if (angle_invalid(t) || angle_invalid(u) || angle_invalid(v))
				return inf();
			return t + u + v;
		}
		case 3: { // LfRbLb
			double u1 = sqrt(xi * xi + eta * eta);
			if (u1 > 4)
				return inf();
			double phi = atan2(eta, xi);
			double alpha = acos(u1 / 4.0);
			t = modulo(kPi2 + alpha + phi, 2 * kPi);
			u = modulo(kPi - 2 * alpha, 2 * kPi);
			v = modulo(t + u - gt, 2 * kPi);
			return t + u + v;
		}
		case 4: { // LfRfLb
			double u1 = sqrt(xi * xi + eta * eta);
			if (u1 > 4)
				return inf();
			double phi = atan2(eta, xi);
			u = acos((8 - u1 * u1) / 8.0);
			double va = sin(u);
			double alpha = asin(2 * va / u1);
			t = modulo(kPi2 - alpha + phi, 2 * kPi);
			v = modulo(t - u - gt, 2 * kPi);
			return t + u + v;
		}
		case 5: { // LfRufLubRb
			double u1 = sqrt(xi * xi + eta * eta);
			if (u1 > 4)
				return inf();
			double phi = atan2(eta, xi);
			if (u1 > 2) {
				double alpha = acos(u1 / 4 - 0.5);
				t = modulo(kPi2 + phi - alpha, 2 * kPi);
				u = modulo(kPi - alpha, 2 * kPi);
				v = modulo(gt - t + 2 * u, 2 * kPi);
			} else {
				double alpha = acos(u1 / 4 + 0.5);
				t = modulo(kPi2 + phi + alpha, 2 * kPi);
				u = modulo(alpha, 2 * kPi);
				v = modulo(gt - t + 2 * u, 2 * kPi);
			}
			return t + u + u + v;
		}
		case 6: { // LfRubLubRf
			double u1 = sqrt(xi * xi + eta * eta);
			if (u1 > 6)
				return inf();
			double phi = atan2(eta, xi);
			double va1 = 1.25 - u1 * u1 / 16;
			if (va1 < 0 || va1 > 1)
				return inf();
			u = acos(va1);
			double va2 = sin(u);
			double alpha = asin(2 * va2 / u1);
			t = modulo(kPi2 + phi + alpha, 2 * kPi);
			v = modulo(t - gt, 2 * kPi);
			return t + u + u + v;
		}
		case 7: { // LfRbpi2SbLb
			double u1squared = xi * xi + eta * eta;
			if (u1squared < 4)
				return inf();
			double phi = atan2(eta, xi);
			u = sqrt(u1squared - 4) - 2;
			if (u < 0)
				return inf();
			double alpha = atan2(2.0, u + 2);
			t = modulo(kPi2 + phi + alpha, 2 * kPi);
			v = modulo(t + kPi2 - gt, 2 * kPi);
			return t + kPi2 + u + v;
		}
		case 8: { // LfRbpi2SbRb
			double u1 = sqrt(xi * xi + eta * eta);
			if (u1 < 2)
				return inf();
			double phi = atan2(eta, xi);
			t = modulo(kPi2 + phi, 2 * kPi);
			u = u1 - 2;
			v = modulo(gt - t - kPi2, 2 * kPi);
			return t + kPi2 + u + v;
		}
		case 9: { // LfSfRfpi2Lb
			double u1squared = xi * xi + eta * eta;
			if (u1squared < 4)
				return inf();
			double phi = atan2(eta, xi);
			u = sqrt(u1squared - 4) - 2;
			if (u < 0)
				return inf();
			double alpha = atan2(u + 2, 2.0);
			t = modulo(kPi2 + phi - alpha, 2 * kPi);
			v = modulo(t - kPi2 - gt, 2 * kPi);
			return t + u + kPi2 + v;
		}
		case 10: { // LfSfLfpi2Rb
			double u1 = sqrt(xi * xi + eta * eta);
			if (u1 < 2)
				return inf();
			double phi = atan2(eta, xi);
			t = modulo(phi, 2 * kPi);
			u = u1 - 2;
			v = modulo(-t - kPi2 + gt, 2 * kPi);
			return t + u + kPi2 + v;
		}
		default: { // 11: LfRbpi2SbLbpi2Rf
			double u1squared = xi * xi + eta * eta;
			if (u1squared < 16)
				return inf();
			double phi = atan2(eta, xi);
			u = sqrt(u1squared - 4) - 4;
			if (u < 0)
				return inf();
			double alpha = atan2(2.0, u + 4);
			t = modulo(kPi2 + phi + alpha, 2 * kPi);
			v = modulo(t - gt, 2 * kPi);
			return t + u + v + kPi;
		}
		}
	}

	/// A path word expanded into motions: reeds_shepp.cpp:306-414 + 566-606.
	struct Segment {
		double len[kNumMotion];
		int8_t steer[kNumMotion];
		int8_t dir[kNumMotion];
		int n;         // number of motions added
		double length; // sum of |len| (normalised)
	};

	// per family: number of motions, steer and direction of each, and which parameter feeds it
	// parameter code: 0 = t, 1 = u, 2 = v, 3 = pi/2
	PPD_INLINE void word_segment(int word, double t, double u, double v, Segment& s)
	{
		// tables in registers/constant memory via switch to stay branch-cheap
		int n;
		int st[5], dr[5], pm[5];
#define PPD_M(i, S, D, P) \
	st[i] = S;            \
	dr[i] = D;            \
	pm[i] = P
		switch (word / 4) {
		case 0: n = 3; PPD_M(0, kLeft, kFwd, 0); PPD_M(1, kStraight, kFwd, 1); PPD_M(2, kLeft, kFwd, 2); break;
		case 1: n = 3; PPD_M(0, kLeft, kFwd, 0); PPD_M(1, kStraight, kFwd, 1); PPD_M(2, kRight, kFwd, 2); break;
		case 2: n = 3; PPD_M(0, kLeft, kFwd, 0); PPD_M(1, kRight, kBwd, 1); PPD_M(2, kLeft, kFwd, 2); break;
		case 3: n = 3; PPD_M(0, kLeft, kFwd, 0); PPD_M(1, kRight, kBwd, 1); PPD_M(2, kLeft, kBwd, 2); break;
		case 4: n = 3; PPD_M(0, kLeft, kFwd, 0); PPD_M(1, kRight, kFwd, 1); PPD_M(2, kLeft, kBwd, 2); break;
		case 5: n = 4; PPD_M(0, kLeft, kFwd, 0); PPD_M(1, kRight, kFwd, 1); PPD_M(2, kLeft, kBwd, 1); PPD_M(3, kRight, kBwd, 2); break;
		case 6: n = 4; PPD_M(0, kLeft, kFwd, 0); PPD_M(1, kRight, kBwd, 1); PPD_M(2, kLeft, kBwd, 1); PPD_M(3, kRight, kFwd, 2); break;
		case 7: n = 4; PPD_M(0, kLeft, kFwd, 0); PPD_M(1, kRight, kBwd, 3); PPD_M(2, kStraight, kBwd, 1); PPD_M(3, kLeft, kBwd, 2); break;
		case 8: n = 4; PPD_M(0, kLeft, kFwd, 0); PPD_M(1, kRight, kBwd, 3); PPD_M(2, kStraight, kBwd, 1); PPD_M(3, kRight, kBwd, 2); break;
		case 9: n = 4; PPD_M(0, kLeft, kFwd, 0); PPD_M(1, kStraight, kFwd, 1); PPD_M(2, kRight, kFwd, 3); PPD_M(3, kLeft, kBwd, 2); break;
		case 10: n = 4; PPD_M(0, kLeft, kFwd, 0); PPD_M(1, kStraight, kFwd, 1); PPD_M(2, kLeft, kFwd, 3); PPD_M(3, kRight, kBwd, 2); break;
		default: n = 5; PPD_M(0, kLeft, kFwd, 0); PPD_M(1, kRight, kBwd, 3); PPD_M(2, kStraight, kBwd, 1); PPD_M(3, kLeft, kBwd, 3); PPD_M(4, kRight, kFwd, 2); break;
		}
#undef PPD_M
		const int tr = word % 4; // 1 timeflip, 2 reflect, 3 both (reeds_shepp.cpp:592-604)
		s.n = n;
		s.length = 0.0;
#pragma unroll
		for (int i = 0; i < kNumMotion; i++) {
			if (i < n) {
				double l = pm[i] == 0 ? t : (pm[i] == 1 ? u : (pm[i] == 2 ? v : kPi2));
				int sd = st[i], dd = dr[i];
				if (tr & 1)
					dd = dd == kBwd ? kFwd : kBwd;
				if (tr & 2)
					sd = sd == kLeft ? kRight : (sd == kRight ? kLeft : sd);
				s.len[i] = l;
				s.steer[i] = (int8_t)sd;
				s.dir[i] = (int8_t)dd;
				s.length += fabs(l); // PathSegment::AddMotion, reeds_shepp.cpp:439-456
			} else {
				s.len[i] = inf();
				s.steer[i] = (int8_t)kLeft;
				s.dir[i] = (int8_t)kNoMotion;
			}
		}
	}

	PPD_INLINE bool motion_valid(const Segment& s, int i) { return s.len[i] != inf() && s.dir[i] != kNoMotion; } // reeds_shepp.cpp:425-428

	/// PathSegment::ComputeCost, reeds_shepp.cpp:469-497 (float result; the fast path ignores forwardCost).
	/// AddMotion stops adding once a motion slot looks invalid; a motion whose length is +inf would
	/// make later motions overwrite it -- lengths here are finite or NaN, so slots are filled in order.
	PPD_INLINE float compute_cost(const Segment& s, double rmin, float reverseCost, float forwardCost, float switchCost)
	{
		if (!motion_valid(s, 0))
			return __builtin_huge_valf();
		if (reverseCost == 1.0f && switchCost == 0.0f)
			return (float)(s.length * rmin);
		float cost = 0;
		int prevDir = s.dir[0];
#pragma unroll
		for (int i = 0; i < kNumMotion; i++) {
			if (!motion_valid(s, i))
				break;
			float motionCost = (float)(s.len[i] * rmin);
			if (s.dir[i] == kFwd)
				motionCost *= forwardCost;
			else if (s.dir[i] == kBwd)
				motionCost *= reverseCost;
			if (s.dir[i] != prevDir)
				motionCost += switchCost;
			prevDir = s.dir[i];
			cost += motionCost;
		}
		return cost;
	}

	/// GetGoalArray element `which` (reeds_shepp.cpp:521-535): the goal relative to the start,
	/// normalised by the turning radius, mirrored; theta goes through the Pose2d constructor.
	PPD_INLINE void goal_variant(const Pose& rel, int which, double& gx, double& gy, double& gt)
	{
		gx = (which & 1) ? -rel.x : rel.x;
		gy = (which & 2) ? -rel.y : rel.y;
		gt = (which == 1 || which == 2) ? -rel.t : rel.t;
		gt = wrap_theta(gt);
	}

	/// PathReedsShepp with Interpolate / Straight / Turn, paths/path_reeds_shepp.cpp:12-47,123-153
	struct Path {
		Pose init;
		Segment seg;
		double rmin;
		double length; // seg.length * rmin

		PPD_INLINE Pose straight(const Pose& start, int dir, double len) const
		{
			if (dir == kBwd)
				len = -len;
			len *= rmin;
			Pose e;
			e.x = start.x + len * cos(start.t);
			e.y = start.y + len * sin(start.t);
			e.t = wrap_theta(start.t);
			return e;
		}
		PPD_INLINE Pose turn(const Pose& start, int dir, int steer, double turnAngle) const
		{
			if (dir == kBwd)
				turnAngle = -turnAngle;
			double phi = turnAngle / 2;
			double cosPhi = cos(phi);
			double sinPhi = sin(phi);
			double L = 2 * sinPhi * rmin;
			double x = L * cosPhi;
			double y = L * sinPhi;
			if (steer == kRight) {
				y *= -1;
				turnAngle *= -1;
			}
			Pose rel;
			rel.x = x;
			rel.y = y;
			rel.t = wrap_theta(turnAngle);
			return compose(start, rel);
		}
		PPD_INLINE Pose interpolate(double ratio) const
		{
			const double totalLength = length;
			if (totalLength == 0)
				return init;
			Pose interp = init;
			double len = 0;
			for (int i = 0; i < kNumMotion; i++) {
				if (!motion_valid(seg, i))
					break;
				double motionLength = seg.len[i] * rmin;
				if (motionLength == 0)
					continue;
				double motionRatio = (ratio * totalLength - len) / motionLength;
				motionRatio = motionRatio < 1.0 ? motionRatio : 1.0; // std::min(motionRatio, 1.0)
				if (seg.steer[i] == kStraight)
					interp = straight(interp, seg.dir[i], seg.len[i] * motionRatio);
				else
					interp = turn(interp, seg.dir[i], seg.steer[i], seg.len[i] * motionRatio);
				len += motionLength;
				if (len >= ratio * totalLength)
					break;
			}
			return interp;
		}
		/// Interpolate's chain, computed once.  PathReedsShepp::Interpolate (paths/path_reeds_shepp.cpp:12-47) walks the motions from m_init
		/// for EVERY sample: the motions before the one that contains the sample are applied in full (motionRatio clamps to 1.0, and
		/// len * 1.0 == len), i.e. with the same operands every time.  make_prefix stores, per motion, the pose it starts from and the
		/// length before it (4 doubles each, then the end pose: kPrefixDoubles in all); interpolate_prefix continues from there with the one
		/// partial motion -- the same operations on the same values as interpolate(), hence the same bits, at a fifth of the trigonometry
		/// for a five-motion word.  The validity march of the search kernels' Reeds-Shepp child (one lane, ~30 samples) uses it.
		static constexpr int kPrefixDoubles = 4 * kNumMotion + 3;
		PPD_INLINE void make_prefix(double* pre) const
		{
			Pose interp = init;
			double len = 0;
			for (int i = 0; i < kNumMotion; i++) {
				pre[4 * i] = interp.x;
				pre[4 * i + 1] = interp.y;
				pre[4 * i + 2] = interp.t;
				pre[4 * i + 3] = len;
				if (!motion_valid(seg, i))
					continue; // (entries behind the last motion are never read)
				const double motionLength = seg.len[i] * rmin;
				if (motionLength == 0)
					continue;
				const double motionRatio = 1.0;
				if (seg.steer[i] == kStraight)
					interp = straight(interp, seg.dir[i], seg.len[i] * motionRatio);
				else
					interp = turn(interp, seg.dir[i], seg.steer[i], seg.len[i] * motionRatio);
				len += motionLength;
			}
			pre[4 * kNumMotion] = interp.x;
			pre[4 * kNumMotion + 1] = interp.y;
			pre[4 * kNumMotion + 2] = interp.t;
		}
		PPD_INLINE Pose interpolate_prefix(const double* pre, double ratio) const
		{
			const double totalLength = length;
			if (totalLength == 0)
				return init;
			for (int i = 0; i < kNumMotion; i++) {
				if (!motion_valid(seg, i))
					break;
				const double motionLength = seg.len[i] * rmin;
				if (motionLength == 0)
					continue;
				const double len = pre[4 * i + 3];
				if (len + motionLength >= ratio * totalLength) { // the motion at which interpolate() leaves its loop
					double motionRatio = (ratio * totalLength - len) / motionLength;
					motionRatio = motionRatio < 1.0 ? motionRatio : 1.0;
					const Pose from = { pre[4 * i], pre[4 * i + 1], pre[4 * i + 2] };
					return seg.steer[i] == kStraight ? straight(from, seg.dir[i], seg.len[i] * motionRatio) : turn(from, seg.dir[i], seg.steer[i], seg.len[i] * motionRatio);
				}
			}
			return { pre[4 * kNumMotion], pre[4 * kNumMotion + 1], pre[4 * kNumMotion + 2] }; // every motion in full
		}
		/// PathReedsShepp::Truncate, paths/path_reeds_shepp.cpp:49-93: returns m_final.  The reference means to drop the
		/// motions behind the cut but resets `m_motions[i]` -- the motion that CONTAINS the cut -- instead of `[ii]` whenever
		/// a later slot exists (i < 4); `q11` = true reproduces that, false drops the later ones.
		PPD_INLINE Pose truncate(double ratio, bool q11)
		{
			const double totalLength = length;
			Pose fin = init;
			if (totalLength != 0) {
				Pose interp = init;
				double len = 0;
				for (int i = 0; i < kNumMotion; i++) {
					if (!motion_valid(seg, i))
						break;
					double motionLength = seg.len[i] * rmin;
					if (motionLength == 0)
						continue;
					double motionRatio = (ratio * totalLength - len) / motionLength;
					motionRatio = motionRatio < 1.0 ? motionRatio : 1.0;
					if (seg.steer[i] == kStraight)
						interp = straight(interp, seg.dir[i], seg.len[i] * motionRatio);
					else
						interp = turn(interp, seg.dir[i], seg.steer[i], seg.len[i] * motionRatio);
					len += motionLength;
					if (len >= ratio * totalLength) {
						seg.len[i] *= motionRatio;
						for (int ii = i + 1; ii < kNumMotion; ii++) {
							const int k = q11 ? i : ii;
							seg.len[k] = inf(); // ReedsShepp::Motion(): {steer (indeterminate, kept), NoMotion, +inf}
							seg.dir[k] = (int8_t)kNoMotion;
						}
						break;
					}
				}
				fin = interp;
			}
			length *= ratio;
			return fin;
		}
		/// PathReedsShepp::GetDirection, paths/path_reeds_shepp.cpp:155-167
		PPD_INLINE int direction(double ratio) const
		{
			if (length == 0)
				return kNoMotion;
			double len = 0;
			for (int i = 0; i < kNumMotion; i++) {
				if (!motion_valid(seg, i))
					break;
				len += seg.len[i] * rmin;
				if (ratio * length <= len)
					return seg.dir[i];
			}
			return kNoMotion;
		}
		/// PathReedsShepp::GetCuspPointRatios, paths/path_reeds_shepp.cpp:95-121: the std::set as a sorted array without
		/// duplicates (at most 4 entries); returns the count
		PPD_INLINE int cusps(double* out) const
		{
			int count = 0;
			if (length == 0.0)
				return 0;
			int nm = 0; // PathSegment::GetNumMotions, reeds_shepp.cpp:458-467
			while (nm < kNumMotion && motion_valid(seg, nm))
				nm++;
			double len = seg.len[0] * rmin;
			for (int i = 1; i < nm; i++) {
				const double ratio = len / length;
				if (ratio > 1.0)
					break;
				if (seg.dir[i] != seg.dir[i - 1]) {
					int pos = 0;
					while (pos < count && out[pos] < ratio)
						pos++;
					if (!(pos < count && out[pos] == ratio)) {
						for (int k = count; k > pos; k--)
							out[k] = out[k - 1];
						out[pos] = ratio;
						count++;
					}
				}
				len += seg.len[i] * rmin;
			}
			return count;
		}
	};

	/// ReedsShepp::Solver::GetOptimalPath, reeds_shepp.cpp:654-683: one thread, all 48 words.
	/// Returns the word (-1 if none); tuv/cost/segLength of the winner.
	PPD_INLINE int optimal_word(const Pose& start, const Pose& goal, double rmin, float reverseCost, float forwardCost, float switchCost, double& bt, double& bu,
		double& bv, float& bestCost, double& segLength)
	{
		Pose rel = between(goal, start);
		rel.x = rel.x / rmin;
		rel.y = rel.y / rmin;
		bestCost = __builtin_huge_valf();
		int bestWord = -1;
		bt = bu = bv = 0.0;
		segLength = 0.0;
		for (int w = 0; w < kNumWords; w++) {
			double gx, gy, gt, t, u, v;
			goal_variant(rel, w % 4, gx, gy, gt);
			double length = base_lengths(w / 4, gx, gy, gt, t, u, v);
			if (length == inf())
				continue;
			Segment s;
			word_segment(w, t, u, v, s);
			float cost = compute_cost(s, rmin, reverseCost, forwardCost, switchCost);
			if (cost < bestCost) {
				bestCost = cost;
				bestWord = w;
				bt = t;
				bu = u;
				bv = v;
				segLength = s.length;
			}
		}
		return bestWord;
	}

/// a Path seen through its prefix (Path::make_prefix): what is_path_valid / voronoi_cost need -- `length` and `interpolate`
struct PrefixedPath {
	const Path& path;
	const double* pre;
	double length;
	PPD_INLINE Pose interpolate(double ratio) const { return path.interpolate_prefix(pre, ratio); }
};

} // namespace rs
} // namespace ppd
