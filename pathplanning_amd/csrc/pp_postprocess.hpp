// What the reference does after the graph search, one workgroup per query, batched over the planner's
// queries; included by pp_planner.hip (it reads the planner's PathRec / RsLogEntry / DevResult records where the search left them).
//   algo/hybrid_a_star.cpp:175-184   GraphSearch::GetCompositePath         (edges = the solution's actions)
//   paths/path_composite.{h,cpp}     PushBack / FindSegment / Interpolate, GetCuspPointRatios (sub-path ratios are inserted
//                                    unscaled, as the reference does)
//   algo/hybrid_a_star.cpp:260-291   sampling ratios every `pathInterpolation` with cusp snapping
//   algo/smoother.cpp:33-226         Smoother::Smooth: gradient descent on the sampled points (path, collision, Voronoi, smoothing
//                                    and curvature terms), at most maxIterations Jacobi steps, then IsPathSafe
//   algo/hybrid_a_star.cpp:293-303   the smoothed path, or the sampled one when smoothing fails
// The serial parts (prefix lengths, cusp set, ratio list: each a running double sum whose order is part of the result) are run
// by one lane; sampling and every smoother iteration are parallel over the points.  A point's gradient receives its
// contributions in the reference's order: the curvature term of point i-1, the five terms of point i, the curvature term of
// point i+1 (the reference walks the optimised indices in ascending order and `+=`s into gradients[i-1], [i], [i+1]).
#pragma once

constexpr int kPostThreads = 256;

struct PostParams {
	float pathInterpolation;
	float stepTolerance;
	int maxIterations;
	float learningRate, pathWeight, smoothWeight, voronoiWeight, collisionWeight, curvatureWeight, collisionRatio, maxCurvature;
	float alpha, dMax; // GVD::alpha / dMax, gvd.h:181
	int maxPoints;
};

struct PostBuffers {
	double* ratios;     // [B][maxPoints]
	double* resampled;  // [B][maxPoints][3]
	double* smoothed;   // [B][maxPoints][3]
	uint8_t* cusp;      // [B][maxPoints]
	uint8_t* optimise;  // [B][maxPoints]: the smoother's `indices`
	double* edgeEnd;    // [B][maxPath + 1]: PathInfo::finalLength of every edge (initLength = the previous one)
	pp_post_result* out; // [B]
};

struct V2d {
	double x, y;
};
__device__ __forceinline__ double v2_dot(V2d a, V2d b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ double v2_sq(V2d a) { return a.x * a.x + a.y * a.y; }
__device__ __forceinline__ double v2_norm(V2d a) { return sqrt(v2_sq(a)); }
__device__ __forceinline__ V2d v2_normalized(V2d a)
{
	const double n = v2_norm(a);
	return { a.x / n, a.y / n };
}
/// smoother.cpp:10-13
__device__ __forceinline__ V2d orthogonal_complement(V2d a, V2d b)
{
	const double d = v2_dot(a, b), sq = v2_sq(b);
	return { a.x - d * b.x / sq, a.y - d * b.y / sq };
}

/// Smoother::CalculateCurvatureTerm, smoother.cpp:160-214: the three gradient contributions of the corner at xi
__device__ __forceinline__ bool curvature_term(const PostParams& P, V2d xim1, V2d xi, V2d xip1, V2d& gim1, V2d& gi, V2d& gip1)
{
	const V2d deltaXi = { xi.x - xim1.x, xi.y - xim1.y }, deltaXip1 = { xip1.x - xi.x, xip1.y - xi.y };
	float c = (float)v2_dot(v2_normalized(deltaXi), v2_normalized(deltaXip1));
	c = c < -1.0f ? -1.0f : (1.0f < c ? 1.0f : c); // std::clamp<float>: a NaN passes through
	// smoother.cpp:164,200 call acos / cos / sqrt UNQUALIFIED on floats: with <cmath> alone (no <math.h>, no using-directive) these
	// are the C library's double functions -- the argument is promoted, the result narrowed on assignment.  So acos runs in
	// double and is stored as float, and 1 - cos^2 is formed in DOUBLE from the double cosine of that float (no float rounding
	// of the cosine in between: rounds 1-2 had one, which made the cancellation below 1e-4 .. 1e-3 relative).
	const float deltaPhi = (float)acos((double)c);
	const float kappa = (float)((double)deltaPhi / v2_norm(deltaXi));
	if (kappa <= P.maxCurvature)
		return false;
	const float denominator = (float)(v2_norm(deltaXi) * v2_norm(deltaXip1));
	const V2d oc1 = orthogonal_complement(deltaXip1, deltaXi), oc2 = orthogonal_complement(deltaXi, deltaXip1);
	const V2d dA = { oc1.x / (double)denominator, oc1.y / (double)denominator }, dB = { oc2.x / (double)denominator, oc2.y / (double)denominator };
	const double cs = cos((double)deltaPhi);
	const float DdeltaPhi = (float)((double)-1.0f / sqrt((double)1.0f - cs * cs));
	const float coef1 = (float)(1 / v2_norm(deltaXi) * (double)DdeltaPhi);
	const V2d nrm = v2_normalized(deltaXi);
	const double c2 = (double)deltaPhi / v2_sq(deltaXi);
	const V2d coef2 = { c2 * nrm.x, c2 * nrm.y };
	const V2d Dxim1 = { -dA.x, -dA.y }, Dxi = { dA.x - dB.x, dA.y - dB.y }, Dxip1 = dB;
	const double k1 = (double)coef1;
	const V2d Dk_im1 = { k1 * Dxim1.x + coef2.x, k1 * Dxim1.y + coef2.y };
	const V2d Dk_i = { k1 * Dxi.x - coef2.x, k1 * Dxi.y - coef2.y };
	const V2d Dk_ip1 = { k1 * Dxip1.x, k1 * Dxip1.y };
	const double w = (double)(P.curvatureWeight * (kappa - P.maxCurvature));
	gim1 = { w * Dk_im1.x, w * Dk_im1.y };
	gi = { w * Dk_i.x, w * Dk_i.y };
	gip1 = { w * Dk_ip1.x, w * Dk_ip1.y };
	return true;
}

/// the edge that ends at PathRec `rec` (records are stored goal first): its path object evaluated at `ratio`
struct PostEdge {
	int kind; // 1 arc, 2 Reeds-Shepp
	Arc arc;
	rs::Path rsp;
	__device__ __forceinline__ Pose interpolate(double ratio) const { return kind == 1 ? arc.interpolate(ratio) : rsp.interpolate(ratio); }
	__device__ __forceinline__ int direction(double ratio) const { return kind == 1 ? (arc.backward ? rs::kBwd : rs::kFwd) : rsp.direction(ratio); }
	__device__ __forceinline__ double length() const { return kind == 1 ? arc.length : rsp.length; }
};

__device__ inline PostEdge load_edge(const SearchArgs& A, const PathRec* recs, int nPath, int e /* 1 .. nPath-1, root-first numbering */, const RsLogEntry* rslog, int nRsLog)
{
	const PathRec child = recs[nPath - 1 - e], parent = recs[nPath - e];
	PostEdge E;
	const Pose from = { parent.x, parent.y, parent.t };
	if (child.action >= 1000) {
		E.kind = 2;
		E.rsp.init = from;
		E.rsp.rmin = A.rmin;
		double t = 0, u = 0, v = 0;
		for (int k = 0; k < nRsLog; k++)
			if (rslog[k].node == child.node) {
				t = rslog[k].t;
				u = rslog[k].u;
				v = rslog[k].v;
			}
		rs::word_segment(child.action - 1000, t, u, v, E.rsp.seg);
		E.rsp.length = E.rsp.seg.length * A.rmin; // PathSegment::GetLength
	} else {
		E.kind = 1;
		E.arc = Arc { from, A.prims.kappa[child.action], child.length, A.prims.backward[child.action] };
	}
	return E;
}

__global__ void __launch_bounds__(kPostThreads) k_postprocess(SearchArgs A, PostParams P, int nQueries, const PathRec* __restrict__ pathBase, const RsLogEntry* __restrict__ rsLogBase,
	const DevResult* __restrict__ results, const uint32_t* __restrict__ obstLabel, const uint32_t* __restrict__ voroLabel, PostBuffers B)
{
	const int q = blockIdx.x, tid = threadIdx.x;
	if (q >= nQueries)
		return;
	const DevResult res = results[q];
	const int cap = P.maxPoints;
	double* const ratios = B.ratios + (size_t)q * cap;
	double* const resampled = B.resampled + (size_t)q * cap * 3;
	double* const smoothed = B.smoothed + (size_t)q * cap * 3;
	uint8_t* const cusp = B.cusp + (size_t)q * cap;
	uint8_t* const optimise = B.optimise + (size_t)q * cap;
	double* const edgeEnd = B.edgeEnd + (size_t)q * (A.maxPath + 1);
	const PathRec* const recs = pathBase + (size_t)q * A.maxPath;
	const RsLogEntry* const rslog = rsLogBase + (size_t)q * kRsLogCap;
	const MapView& m = A.m;
	__shared__ int s_n, s_status, s_iter, s_go;
	__shared__ double s_length;
	__shared__ float s_stepW[kPostThreads / 64];
	__shared__ float s_step;
	__shared__ int s_unsafe;
	extern __shared__ double s_pos[]; // [2 * cap] current positions (x, y interleaved)

	const int nPath = res.r.n_path;
	if (res.r.status != 0 || nPath < 2 || nPath > A.maxPath) {
		if (tid == 0)
			B.out[q] = pp_post_result { 0, res.r.status != 0 ? -1 : (nPath > A.maxPath ? -4 : 2), 0, 0, 0.0 };
		return; // (a one-node path: start == goal; nothing to sample)
	}
	// ---------------- serial prologue (lane 0): composite lengths, cusp set, sampling ratios
	if (tid == 0) {
		const int nEdges = nPath - 1;
		double length = 0.0;
		edgeEnd[0] = 0.0;
		for (int e = 1; e <= nEdges; e++) { // PushBack, path_composite.h:33-39
			const PostEdge E = load_edge(A, recs, nPath, e, rslog, res.nRsLog);
			length += E.length();
			edgeEnd[e] = length;
		}
		// GetCuspPointRatios, path_composite.cpp:4-27, as a sorted array without duplicates, plus 0 and 1 (hybrid_a_star.cpp:272-273)
		constexpr int kMaxCusps = 64;
		double cs[kMaxCusps];
		int nc = 0;
		auto insert = [&](double r) {
			int pos = 0;
			while (pos < nc && cs[pos] < r)
				pos++;
			if (pos < nc && cs[pos] == r)
				return;
			if (nc == kMaxCusps)
				return;
			for (int k = nc; k > pos; k--)
				cs[k] = cs[k - 1];
			cs[pos] = r;
			nc++;
		};
		{
			double len = 0.0;
			int prevDirection = load_edge(A, recs, nPath, 1, rslog, res.nRsLog).direction(0.0);
			for (int e = 1; e <= nEdges; e++) {
				const PostEdge E = load_edge(A, recs, nPath, e, rslog, res.nRsLog);
				if (E.direction(0.0) != prevDirection)
					insert(len / length);
				if (E.kind == 2) {
					double sub[4];
					const int k = E.rsp.cusps(sub);
					for (int j = 0; j < k; j++)
						insert(sub[j]); // as a ratio of the sub-path: the reference does not rescale it
				}
				prevDirection = E.direction(1.0);
				len += E.length();
			}
		}
		insert(0.0);
		insert(1.0);
		int n = 0, ci = 0;
		bool overflow = false;
		const float pi = P.pathInterpolation;
		for (double l = 0.0; l <= length; l += (double)pi) {
			if (n >= cap) {
				overflow = true;
				break;
			}
			// `*cuspRatioIt` past the end of the set is undefined in the reference; here nothing snaps any more
			const double cuspLength = ci < nc ? cs[ci] * length : __builtin_huge_val();
			if (ci < nc && l >= cuspLength - (double)(pi / 2) && l < cuspLength + (double)(pi / 2)) {
				cusp[n] = 1;
				ratios[n] = cs[ci];
				ci++;
			} else {
				cusp[n] = 0;
				ratios[n] = l / length;
			}
			n++;
			if (!(pi > 0.0f))
				break;
		}
		s_n = n;
		s_length = length;
		s_status = overflow ? -4 : 0;
	}
	__syncthreads();
	const int n = s_n;
	if (s_status == -4) {
		if (tid == 0)
			B.out[q] = pp_post_result { 0, -4, 0, 0, s_length };
		return;
	}
	// ---------------- sampling: PathComposite::Interpolate for every ratio
	{
		const int nEdges = nPath - 1;
		for (int i = tid; i < n; i += kPostThreads) {
			const double len = ratios[i] * s_length;
			int lo = 1, hi = nEdges + 1; // upper_bound over finalLength
			while (lo < hi) {
				const int mid = lo + (hi - lo) / 2;
				if (len < edgeEnd[mid])
					hi = mid;
				else
					lo = mid + 1;
			}
			int e = lo;
			double pathRatio = 0.0;
			if (e == nEdges + 1) {
				e = nEdges;
				pathRatio = 1.0;
			}
			const PostEdge E = load_edge(A, recs, nPath, e, rslog, res.nRsLog);
			if (lo != nEdges + 1) // FindSegment, path_composite.h:66-82: the edge's own length, its start on the running sum
				pathRatio = E.length() == 0.0 ? 0.0 : (len - edgeEnd[e - 1]) / E.length();
			const Pose s = E.interpolate(pathRatio);
			resampled[3 * i] = s.x, resampled[3 * i + 1] = s.y, resampled[3 * i + 2] = s.t;
			smoothed[3 * i] = s.x, smoothed[3 * i + 1] = s.y, smoothed[3 * i + 2] = s.t;
			s_pos[2 * i] = s.x, s_pos[2 * i + 1] = s.y;
			optimise[i] = 0;
		}
	}
	__syncthreads();
	// ---------------- Smoother::Smooth
	if (n < 5) { // smoother.cpp:45-50
		int bad = 0;
		for (int i = tid; i < n; i += kPostThreads) {
			float d;
			if (!is_state_valid(m, resampled[3 * i], resampled[3 * i + 1], resampled[3 * i + 2], d))
				bad = 1;
		}
		if (tid == 0)
			s_unsafe = 0;
		__syncthreads();
		if (bad)
			atomicOr(&s_unsafe, 1);
		__syncthreads();
		if (tid == 0)
			B.out[q] = pp_post_result { n, s_unsafe ? -1 : 2, 0, 0, s_length };
		return;
	}
	if (tid == 0) { // the indices to optimise, smoother.cpp:54-76
		for (int i = 0; i < n - 4; i++) {
			if (cusp[i + 4]) {
				i += 3;
				continue;
			}
			if (cusp[i + 3]) {
				i += 2;
				continue;
			}
			if (cusp[i + 2]) {
				i += 1;
				continue;
			}
			if (cusp[i + 1])
				continue;
			if (cusp[i])
				continue;
			optimise[i + 2] = 1;
		}
		s_step = P.stepTolerance;
		s_iter = 0;
		s_go = 1;
	}
	__syncthreads();
	const float unsafeRadius = m.minSafeRadius * (1 + P.collisionRatio);
	constexpr int kPer = 8; // points per thread: maxPoints <= kPer * kPostThreads
	int count = -1, status = -1;
	for (;;) {
		count++;
		if (count >= P.maxIterations) {
			status = 0; // MaxIteration
			break;
		}
		if (s_step < P.stepTolerance) {
			status = 1; // StepTolerance
			break;
		}
		V2d g[kPer];
		float stepLocal = 0.0f;
#pragma unroll
		for (int u = 0; u < kPer; u++) {
			const int i = tid + u * kPostThreads;
			g[u] = { 0.0, 0.0 };
			if (i >= n)
				continue;
			auto pos = [&](int k) -> V2d { return { s_pos[2 * k], s_pos[2 * k + 1] }; };
			V2d a, b, c;
			// contribution of the corner at i-1 (its gip1), then the terms of i, then of the corner at i+1 (its gim1)
			if (i - 1 >= 2 && optimise[i - 1] && curvature_term(P, pos(i - 2), pos(i - 1), pos(i), a, b, c)) {
				g[u].x += c.x;
				g[u].y += c.y;
			}
			if (optimise[i]) {
				const V2d curr = pos(i);
				g[u].x += (double)P.pathWeight * (resampled[3 * i] - curr.x);
				g[u].y += (double)P.pathWeight * (resampled[3 * i + 1] - curr.y);
				int row, col;
				world_to_cell(m, curr.x, curr.y, row, col);
				if (inside_map(m, row, col)) {
					const size_t cell = (size_t)row * m.cols + col;
					const uint32_t lo = obstLabel[cell];
					// GridCellToWorldPosition, occupancy_map.h:94-97,150-153: grid origin + cell * resolution (float product)
					const V2d ow = { m.gx + (double)((int)(lo >> 16) * m.res), m.gy + (double)((int)(lo & 0xFFFFu) * m.res) };
					const V2d dirObs = { curr.x - ow.x, curr.y - ow.y };
					const float obstDist = (float)v2_norm(dirObs);
					if (obstDist < unsafeRadius) {
						const double s = (double)(P.collisionWeight * (obstDist - unsafeRadius));
						g[u].x += s * dirObs.x / (double)obstDist;
						g[u].y += s * dirObs.y / (double)obstDist;
					}
					if (obstDist < P.dMax && P.voronoiWeight > 0.0f) {
						const uint32_t lv = voroLabel[cell];
						const V2d vw = { m.gx + (double)((int)(lv >> 16) * m.res), m.gy + (double)((int)(lv & 0xFFFFu) * m.res) };
						const V2d dirVoro = { curr.x - vw.x, curr.y - vw.y };
						const float voroDist = (float)v2_norm(dirVoro);
						if (voroDist > 0.0f) {
							const float alphaPlusObstDist = P.alpha + obstDist;
							const float obstDistMinusDMax = obstDist - P.dMax;
							const float obstDistPlusVoroDist = obstDist + voroDist;
							const float dMaxSquared = P.dMax * P.dMax;
							const float pvdv = (P.alpha / alphaPlusObstDist) * (obstDistMinusDMax * obstDistMinusDMax / dMaxSquared) * (obstDist / (obstDistPlusVoroDist * obstDistPlusVoroDist));
							const float pvdo = (P.alpha / alphaPlusObstDist) * (voroDist / obstDistPlusVoroDist) * (obstDistMinusDMax / dMaxSquared)
								* (-obstDistMinusDMax / alphaPlusObstDist - obstDistMinusDMax / obstDistPlusVoroDist + 2);
							g[u].x += (double)P.voronoiWeight * ((double)pvdo * dirObs.x / (double)obstDist + (double)pvdv * dirVoro.x / (double)voroDist);
							g[u].y += (double)P.voronoiWeight * ((double)pvdo * dirObs.y / (double)obstDist + (double)pvdv * dirVoro.y / (double)voroDist);
						}
					}
				}
				const V2d p2 = pos(i - 2), p1 = pos(i - 1), n1 = pos(i + 1), n2 = pos(i + 2);
				g[u].x += (double)P.smoothWeight * (p2.x - 4 * p1.x + 6 * curr.x - 4 * n1.x + n2.x);
				g[u].y += (double)P.smoothWeight * (p2.y - 4 * p1.y + 6 * curr.y - 4 * n1.y + n2.y);
				if (curvature_term(P, p1, curr, n1, a, b, c)) {
					g[u].x += b.x;
					g[u].y += b.y;
				}
			}
			if (i + 1 < n - 2 && optimise[i + 1] && curvature_term(P, pos(i), pos(i + 1), pos(i + 2), a, b, c)) {
				g[u].x += a.x;
				g[u].y += a.y;
			}
			const float nrm = (float)v2_norm(g[u]);
			stepLocal = (stepLocal < nrm) ? nrm : stepLocal; // std::max<float>(step, norm): a NaN norm is not taken
		}
		// step = max over all points; wave reduce, then across the waves
		for (int off = 32; off > 0; off >>= 1) {
			const float o = __shfl_xor(stepLocal, off, 64);
			stepLocal = stepLocal < o ? o : stepLocal;
		}
		__syncthreads(); // every read of the current positions is done
		if ((tid & 63) == 0)
			s_stepW[tid >> 6] = stepLocal;
#pragma unroll
		for (int u = 0; u < kPer; u++) {
			const int i = tid + u * kPostThreads;
			if (i < n) {
				s_pos[2 * i] = s_pos[2 * i] - (double)P.learningRate * g[u].x;
				s_pos[2 * i + 1] = s_pos[2 * i + 1] - (double)P.learningRate * g[u].y;
			}
		}
		__syncthreads();
		if (tid == 0) {
			float st = 0.0f;
			for (int w = 0; w < kPostThreads / 64; w++)
				st = st < s_stepW[w] ? s_stepW[w] : st;
			s_step = st;
		}
		__syncthreads();
	}
	// ---------------- IsPathSafe + results
	if (tid == 0)
		s_unsafe = 0;
	__syncthreads();
	int bad = 0;
	for (int i = tid; i < n; i += kPostThreads) {
		smoothed[3 * i] = s_pos[2 * i];
		smoothed[3 * i + 1] = s_pos[2 * i + 1];
		float d;
		if (!is_state_valid(m, s_pos[2 * i], s_pos[2 * i + 1], smoothed[3 * i + 2], d))
			bad = 1;
	}
	if (bad)
		atomicOr(&s_unsafe, 1);
	__syncthreads();
	if (tid == 0)
		B.out[q] = pp_post_result { n, s_unsafe ? -1 : status, count, 0, s_length };
}
