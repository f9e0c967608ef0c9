// TEST INFRASTRUCTURE ONLY -- thin C entry points around the few reference
// headers that compile in this image WITHOUT Eigen/flann/spdlog:
//   utils/frontier.h (open list), utils/grid.{h,cpp} (neighbour order),
//   utils/random.h (global RNG), utils/maths.h.
// Built by oracle/Makefile from the sources where they lie under
// /root/reference (never copied), into oracle/_ref/libppref.so, and used by
// tests/test_oracle_ref.py to pin the oracle's open-list pop order (Appendix
// A Q1), neighbour enumeration (Q4), uniform sampling (Q9) and the SE(2) state-space
// samplers (uniform + Gaussian, state_space_se2.cpp:27-52) to the
// reference's own code.  Compiled with -fno-access-control so the driver can
// reseed Random<double>::s_engine (private static).
#include <algorithm>
#include <cassert>
#include <limits>
#include <string>
#include <stdexcept>
#include <vector>

#include "core/base.h"
#include "utils/frontier.h"
#include "utils/grid.h"
#include "utils/maths.h"
#include "utils/random.h"

namespace {
struct E {
	double cost;
	int id;
};
struct Cmp { // same shape as AStar::CompareNode (algo/a_star.h:226-231)
	bool operator()(const E& a, const E& b) const { return a.cost > b.cost; }
};
struct Hash {
	std::size_t operator()(const E& e) const { return std::hash<int>()(e.id); }
};
struct Eq {
	bool operator()(const E& a, const E& b) const { return a.id == b.id; }
};
}

extern "C" {

int ref_frontier_replay(int nops, const int* ops, const double* costs, int* popped)
{
	Planner::Frontier<E, Cmp, Hash, Eq> f;
	int np = 0;
	for (int i = 0; i < nops; i++) {
		if (ops[i] >= 0)
			f.Push({ costs[ops[i]], ops[i] });
		else if (!f.Empty())
			popped[np++] = f.Pop().id;
	}
	return np;
}

/// Frontier::Remove + Push of the same id with a new cost (the shortcut path, a_star.h:417-427)
int ref_frontier_replay_with_replace(int nops, const int* ops, const double* costs, const double* newCosts, int* popped)
{
	// ops[i] >= 0: push id ops[i] with costs[]; ops[i] == -1: pop; ops[i] <= -2: replace id (-2 - ops[i]) with newCosts[]
	Planner::Frontier<E, Cmp, Hash, Eq> f;
	int np = 0;
	for (int i = 0; i < nops; i++) {
		if (ops[i] >= 0)
			f.Push({ costs[ops[i]], ops[i] });
		else if (ops[i] == -1) {
			if (!f.Empty())
				popped[np++] = f.Pop().id;
		} else {
			int id = -2 - ops[i];
			if (f.Remove({ 0.0, id }))
				f.Push({ newCosts[id], id });
		}
	}
	return np;
}

void ref_neighbors(int row, int col, int rows, int cols, int* n, int* rc)
{
	auto v = Planner::GridCellPosition(row, col).GetNeighbors(rows, cols);
	*n = (int)v.size();
	for (size_t i = 0; i < v.size(); i++) {
		rc[2 * i] = v[i].row;
		rc[2 * i + 1] = v[i].col;
	}
}

void ref_rng_uniform(unsigned long long seed, long long n, double lb, double ub, double* out)
{
	Planner::Random<double>::Init();
	Planner::Random<double>::s_engine->seed(seed);
	Planner::Random<double>::s_uniformDistribution.reset();
	for (long long i = 0; i < n; i++)
		out[i] = Planner::Random<double>::SampleUniform(lb, ub);
}

/// n draws of StateSpaceSE2::SampleUniform's pattern (state_space_se2.cpp:27-38: x, y, theta from Random<double>::SampleUniform) followed
/// by n draws of SampleGaussian's (:40-52, before EnforceBounds), all from the reference's own utils/random.h
void ref_rng_se2(unsigned long long seed, long long n, const double* lb, const double* ub, const double* mean, const double* stdDev, double* uniformOut, double* gaussOut)
{
	Planner::Random<double>::Init();
	Planner::Random<double>::s_engine->seed(seed);
	Planner::Random<double>::s_uniformDistribution.reset();
	Planner::Random<double>::s_gaussianDistribution.reset();
	for (long long i = 0; i < n; i++)
		for (int k = 0; k < 3; k++)
			uniformOut[3 * i + k] = Planner::Random<double>::SampleUniform(lb[k], ub[k]);
	for (long long i = 0; i < n; i++)
		for (int k = 0; k < 3; k++)
			gaussOut[3 * i + k] = Planner::Random<double>::SampleGaussian(mean[k], stdDev[k]);
}

double ref_modulo(double a, double b) { return Planner::Maths::Modulo(a, b); }

} // extern "C"
