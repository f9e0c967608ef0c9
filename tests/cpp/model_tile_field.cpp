// CPU model of the tile-fixpoint obstacle-heuristic field (the algorithm k_wavefront_tiles runs on the GPU,
// pathplanning_amd/csrc/pp_wavefront_tiles.hip).  TEST INFRASTRUCTURE: tests/test_tile_field_model.py compares it with the
// oracle's sequential restatement of ObstaclesHeuristic::Update (algo/heuristics.cpp:106-153) and records the work figures
// (tile visits, rounds, candidate passes) the kernel's design is sized with.
//
// The reference pops a sorted open list (LIFO among equal costs) and never relaxes: a cell keeps the cost it got from the
// FIRST neighbour that was popped, i.e. from its allowed neighbour of smallest cost,
//     cost[n] = fl(cost[p] + edge(p, n)),  p = argmin over the allowed, reached neighbours of n,
// and the pop order among equal costs only matters when a straight and a diagonal neighbour tie for that minimum
// (the two edges differ).  Every other tie gives the same value whoever wins.  So the field is the unique fixed point of
// that equation (parents are strictly cheaper, every chain ends at the goal) unless such a tie occurs, which is DETECTED.
// A fixed point can be computed in any order: here tile by tile (64 x 64 cells, one wave on the GPU), each tile solved from
// scratch from its one-cell halo by bucket rounds (round k settles the neighbours of the cells whose cost lies in [k, k+1);
// every edge costs >= 1, so those neighbours land in buckets k+1 / k+2), and re-solved whenever a neighbouring tile's border
// changed afterwards.  When no tile is pending every cell satisfies the equation against its final neighbours: the field is
// the reference's, bit for bit.  Ties and runs that do not settle are reported; the caller falls back to the ordered kernel.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

constexpr int T = 64;
constexpr float kInf = std::numeric_limits<float>::infinity();

struct Stats {
	int64_t visits, rounds, passes, cells, emptyRounds, maskOps;
	int32_t tie, unsettled, tiles, maxVisits;
};

struct Model {
	const uint8_t* occ;
	int rows, cols, TR, TC;
	float* cost; // row-major, rows x cols
	std::vector<float> prio;
	std::vector<uint8_t> pending, visited;
	std::vector<int> nvis;
	Stats st {};
	float kDiag = std::sqrt(2.0f);

	bool occupied(int r, int c) const { return r < 0 || c < 0 || r >= rows || c >= cols || occ[(size_t)r * cols + c] != 0; }
	/// value a tile reads from its halo: the neighbour tile's cell if that tile has been solved at least once
	float halo(int r, int c) const
	{
		if (r < 0 || c < 0 || r >= rows || c >= cols)
			return kInf;
		if (!visited[(r / T) * TC + (c / T)])
			return kInf;
		return cost[(size_t)r * cols + c];
	}

	void solve(int tr, int tc, int goalR, int goalC)
	{
		const int r0 = tr * T, c0 = tc * T;
		static thread_local float L[T + 2][T + 2]; // local costs incl. halo; +inf = undiscovered or occupied
		static thread_local uint8_t O[T + 2][T + 2];
		for (int i = 0; i < T + 2; i++)
			for (int j = 0; j < T + 2; j++) {
				const int r = r0 + i - 1, c = c0 + j - 1;
				O[i][j] = occupied(r, c);
				const bool interior = i >= 1 && i <= T && j >= 1 && j <= T;
				L[i][j] = interior ? kInf : halo(r, c);
			}
		// bit c of row r (interior coordinates 0..63)
		uint64_t cur[T] = {}, nx1[T] = {}, nx2[T] = {}, closed[T] = {};
		for (int i = 0; i < T; i++)
			for (int j = 0; j < T; j++)
				if (O[i + 1][j + 1])
					closed[i] |= 1ull << j;
		// static masks of allowed diagonal moves INTO cell (i, j) from (i-1, j-1) etc. (heuristics.cpp:130-132)
		int kmin = INT32_MAX;
		bool goalHere = goalR >= r0 && goalR < r0 + T && goalC >= c0 && goalC < c0 + T;
		if (goalHere) {
			L[goalR - r0 + 1][goalC - c0 + 1] = 0.0f;
			closed[goalR - r0] |= 1ull << (goalC - c0);
			cur[goalR - r0] |= 1ull << (goalC - c0);
			kmin = 0;
		}
		float hmax = -1.0f;
		for (int i = 0; i < T + 2; i++)
			for (int j = 0; j < T + 2; j++) {
				const bool interior = i >= 1 && i <= T && j >= 1 && j <= T;
				if (!interior && L[i][j] < kInf) {
					const int b = (int)L[i][j];
					if (b < kmin)
						kmin = b;
					if (L[i][j] > hmax)
						hmax = L[i][j];
				}
			}
		st.visits++;
		if (kmin == INT32_MAX)
			goto writeback;
		for (int k = kmin;; k++) {
			// members of bucket k: interior `cur` bits + halo cells whose cost lies in [k, k+1)
			auto member = [&](int i, int j) -> bool { // padded coordinates
				const bool interior = i >= 1 && i <= T && j >= 1 && j <= T;
				if (interior)
					return (cur[i - 1] >> (j - 1)) & 1ull;
				const float v = L[i][j];
				return v >= (float)k && v < (float)(k + 1);
			};
			bool anyMember = false;
			uint64_t cand[T];
			for (int i = 0; i < T; i++) {
				uint64_t m = 0;
				for (int j = 0; j < T; j++) {
					if ((closed[i] >> j) & 1ull)
						continue;
					const int pi = i + 1, pj = j + 1;
					bool reach = member(pi, pj - 1) || member(pi, pj + 1) || member(pi - 1, pj) || member(pi + 1, pj);
					for (int di = -1; di <= 1 && !reach; di += 2)
						for (int dj = -1; dj <= 1; dj += 2)
							if (member(pi + di, pj + dj) && !(O[pi + di][pj] && O[pi][pj + dj]))
								reach = true;
					if (reach)
						m |= 1ull << j;
				}
				cand[i] = m;
			}
			for (int i = 0; i < T + 2 && !anyMember; i++)
				for (int j = 0; j < T + 2; j++)
					if (member(i, j)) {
						anyMember = true;
						break;
					}
			st.rounds++;
			st.maskOps++;
			if (!anyMember)
				st.emptyRounds++;
			int maxPer = 0;
			for (int i = 0; i < T; i++) {
				const int pcnt = __builtin_popcountll(cand[i]);
				if (pcnt > maxPer)
					maxPer = pcnt;
				uint64_t m = cand[i];
				while (m) {
					const int j = __builtin_ctzll(m);
					m &= m - 1;
					const int pi = i + 1, pj = j + 1;
					float minS = kInf, minD = kInf;
					minS = std::fmin(std::fmin(L[pi][pj - 1], L[pi][pj + 1]), std::fmin(L[pi - 1][pj], L[pi + 1][pj]));
					for (int di = -1; di <= 1; di += 2)
						for (int dj = -1; dj <= 1; dj += 2)
							if (!(O[pi + di][pj] && O[pi][pj + dj]))
								minD = std::fmin(minD, L[pi + di][pj + dj]);
					if (minS == minD)
						st.tie = 1; // a straight and a diagonal neighbour tie for the minimum: the pop order decides
					const float v = minS <= minD ? minS + 1.0f : minD + kDiag;
					L[pi][pj] = v;
					closed[i] |= 1ull << j;
					if (v < (float)(k + 2))
						nx1[i] |= 1ull << j;
					else
						nx2[i] |= 1ull << j;
					st.cells++;
				}
			}
			st.passes += maxPer;
			bool more = false, open = false;
			for (int i = 0; i < T; i++) {
				cur[i] = nx1[i];
				nx1[i] = nx2[i];
				nx2[i] = 0;
				more |= (cur[i] | nx1[i]) != 0;
				open |= closed[i] != ~0ull;
			}
			if (!open)
				break; // every free cell of the tile has its cost
			if (!more && (float)(k + 1) > hmax)
				break; // no member left in any later bucket
		}
	writeback:
		// changed border cells re-queue the neighbouring tiles
		const bool first = !visited[tr * TC + tc];
		for (int i = 0; i < T; i++)
			for (int j = 0; j < T; j++) {
				const int r = r0 + i, c = c0 + j;
				if (r >= rows || c >= cols)
					continue;
				const float v = L[i + 1][j + 1];
				const float old = first ? kInf : cost[(size_t)r * cols + c];
				cost[(size_t)r * cols + c] = v;
				if ((i == 0 || i == T - 1 || j == 0 || j == T - 1) && std::memcmp(&v, &old, 4) != 0) {
					const float p = v < old ? v : old;
					// which neighbouring tiles can see the change: a halo cell n of this tile (= a border cell of the neighbour, as it was
					// when this solve began) next to (i, j) whose parent this cell may be or may have been: n is free and either has
					// no cost yet or one that a parent of cost p could explain (cost[n] = cost[parent] + edge, edge >= 1)
					for (int di = -1; di <= 1; di++)
						for (int dj = -1; dj <= 1; dj++) {
							if (!di && !dj)
								continue;
							const int pi = i + 1 + di, pj = j + 1 + dj; // padded coordinates of n
							if (pi >= 1 && pi <= T && pj >= 1 && pj <= T)
								continue; // interior
							if (O[pi][pj])
								continue;
							if (!(L[pi][pj] == kInf || p < L[pi][pj] - 0.99f))
								continue;
							const int nr = tr + (pi < 1 ? -1 : pi > T ? 1 : 0), nc = tc + (pj < 1 ? -1 : pj > T ? 1 : 0);
							if (nr < 0 || nc < 0 || nr >= TR || nc >= TC)
								continue;
							pending[nr * TC + nc] = 1;
							if (p < prio[nr * TC + nc])
								prio[nr * TC + nc] = p;
						}
				}
			}
		visited[tr * TC + tc] = 1;
		if (++nvis[tr * TC + tc] > st.maxVisits)
			st.maxVisits = nvis[tr * TC + tc];
	}

	void run(int goalR, int goalC, int maxVisitsPerTile)
	{
		TR = (rows + T - 1) / T;
		TC = (cols + T - 1) / T;
		prio.assign((size_t)TR * TC, kInf);
		pending.assign((size_t)TR * TC, 0);
		visited.assign((size_t)TR * TC, 0);
		nvis.assign((size_t)TR * TC, 0);
		st.tiles = TR * TC;
		for (size_t i = 0; i < (size_t)rows * cols; i++)
			cost[i] = kInf;
		if (goalR < 0)
			return;
		pending[(goalR / T) * TC + goalC / T] = 1;
		prio[(goalR / T) * TC + goalC / T] = 0.0f;
		for (;;) {
			int best = -1;
			for (int t = 0; t < TR * TC; t++)
				if (pending[t] && (best < 0 || prio[t] < prio[best]))
					best = t;
			if (best < 0)
				break;
			if (nvis[best] >= maxVisitsPerTile) {
				st.unsettled = 1;
				break;
			}
			pending[best] = 0;
			prio[best] = kInf;
			solve(best / TC, best % TC, goalR, goalC);
		}
	}
};

} // namespace

extern "C" int pp_model_tile_field(const uint8_t* occ, int rows, int cols, int goalRow, int goalCol, float* cost, int64_t* stats, int maxVisitsPerTile)
{
	Model m;
	m.occ = occ;
	m.rows = rows;
	m.cols = cols;
	m.cost = cost;
	m.run(goalRow, goalCol, maxVisitsPerTile > 0 ? maxVisitsPerTile : 64);
	stats[0] = m.st.visits;
	stats[1] = m.st.rounds;
	stats[2] = m.st.passes;
	stats[3] = m.st.cells;
	stats[4] = m.st.emptyRounds;
	stats[5] = m.st.tie;
	stats[6] = m.st.unsettled;
	stats[7] = m.st.tiles;
	stats[8] = m.st.maxVisits;
	return m.st.tie || m.st.unsettled;
}
