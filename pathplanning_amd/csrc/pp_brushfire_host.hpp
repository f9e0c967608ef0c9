// Reference-order mode of GVD::Update ("needs a parity mode vs Lau brushfire").
//
// The reference's two distance maps (state_validator/gvd.cpp:30-72 ObstacleDistanceMap::Update, :200-237
// VoronoiDistanceMap::Update, :74-89 / :239-255 Set / Unset, :105-131 CheckVoro) are Lau's dynamic brushfire driven by a
// std::priority_queue whose comparator looks at the key only (utils/grid.h:51-61).  Among equal keys the pop order is
// whatever libstdc++'s push_heap / pop_heap leave on top, and that order decides which source a tie keeps and what its
// neighbours inherit -- so the grids can only be reproduced bit for bit by running the same heap over the same sequence of
// pushes.  That is what this file does, on the host: one generic sweep (`Brushfire`) instantiated twice, the obstacle
// map's Voronoi test as a hook.  It is a sequential algorithm by definition (every pop depends on the previous one), which
// is why this mode lives next to the device code rather than in it; the device mode (exact Euclidean transform,
// pp_gvd.hip) is the throughput path.  The state persists between updates, so AddObstacle / RemoveObstacle after the
// first build cost only the cells they disturb (the reference's raise / lower waves), not a rebuild.
//
// Same libstdc++ as the reference's build => same heap order; the element type does not matter to push_heap / pop_heap,
// only the sequence of comparisons, which depends on keys alone.
#pragma once

#include <climits>
#include <cstdint>
#include <cstdlib>
#include <queue>
#include <vector>

namespace pph {

/// GridCellPosition::GetNeighbors (utils/grid.cpp:16-50): left column first (centre, below, above), then the right
/// column, then the two cells of the own column; cells outside the grid are left out
inline int brushfire_neighbours(int r, int c, int rows, int cols, int out[8])
{
	static const int dR[8] = { 0, -1, 1, 0, -1, 1, -1, 1 }, dC[8] = { -1, -1, -1, 1, 1, 1, 0, 0 };
	int n = 0;
	for (int j = 0; j < 8; j++) {
		const int rr = r + dR[j], cc = c + dC[j];
		if (rr >= 0 && rr < rows && cc >= 0 && cc < cols)
			out[n++] = rr * cols + cc;
	}
	return n;
}

/// One dynamic-brushfire distance map.  Cells are row * cols + col; `source[i]` is the cell the distance of i is measured
/// from (-1: none), `dist[i]` its squared distance (INT_MAX: none).
class Brushfire {
public:
	Brushfire(int rows_, int cols_) :
		rows(rows_), cols(cols_), dist((size_t)rows_ * cols_, INT_MAX), source((size_t)rows_ * cols_, -1), raise((size_t)rows_ * cols_, 0), pending((size_t)rows_ * cols_, 0)
	{
	}
	const int rows, cols;
	std::vector<int32_t> dist, source;

	bool is_source(int cell) const { return cell >= 0 && source[cell] == cell; } // IsOccupied, gvd.cpp:186-189 / :257-260
	/// SetObstacle / SetEdge (gvd.cpp:74-80, :239-245)
	void set(int cell)
	{
		source[cell] = cell;
		dist[cell] = 0;
		open.push({ cell, 0 });
		pending[cell] = 1;
	}
	/// UnsetObstacle / UnsetEdge (gvd.cpp:82-89, :247-255)
	void unset(int cell)
	{
		dist[cell] = INT_MAX;
		source[cell] = -1;
		raise[cell] = 1;
		open.push({ cell, INT_MAX });
		pending[cell] = 1;
	}
	int sq(int a, int b) const
	{
		const int dr = a / cols - b / cols, dc = a % cols - b % cols;
		return dr * dr + dc * dc;
	}
	/// Update (gvd.cpp:30-72, :200-237).  `lowered(s)` is called when a cell that still has its source is popped (the obstacle
	/// map clears the cell's Voronoi mark there), `tie(s, n)` for a neighbour the popped cell cannot improve (CheckVoro).
	template <class Lowered, class Tie>
	long long update(Lowered lowered, Tie tie)
	{
		long long pops = 0;
		int nb[8];
		while (!open.empty()) {
			const int s = open.top().cell;
			open.pop();
			pops++;
			if (!pending[s])
				continue;
			const int sr = s / cols, sc = s % cols;
			if (raise[s]) {
				const int k = brushfire_neighbours(sr, sc, rows, cols, nb);
				for (int q = 0; q < k; q++) {
					const int n = nb[q];
					if (source[n] >= 0 && !raise[n]) {
						if (!is_source(source[n])) {
							dist[n] = INT_MAX;
							source[n] = -1;
							raise[n] = 1;
						}
						open.push({ n, dist[n] });
						pending[n] = 1;
					}
				}
				raise[s] = 0;
			} else if (is_source(source[s])) {
				lowered(s);
				pending[s] = 0;
				const int k = brushfire_neighbours(sr, sc, rows, cols, nb);
				for (int q = 0; q < k; q++) {
					const int n = nb[q];
					if (raise[n])
						continue;
					const int d = sq(source[s], n);
					if (d < dist[n]) {
						dist[n] = d;
						source[n] = source[s];
						open.push({ n, d });
						pending[n] = 1;
					} else {
						tie(s, n);
					}
				}
			}
		}
		return pops;
	}
	bool idle() const { return open.empty(); }

private:
	struct Entry {
		int32_t cell, key;
	};
	struct Later { // std::greater<GridCell<int>>: the key alone (utils/grid.h:57-61)
		bool operator()(const Entry& a, const Entry& b) const { return a.key > b.key; }
	};
	std::vector<uint8_t> raise, pending;
	std::priority_queue<Entry, std::vector<Entry>, Later> open;
};

/// The pair of maps GVD::Update sweeps, with the host copy of the occupancy ids CheckVoro compares.
struct GvdReference {
	GvdReference(int rows, int cols) :
		obstacles(rows, cols), edges(rows, cols), occ((size_t)rows * cols, -1) { }
	Brushfire obstacles, edges;
	std::vector<int32_t> occ; // obstacle id per cell, -1 free (obstacle_list_occupancy_map.cpp:29-61)
	long long pops = 0;

	/// AddObstacle's / RemoveObstacle's loop body for one boundary cell (value >= 0: id written, SetObstacle; < 0: -1, Unset)
	void edit(int cell, int32_t value)
	{
		occ[cell] = value >= 0 ? value : -1;
		if (value >= 0)
			obstacles.set(cell);
		else
			obstacles.unset(cell);
	}
	/// CheckVoro, gvd.cpp:105-131
	void check_voro(int s, int n)
	{
		const int oS = obstacles.source[s], oN = obstacles.source[n];
		if (oN < 0)
			return; // (unreachable from update(): a neighbour that cannot be improved has a finite distance, hence a source)
		if (occ[oS] == occ[oN])
			return;
		const int cols = obstacles.cols;
		if (!(obstacles.dist[s] > 1 || obstacles.dist[n] > 1))
			return;
		if (!(std::abs(oS / cols - oN / cols) > 1 || std::abs(oS % cols - oN % cols) > 1))
			return;
		const int sStability = obstacles.sq(s, oN) - obstacles.dist[s], nStability = obstacles.sq(n, oS) - obstacles.dist[n];
		if (sStability < 0 || nStability < 0)
			return;
		if (sStability <= nStability)
			edges.set(s);
		if (nStability <= sStability)
			edges.set(n);
	}
	/// GVD::Update's two sweeps, gvd.cpp:294-301 (the path-cost map is elementwise and runs on the device)
	void update()
	{
		pops += obstacles.update([&](int s) { edges.unset(s); }, [&](int s, int n) { check_voro(s, n); });
		pops += edges.update([](int) {}, [](int, int) {});
	}
};

} // namespace pph
