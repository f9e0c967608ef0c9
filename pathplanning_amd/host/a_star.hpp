// Host-side best-first search engine of the plugin surface (BASELINE config 1: "CPU, no GPU").
// Same names and observable behaviour as the reference's
//   algo/a_star.h:15-144        AStarStatePropagator, AStarHeuristic, AStarConcreteHeuristic(Fcn), AStarCombinedHeuristic,
//                               AStarHeuristicAdapter
//   algo/a_star.h:213-441       AStar (SearchPath, Expand, ProcessPossibleShortcut, GetPath, GetActions, GetOptimalCost)
//   algo/bidirectional_a_star.h AverageHeuristic, BidirectionalAStar
//   algo/a_star_n2.{h,cpp}      AStarStatePropagatorFcnN2, AStarHeuristicFcnN2, AStarN2, BidirectionalAStarN2
// but not its data structures: the reference keeps an owning pointer tree and an open list that is a sorted
// std::vector (O(n) insertion, utils/frontier.h:39-48).  Here nodes live in one arena addressed by index, and the open
// list is a binary heap ordered by (totalCost ascending, push sequence descending) with lazy deletion -- which pops in
// exactly the reference's order: Frontier::Push inserts behind every element whose cost is >= the new one and Pop takes
// the back, i.e. lowest cost first and, among equal costs, the most recently pushed first.
#pragma once

#include <cstdint>
#include <functional>
#include <limits>
#include <memory>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "planner_hip.hpp"

namespace Planner {

struct NullAction { };

/// algo/a_star.h:15-27
template <typename State, typename Action = NullAction>
class AStarStatePropagator {
public:
	virtual ~AStarStatePropagator() = default;
	/// (neighbour state, action, transition cost) of every valid transition out of `state`
	virtual std::vector<std::tuple<State, Action, double>> GetNeighborStates(const State& state) = 0;
};

/// algo/a_star.h:30-49 (SetGoal is protected + friend-accessed in the reference; public here, a superset)
template <typename State>
class AStarHeuristic {
public:
	virtual ~AStarHeuristic() = default;
	virtual double GetHeuristicValue(const State& state) = 0;
	virtual void SetGoal(const State& goal) = 0;
};

/// algo/a_star.h:53-64
template <typename State>
class AStarConcreteHeuristic : public AStarHeuristic<State> {
public:
	void SetGoal(const State& goal) override { m_goal = goal; }

protected:
	State m_goal;
};

/// algo/a_star.h:67-81: h(state) = func(state, goal)
template <typename State, typename Func>
class AStarConcreteHeuristicFcn : public AStarConcreteHeuristic<State> {
public:
	explicit AStarConcreteHeuristicFcn(Func func) : m_func(func) { }
	double GetHeuristicValue(const State& state) override { return m_func(state, this->m_goal); }

protected:
	Func m_func;
};

/// algo/a_star.h:85-120: max over admissible heuristics (-inf when empty)
template <typename State>
class AStarCombinedHeuristic : public AStarHeuristic<State> {
public:
	void Add(const Ref<AStarHeuristic<State>>& h) { m_parts.push_back(h); }
	template <typename... More>
	void Add(const Ref<AStarHeuristic<State>>& h, More&&... more)
	{
		m_parts.push_back(h);
		Add(std::forward<More>(more)...);
	}
	double GetHeuristicValue(const State& state) override
	{
		double best = -std::numeric_limits<double>::infinity();
		for (auto& h : m_parts)
			best = std::max(best, h->GetHeuristicValue(state));
		return best;
	}
	void SetGoal(const State& goal) override
	{
		for (auto& h : m_parts)
			h->SetGoal(goal);
	}

private:
	std::vector<Ref<AStarHeuristic<State>>> m_parts;
};

/// algo/a_star.h:124-144: heuristic over S2 used by a search over S1 through a state conversion
template <typename S1, typename S2, typename Func>
class AStarHeuristicAdapter : public AStarHeuristic<S1> {
public:
	AStarHeuristicAdapter(const Ref<AStarHeuristic<S2>>& inner, Func convert) : m_inner(inner), m_convert(convert) { }
	double GetHeuristicValue(const S1& state) override { return m_inner->GetHeuristicValue(m_convert(state)); }
	void SetGoal(const S1& goal) override { m_inner->SetGoal(m_convert(goal)); }

private:
	Ref<AStarHeuristic<S2>> m_inner;
	Func m_convert;
};

/// The engine.  GraphSearch = true: a state is expanded at most once, a child whose state is already open replaces
/// the open node only when its totalCost is strictly lower (a_star.h:391-402, 417-427; Appendix A Q2).
template <typename State, typename Action = NullAction, typename HashState = std::hash<State>, typename EqualState = std::equal_to<State>, bool GraphSearch = true>
class AStar : public PathPlanner<State> {
	static_assert(std::is_copy_constructible<State>::value, "State must be copyable");

public:
	using StateSet = std::unordered_set<State, HashState, EqualState>;

	AStar() = default;

	/// a_star.h:315-323
	bool Initialize(const Ref<AStarStatePropagator<State, Action>>& propagator, const Ref<AStarHeuristic<State>>& heuristic)
	{
		if (!propagator || !heuristic)
			return m_ready = false;
		m_propagator = propagator;
		m_heuristic = heuristic;
		return m_ready = true;
	}

	/// a_star.h:326-346
	Status SearchPath() override
	{
		if (!m_ready)
			return Status::Failure; // "The algorithm has not been initialized successfully."
		InitializeSearch();
		while (!OpenEmpty()) {
			const int n = PopOpen();
			if (IsSolution(m_nodes[n].state)) {
				m_solution = n;
				return Status::Success;
			}
			Expand(n);
		}
		return Status::Failure;
	}

	/// a_star.h:254-269: states root .. solution
	std::vector<State> GetPath() const override { return Chain<State>(m_solution, [this](int n) { return m_nodes[n].state; }, true); }
	/// a_star.h:272-288: one action per edge
	std::vector<Action> GetActions() const { return Chain<Action>(m_solution, [this](int n) { return m_nodes[n].action; }, false); }
	/// a_star.h:294-299
	double GetOptimalCost() const { return m_solution < 0 ? std::numeric_limits<double>::infinity() : m_nodes[m_solution].pathCost; }
	/// a_star.h:291 (as a set of states; the root counts as explored from the start, a_star.h:361)
	StateSet GetExploredStates() const
	{
		StateSet s;
		for (const auto& kv : m_explored)
			s.insert(kv.first);
		return s;
	}
	/// expansion order (not in the reference: what the parity tests compare)
	const std::vector<State>& GetExpansionOrder() const { return m_expansionOrder; }
	const Ref<AStarStatePropagator<State, Action>>& GetStatePropagator() const { return m_propagator; }
	const Ref<AStarHeuristic<State>>& GetHeuristic() const { return m_heuristic; }

protected:
	template <typename S, typename A, typename H, typename E, bool G>
	friend class BidirectionalAStar;

	struct Rec {
		State state;
		Action action;
		double pathCost, totalCost;
		int parent;
		bool open; // still the live open-list entry of its state
	};
	struct OpenKey {
		double cost;
		uint64_t seq;
		int node;
	};
	struct OpenLater { // heap top = lowest cost, then the latest push
		bool operator()(const OpenKey& a, const OpenKey& b) const { return a.cost > b.cost || (a.cost == b.cost && a.seq < b.seq); }
	};

	virtual bool IsSolution(const State& s) { return EqualState()(s, this->m_goal); } // a_star.h:368-373

	/// a_star.h:350-364
	void InitializeSearch()
	{
		m_nodes.clear();
		m_heap.clear();
		m_openOf.clear();
		m_explored.clear();
		m_expansionOrder.clear();
		m_solution = -1;
		m_seq = 0;
		m_nodes.push_back({ this->m_init, Action(), 0.0, 0.0, -1, false });
		PushOpen(0);
		m_explored.emplace(this->m_init, 0);
		m_heuristic->SetGoal(this->m_goal);
	}

	/// a_star.h:377-409
	void Expand(int n)
	{
		m_explored.emplace(m_nodes[n].state, n); // first insertion wins, as unordered_map::insert
		m_expansionOrder.push_back(m_nodes[n].state);
		const State from = m_nodes[n].state;
		const double g = m_nodes[n].pathCost;
		for (auto& [childState, action, transitionCost] : m_propagator->GetNeighborStates(from)) {
			const double pathCost = g + transitionCost;
			const double totalCost = pathCost + m_heuristic->GetHeuristicValue(childState);
			auto open = m_openOf.find(childState);
			if (GraphSearch) {
				if (open == m_openOf.end()) {
					if (m_explored.find(childState) == m_explored.end())
						PushOpen(NewNode(childState, action, pathCost, totalCost, n));
				} else if (m_nodes[open->second].totalCost > totalCost) { // ProcessPossibleShortcut, a_star.h:417-427
					m_nodes[open->second].open = false; // Frontier::Remove; the old node stays behind as a dead leaf
					m_openOf.erase(open);
					PushOpen(NewNode(childState, action, pathCost, totalCost, n));
				}
			} else if (open == m_openOf.end()) { // tree search: Frontier::Push still refuses a second element with an equal key
				PushOpen(NewNode(childState, action, pathCost, totalCost, n));
			}
		}
	}

	int NewNode(const State& s, const Action& a, double pathCost, double totalCost, int parent)
	{
		m_nodes.push_back({ s, a, pathCost, totalCost, parent, false });
		return (int)m_nodes.size() - 1;
	}
	void PushOpen(int n)
	{
		m_nodes[n].open = true;
		m_openOf.emplace(m_nodes[n].state, n);
		m_heap.push_back({ m_nodes[n].totalCost, m_seq++, n });
		std::push_heap(m_heap.begin(), m_heap.end(), OpenLater());
	}
	bool OpenEmpty()
	{
		while (!m_heap.empty() && !m_nodes[m_heap.front().node].open) { // drop entries replaced by a shortcut
			std::pop_heap(m_heap.begin(), m_heap.end(), OpenLater());
			m_heap.pop_back();
		}
		return m_heap.empty();
	}
	int TopOpen() { return OpenEmpty() ? -1 : m_heap.front().node; }
	int PopOpen()
	{
		const int n = TopOpen();
		std::pop_heap(m_heap.begin(), m_heap.end(), OpenLater());
		m_heap.pop_back();
		m_nodes[n].open = false;
		m_openOf.erase(m_nodes[n].state);
		return n;
	}
	template <typename T, typename Get>
	std::vector<T> Chain(int leaf, Get get, bool withRoot) const
	{
		std::vector<T> out;
		for (int n = leaf; n >= 0 && (withRoot || m_nodes[n].parent >= 0); n = m_nodes[n].parent)
			out.push_back(get(n));
		std::reverse(out.begin(), out.end());
		return out;
	}

	Ref<AStarStatePropagator<State, Action>> m_propagator;
	Ref<AStarHeuristic<State>> m_heuristic;
	std::vector<Rec> m_nodes;
	std::vector<OpenKey> m_heap;
	std::unordered_map<State, int, HashState, EqualState> m_openOf; // state -> its live open node
	std::unordered_map<State, int, HashState, EqualState> m_explored; // state -> node (the reference's ExploredMap)
	std::vector<State> m_expansionOrder;
	int m_solution = -1;
	uint64_t m_seq = 0;

private:
	bool m_ready = false;
};

/// algo/bidirectional_a_star.h:10-39.  As in the reference, SetGoal / Update only store this object's own states: the two
/// wrapped heuristics keep whatever goal they were last given (e.g. by an earlier unidirectional search that shared them).
template <typename State>
class AverageHeuristic : public AStarConcreteHeuristic<State> {
public:
	AverageHeuristic(const Ref<AStarHeuristic<State>>& toGoal, const Ref<AStarHeuristic<State>>& toInit) : m_toGoal(toGoal), m_toInit(toInit) { }
	double GetHeuristicValue(const State& s) override { return m_constant + (m_toGoal->GetHeuristicValue(s) - m_toInit->GetHeuristicValue(s)) / 2.0; }
	/// init / goal in the direction of the search that owns this heuristic
	void Update(const State& init, const State& goal)
	{
		m_init = init;
		this->m_goal = goal;
		m_constant = m_toInit->GetHeuristicValue(goal) / 2.0;
	}

private:
	Ref<AStarHeuristic<State>> m_toGoal, m_toInit;
	State m_init;
	double m_constant = 0.0;
};

/// algo/bidirectional_a_star.h:42-204: two searches stepped alternately; stops when the open lists' best path costs
/// cannot improve the best meeting found (`fTop + rTop >= best + offset`).
template <typename State, typename Action = NullAction, typename HashState = std::hash<State>, typename EqualState = std::equal_to<State>, bool GraphSearch = true>
class BidirectionalAStar : public PathPlanner<State> {
	using Search = AStar<State, Action, HashState, EqualState, GraphSearch>;

public:
	using StateSet = typename Search::StateSet;

	/// bidirectional_a_star.h:58-63
	static std::tuple<Ref<AStarHeuristic<State>>, Ref<AStarHeuristic<State>>> GetAverageHeuristicPair(const Ref<AStarHeuristic<State>>& fHeuristic, const Ref<AStarHeuristic<State>>& rHeuristic)
	{
		Ref<AStarHeuristic<State>> f = makeRef<AverageHeuristic<State>>(fHeuristic, rHeuristic);
		Ref<AStarHeuristic<State>> r = makeRef<AverageHeuristic<State>>(rHeuristic, fHeuristic);
		return std::make_tuple(f, r);
	}

	/// bidirectional_a_star.h:115-127
	bool Initialize(const Ref<AStarStatePropagator<State, Action>>& fPropagator, const Ref<AStarStatePropagator<State, Action>>& rPropagator,
		const Ref<AStarHeuristic<State>>& fHeuristic, const Ref<AStarHeuristic<State>>& rHeuristic)
	{
		if (!m_f.Initialize(fPropagator, fHeuristic) || !m_r.Initialize(rPropagator, rHeuristic))
			return m_ready = false;
		m_fAverage = dynamic_cast<AverageHeuristic<State>*>(fHeuristic.get());
		m_rAverage = dynamic_cast<AverageHeuristic<State>*>(rHeuristic.get());
		return m_ready = true;
	}

	/// bidirectional_a_star.h:130-178
	Status SearchPath() override
	{
		if (!m_ready)
			return Status::Failure;
		if (m_fAverage && m_rAverage) {
			m_fAverage->Update(this->m_init, this->m_goal);
			m_rAverage->Update(this->m_goal, this->m_init);
		}
		m_f.SetInitState(this->m_init);
		m_f.SetGoalState(this->m_goal);
		m_r.SetInitState(this->m_goal);
		m_r.SetGoalState(this->m_init);
		m_f.InitializeSearch();
		m_r.InitializeSearch();
		const double offset = m_f.m_heuristic->GetHeuristicValue(this->m_goal) + m_r.m_heuristic->GetHeuristicValue(this->m_goal);
		double best = std::numeric_limits<double>::infinity();
		while (!m_f.OpenEmpty() && !m_r.OpenEmpty()) {
			Step(m_f, m_r, best);
			Step(m_r, m_f, best);
			if (m_f.m_solution >= 0 && m_r.m_solution >= 0) {
				if (m_f.OpenEmpty() || m_r.OpenEmpty())
					return Status::Success;
				if (m_f.m_nodes[m_f.TopOpen()].pathCost + m_r.m_nodes[m_r.TopOpen()].pathCost >= best + offset)
					return Status::Success;
			}
		}
		return Status::Failure;
	}

	/// bidirectional_a_star.h:66-72: forward path + reversed reverse path; the meeting state appears twice (Appendix A Q16)
	std::vector<State> GetPath() const override
	{
		auto path = m_f.GetPath();
		auto back = m_r.GetPath();
		path.insert(path.end(), back.rbegin(), back.rend());
		return path;
	}
	std::tuple<StateSet, StateSet> GetExploredStates() const { return std::make_tuple(m_f.GetExploredStates(), m_r.GetExploredStates()); }
	std::tuple<std::vector<State>, std::vector<State>> GetExpansionOrders() const { return std::make_tuple(m_f.GetExpansionOrder(), m_r.GetExpansionOrder()); }
	double GetOptimalCost() const { return m_f.GetOptimalCost() + m_r.GetOptimalCost(); }

private:
	/// pop + expand one node of `a`, then look its state up among the states `b` has explored (:155-157, 181-196)
	static void Step(Search& a, Search& b, double& best)
	{
		const int n = a.PopOpen();
		a.Expand(n);
		auto hit = b.m_explored.find(a.m_nodes[n].state);
		if (hit == b.m_explored.end())
			return;
		const double through = a.m_nodes[n].pathCost + b.m_nodes[hit->second].pathCost;
		if (through < best) {
			best = through;
			a.m_solution = n;
			b.m_solution = hit->second;
		}
	}

	Search m_f, m_r;
	AverageHeuristic<State>*m_fAverage = nullptr, *m_rAverage = nullptr;
	bool m_ready = false;
};

} // namespace Planner

namespace std {
template <>
struct hash<Planner::GridCellPosition> { // utils/grid.h:30-41 hashes (row, col) with HashCombine; any injective mix serves the same purpose
	size_t operator()(const Planner::GridCellPosition& c) const { return (size_t)(((uint64_t)(uint32_t)c.row << 32) | (uint32_t)c.col) * 0x9E3779B97F4A7C15ull; }
};
}

namespace Planner {

using CellCostFcn = std::function<double(const GridCellPosition&, const GridCellPosition&)>;

/// algo/a_star_n2.cpp:12-28: 8-connected moves over an occupancy map in the fixed neighbour order of utils/grid.cpp:29-47
/// (Appendix A Q4); a diagonal move is refused only when BOTH orthogonal cells it cuts between are occupied.
class AStarStatePropagatorFcnN2 : public AStarStatePropagator<GridCellPosition, NullAction> {
public:
	AStarStatePropagatorFcnN2(const Ref<OccupancyMap>& map, const CellCostFcn& pathCostFcn) : m_map(map), m_cost(pathCostFcn) { }
	std::vector<std::tuple<GridCellPosition, NullAction, double>> GetNeighborStates(const GridCellPosition& cell) override
	{
		std::vector<std::tuple<GridCellPosition, NullAction, double>> out;
		out.reserve(8);
		for (const GridCellPosition& n : cell.GetNeighbors(m_map->Rows(), m_map->Columns())) {
			if (m_map->IsOccupied(n))
				continue;
			if (n.IsDiagonalTo(cell) && m_map->IsOccupied({ n.row, cell.col }) && m_map->IsOccupied({ cell.row, n.col }))
				continue;
			out.emplace_back(n, NullAction(), m_cost(cell, n));
		}
		return out;
	}

private:
	Ref<OccupancyMap> m_map;
	CellCostFcn m_cost;
};

/// Many (init, goal) pairs at once on the device (pp_grid_astar_batch, one wave per query): AStarN2 / BidirectionalAStarN2 with
/// AStarStatePropagatorFcnN2 over the map and the transition cost / heuristic of the reference's own script
/// (example_a_star_grid.py:46-52: Euclidean distance between the cells).  Same paths, costs and expansion orders as the
/// host engine above with those functions; any other function needs that engine (per-edge callbacks, as in the reference).
struct GridSearchResult {
	Status status = Status::Failure;
	double cost = std::numeric_limits<double>::infinity(); // GetOptimalCost
	std::vector<GridCellPosition> path;                    // GetPath (bidirectional: the meeting cell twice, Appendix A Q16)
	std::vector<GridCellPosition> expanded, expandedReverse; // expansion order = GetExploredStates as a sequence
};
class GridAStarBatchHip {
public:
	explicit GridAStarBatchHip(const Ref<OccupancyMap>& map) : m_map(map) { }
	/// wantExpanded: also return the expansion orders ([n][rows * columns] cells of buffer per direction: for small batches / maps);
	/// innerGoals (bidirectional only, optional): per query the goals held by the two heuristics the AverageHeuristic pair wraps
	/// {forward, reverse}; default forward -> goal, reverse -> init
	std::vector<GridSearchResult> SearchBatch(const std::vector<GridCellPosition>& inits, const std::vector<GridCellPosition>& goals, bool bidirectional = false,
		bool wantExpanded = false, const std::vector<std::pair<GridCellPosition, GridCellPosition>>* innerGoals = nullptr)
	{
		if (inits.size() != goals.size() || (innerGoals && innerGoals->size() != inits.size()))
			throw std::invalid_argument("GridAStarBatchHip::SearchBatch: one goal (and one inner-goal pair) per init");
		const int n = (int)inits.size();
		std::vector<GridSearchResult> out((size_t)n);
		if (n == 0)
			return out;
		const int rows = m_map->Rows(), cols = m_map->Columns();
		const int maxPath = std::min<long long>((long long)rows * cols + 1, 4ll * (rows + cols));
		const int maxExpanded = wantExpanded ? rows * cols : 0;
		std::vector<int32_t> ic((size_t)n * 2), gc((size_t)n * 2), ig;
		for (int i = 0; i < n; i++) {
			ic[2 * i] = inits[i].row, ic[2 * i + 1] = inits[i].col;
			gc[2 * i] = goals[i].row, gc[2 * i + 1] = goals[i].col;
		}
		if (innerGoals) {
			ig.resize((size_t)n * 4);
			for (int i = 0; i < n; i++) {
				ig[4 * i] = (*innerGoals)[i].first.row, ig[4 * i + 1] = (*innerGoals)[i].first.col;
				ig[4 * i + 2] = (*innerGoals)[i].second.row, ig[4 * i + 3] = (*innerGoals)[i].second.col;
			}
		}
		std::vector<pp_grid_result> res((size_t)n);
		std::vector<int32_t> paths((size_t)n * maxPath * 2), exp((size_t)n * maxExpanded * 2), expR(bidirectional ? (size_t)n * maxExpanded * 2 : 0);
		ppCheck(pp_grid_astar_batch(m_map->Device(), n, ic.data(), gc.data(), bidirectional ? 1 : 0, innerGoals ? ig.data() : nullptr, maxPath, maxExpanded, res.data(),
			paths.data(), wantExpanded ? exp.data() : nullptr, wantExpanded && bidirectional ? expR.data() : nullptr));
		auto cells = [](const int32_t* p, int count) {
			std::vector<GridCellPosition> v;
			v.reserve((size_t)count);
			for (int k = 0; k < count; k++)
				v.emplace_back(p[2 * k], p[2 * k + 1]);
			return v;
		};
		for (int i = 0; i < n; i++) {
			if (res[i].n_path > maxPath)
				throw std::runtime_error("GridAStarBatchHip: path longer than 4 * (rows + columns) cells");
			out[i].status = res[i].status == 0 ? Status::Success : Status::Failure;
			out[i].cost = res[i].cost;
			out[i].path = cells(&paths[(size_t)i * maxPath * 2], res[i].n_path);
			if (wantExpanded) {
				out[i].expanded = cells(&exp[(size_t)i * maxExpanded * 2], res[i].n_expanded);
				if (bidirectional)
					out[i].expandedReverse = cells(&expR[(size_t)i * maxExpanded * 2], res[i].n_expanded_reverse);
			}
		}
		return out;
	}

private:
	Ref<OccupancyMap> m_map;
};

using AStarHeuristicFcnN2 = AStarConcreteHeuristicFcn<GridCellPosition, CellCostFcn>; // a_star_n2.h:26
using PathPlannerN2Base = PathPlanner<GridCellPosition>; // path_planner.h:43
using AStarN2 = AStar<GridCellPosition, NullAction>; // a_star_n2.h:29-34
using BidirectionalAStarN2 = BidirectionalAStar<GridCellPosition, NullAction>; // a_star_n2.h:37-42

} // namespace Planner
