"""-m gpu: batched Hybrid-A* graph search against the CPU oracle: expanded-cell sequence bit-exact,
path poses and cost within 1e-5, RNG-gated Reeds-Shepp expansion in step with std::mt19937_64."""
import math

import numpy as np
import pytest

import oracle_lib as O
from gpu_common import make_pair, valid_random_poses

pytestmark = pytest.mark.gpu


def run_pair(w, ms, val, params_kw, starts, goals, seeds, max_nodes=32768, table=None, alias=True, negk=True, search_rows=0):
    import pathplanning_amd as pa
    P = pa.HybridAStarSearchParameters(heading_alias=alias, negative_k_read=negk, **params_kw)
    planner = pa.HybridAStarBatch(val, P, max_batch=len(starts), max_nodes=max_nodes, search_rows=search_rows)
    planner.initialize(table)
    res = planner.search_batch(starts, goals, seeds)
    h = O.Hybrid(w, O.params_array(**params_kw), heading_alias=alias, negative_k_read=negk, table=planner.nonholo_table())
    return planner, res, h


def compare(planner, res, h, starts, goals, seeds):
    n_ok = 0
    for q in range(len(starts)):
        r = h.search(starts[q], goals[q], int(seeds[q]))
        g = res[q]
        assert g.status == r["status"], (q, g.status, r["status"])
        exp = planner.get_expanded_of(q)
        assert g.n_expanded == len(r["expanded"]), (q, g.n_expanded, len(r["expanded"]))
        assert np.array_equal(exp, r["expanded"]), q
        assert g.n_nodes == r["n_nodes"], q
        assert g.n_rng_draws == r["n_rng_draws"], q
        assert g.n_rs_attempts == r["n_rs_attempts"], q
        assert g.n_state_checks == r["n_state_checks"], (q, g.n_state_checks, r["n_state_checks"])
        assert g.n_path_checks == r["n_path_checks"], (q, g.n_path_checks, r["n_path_checks"])
        assert g.n_lattice_boundary_hits == r["n_lattice_boundary_hits"], (q, g.n_lattice_boundary_hits, r["n_lattice_boundary_hits"])
        if r["status"] == 0:
            n_ok += 1
            assert abs(g.cost - r["cost"]) < 1e-5
            path = planner.get_path_of(q)
            assert len(path["poses"]) == len(r["path_poses"])
            assert np.abs(path["poses"] - r["path_poses"]).max() < 1e-5
            assert np.array_equal(path["kind"], r["path_kind"])
            assert np.abs(path["length"] - r["path_length"]).max() < 1e-5
            rsn = path["kind"] == 2
            assert np.array_equal(path["prim"][rsn], r["path_rsword"][rsn])
    return n_ok


def test_smoke_case_from_reference_test():
    """planner/tests/test_hybrid_a_star.cpp:9-36 on the device."""
    import pathplanning_amd as pa
    w = O.World(10.0, 10.0, 0.1)
    w.update()
    ctx = pa.Context(0)
    ms = pa.OccupancyMapSet.from_bounds(ctx, w.lb, w.ub, 0.1)
    ms.upload_dist2(w.d2())
    ms.upload_occupancy(w.occ())
    ms.upload_path_cost(w.pathcost())
    val = pa.StateValidatorOccupancyMap(ms)
    planner = pa.HybridAStarBatch(val)
    assert planner.search_path() == pa.Status.FAILURE  # not initialised (hybrid_a_star.cpp:243-246)
    planner.initialize()
    planner.set_init_state([0.0, 0.0, 0.0])
    planner.set_goal_state([8.0, 8.0, 0.78])
    assert planner.search_path() == pa.Status.SUCCESS
    path = planner.get_path()
    assert len(path) >= 2
    assert np.hypot(*(path[0][:2])) < 0.1
    assert np.hypot(*(path[-1][:2] - np.array([8.0, 8.0]))) < 0.1
    assert abs(path[-1][2] - 0.78) < math.radians(5)


def test_batch_parity_256():
    w, ms, val, ctx = make_pair(256, 6, 3)
    rng = np.random.RandomState(11)
    n = 24
    starts = valid_random_poses(rng, w, n)
    goals = valid_random_poses(rng, w, n)
    goals[0] = starts[0]  # start == goal
    goals[1] = [100.0, 0.0, 0.0]  # goal outside the map: obstacle field stays +inf, search fails
    seeds = np.arange(n, dtype=np.uint64) + 100
    planner, res, h = run_pair(w, ms, val, {}, starts, goals, seeds)
    n_ok = compare(planner, res, h, starts, goals, seeds)
    assert n_ok >= n // 2


def test_batch_parity_512_more_primitives_and_costs():
    w, ms, val, ctx = make_pair(512, 12, 1)
    rng = np.random.RandomState(12)
    n = 8
    starts = valid_random_poses(rng, w, n)
    goals = valid_random_poses(rng, w, n)
    starts[0] = [-23.04, -23.04, 0.0]
    goals[0] = [23.04, 23.04, 0.0]  # SURVEY 8(d) config 2 query
    seeds = np.full(n, 12345, dtype=np.uint64)
    kw = dict(num_generated_motion=9, reverse_cost_multiplier=2.0, direction_switching_cost=0.3, voronoi_cost_multiplier=0.5)
    planner, res, h = run_pair(w, ms, val, kw, starts, goals, seeds, max_nodes=65536)
    assert compare(planner, res, h, starts, goals, seeds) >= 4


def test_quirk_switches_off():
    """'fixed' mode (no heading aliasing, wrapped heading bin in the table) also matches the oracle in the same mode."""
    w, ms, val, ctx = make_pair(256, 6, 3)
    rng = np.random.RandomState(13)
    n = 6
    starts = valid_random_poses(rng, w, n)
    goals = valid_random_poses(rng, w, n)
    seeds = np.arange(n, dtype=np.uint64)
    planner, res, h = run_pair(w, ms, val, {}, starts, goals, seeds, alias=False, negk=False, max_nodes=131072)
    compare(planner, res, h, starts, goals, seeds)


def test_node_capacity_is_reported():
    import pathplanning_amd as pa
    w, ms, val, ctx = make_pair(256, 6, 3)
    planner = pa.HybridAStarBatch(val, max_batch=1, max_nodes=64)
    planner.initialize()
    res = planner.search_batch([[-11.0, -11.0, 0.0]], [[11.0, 11.0, 0.0]], [1])
    assert res[0].status == -4


def test_four_queries_per_wave_kernel_parity(monkeypatch):
    """PP_SEARCH_ROWS=1: the experimental kernel with one query per 16-lane row (pp_planner_rows.hpp) must walk
    exactly the same expansions -- including P > 16 primitives (two child batches per expansion)."""
    monkeypatch.setenv("PP_SEARCH_ROWS", "1")
    w, ms, val, ctx = make_pair(256, 6, 3)
    rng = np.random.RandomState(21)
    n = 21  # not a multiple of 4: the last wave has idle rows
    starts = valid_random_poses(rng, w, n)
    goals = valid_random_poses(rng, w, n)
    goals[2] = starts[2]
    goals[3] = [100.0, 0.0, 0.0]
    seeds = np.arange(n, dtype=np.uint64) + 7
    planner, res, h = run_pair(w, ms, val, {}, starts, goals, seeds)
    assert compare(planner, res, h, starts, goals, seeds) >= n // 2
    kw = dict(num_generated_motion=9, reverse_cost_multiplier=2.0, direction_switching_cost=0.3, voronoi_cost_multiplier=0.5)
    planner, res, h = run_pair(w, ms, val, kw, starts[:6], goals[:6], seeds[:6], max_nodes=65536)
    compare(planner, res, h, starts[:6], goals[:6], seeds[:6])


def test_search_rows_are_reused_by_successive_queries(monkeypatch):
    """A throughput-sized planner (max_batch > 64 selects the rows kernel by itself) with only 8 rows: every row runs
    ~9 queries one after the other in the same node / heap / key-map buffers; results, expansion logs and paths (copied
    out by the kernel when a query ends) must still be the oracle's."""
    monkeypatch.delenv("PP_SEARCH_ROWS", raising=False)
    w, ms, val, ctx = make_pair(256, 6, 3)
    rng = np.random.RandomState(31)
    n = 70
    starts = valid_random_poses(rng, w, n)
    goals = valid_random_poses(rng, w, n)
    seeds = np.arange(n, dtype=np.uint64) + 1000
    planner, res, h = run_pair(w, ms, val, {}, starts, goals, seeds, search_rows=8)
    assert compare(planner, res, h, starts, goals, seeds) >= n // 2
    # the same planner again: buffers of the previous batch are recycled
    res2 = planner.search_batch(starts[::-1].copy(), goals[::-1].copy(), seeds[::-1].copy())
    assert [r.n_expanded for r in res2] == [r.n_expanded for r in res][::-1]


def test_probable_longest_queries_run_one_per_wave(monkeypatch):
    """PP_SEARCH_DIRECT=k: the first k queries of the hand-out order (ranked by the wavefront kernel) are searched by the
    one-query-per-wave kernel on a second stream while the rows kernel works on the rest; every query must still be the
    oracle's, whichever kernel ran it, and a second batch on the same planner must reuse the slots cleanly."""
    monkeypatch.delenv("PP_SEARCH_ROWS", raising=False)
    monkeypatch.setenv("PP_SEARCH_DIRECT", "12")
    w, ms, val, ctx = make_pair(256, 6, 3)
    rng = np.random.RandomState(77)
    n = 70
    starts = valid_random_poses(rng, w, n)
    goals = valid_random_poses(rng, w, n)
    seeds = np.arange(n, dtype=np.uint64) + 4000
    planner, res, h = run_pair(w, ms, val, {}, starts, goals, seeds, search_rows=8)
    assert compare(planner, res, h, starts, goals, seeds) >= n // 2
    res2 = planner.search_batch(starts[::-1].copy(), goals[::-1].copy(), seeds[::-1].copy())
    assert [r.n_expanded for r in res2] == [r.n_expanded for r in res][::-1]


def test_row_primitives_selftest():
    """DPP row shifts / butterflies / bpermute reads used by the four-queries-per-wave kernel, against scalar loops."""
    import subprocess
    from pathplanning_amd import build
    exe = build.build_row_test(verbose=False)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "row primitives OK" in out.stdout


def test_config5_4096_map_queries():
    """BASELINE config 5 scale: 4096 x 4096 cells (409.6 m), non-holonomic table 411 x 411 x 73 built on the device,
    obstacle heuristic on the 16.7 M-cell grid, RS analytic expansion on.  Distances come from an exact EDT here (the
    distance grid is an input of the path; any consistent grid pins parity)."""
    import pathplanning_amd as pa
    from scipy import ndimage
    w = O.World(204.8, 204.8, 0.1)
    rng = np.random.RandomState(5)
    occ = np.full((w.rows, w.cols), -1, dtype=np.int32)
    for _ in range(60):
        r0, c0 = rng.randint(200, 3500, 2)
        h, wd = rng.randint(80, 400, 2)
        occ[r0:r0 + h, c0] = 0
        occ[r0, c0:c0 + wd] = 0
    d = ndimage.distance_transform_edt(occ != 0)
    d2 = np.minimum(np.rint(d * d), 2 ** 30).astype(np.int32)
    w.set_occ(occ)
    w.set_d2(d2)
    w.set_pathcost(np.zeros((w.rows, w.cols), dtype=np.float32))
    ctx = pa.Context(0)
    ms = pa.OccupancyMapSet.from_bounds(ctx, w.lb, w.ub, 0.1)
    ms.upload_dist2(d2)
    ms.upload_occupancy(occ)
    ms.upload_path_cost(np.zeros((w.rows, w.cols), dtype=np.float32))
    val = pa.StateValidatorOccupancyMap(ms)
    # (not on lattice boundaries: a goal at exact multiples of the 1 m spatial resolution makes the cell of the RS child
    # -- PathReedsShepp's m_final = Interpolate(1.0), path_reeds_shepp.cpp:9 -- depend on the last ulp of sin/cos)
    starts = np.array([[-150.3, -150.2, 0.3], [10.4, 20.7, -2.0], [180.6, -170.1, 1.0]])
    goals = np.array([[-120.7, -100.4, 1.2], [60.37, 45.21, 0.5], [150.3, -120.6, 2.0]])
    ok_s = w.is_state_valid(starts)
    ok_g = w.is_state_valid(goals)
    assert ok_s.all() and ok_g.all()
    seeds = np.array([1, 2, 3], dtype=np.uint64)
    planner, res, h = run_pair(w, ms, val, {}, starts, goals, seeds, max_nodes=65536)
    assert planner.nonholo_table().shape[:2] == (411, 411)
    assert compare(planner, res, h, starts, goals, seeds) >= 2


def test_config5_4096_map_reference_order_fields_16_queries():
    """Config 5 with the fields the drop-in builds by default: 96 rectangle outlines rasterised through the product, GVD::Update in
    REFERENCE ORDER (a non-zero path-cost grid: the Voronoi term of every edge cost is live), 16 random queries -- against the oracle planning
    on ITS OWN brushfire's grids.  Status, expansion sequence, counters, cost and path of every query."""
    import pathplanning_amd as pa
    from pathplanning_amd import synthetic
    ctx = pa.Context(0)
    m, info = synthetic.make_map_product(ctx, 4096, 96, 11, reference_order=True)
    assert float(m["path_cost"].max()) > 0.0
    w = O.synthetic_world(4096, 96, 11)  # the same outlines through the oracle's own rasteriser and brushfire
    assert np.array_equal(w.occ() >= 0, m["occ"] >= 0)
    ms, val = synthetic.upload(ctx, m)
    n = 16
    # (random poses: off the lattice lines; connected to the bulk of the free space for the robot, so that no query has to exhaust the
    # 1.2 M cells of this map's lattice -- the planner's node buffers here hold 262 144)
    reach = synthetic.reachable_mask(val, m)
    starts = synthetic.sample_valid_poses(val, m, n, seed=12, reachable=reach)
    goals = synthetic.sample_valid_poses(val, m, n, seed=13, reachable=reach)
    seeds = np.arange(n, dtype=np.uint64) + 31
    planner, res, h = run_pair(w, ms, val, {}, starts, goals, seeds, max_nodes=262144)
    assert all(r.status in (0, -1) for r in res), [r.status for r in res]
    h.set_max_expansions(262144)
    assert compare(planner, res, h, starts, goals, seeds) >= 8


def test_long_queries_are_handed_over_to_the_one_query_kernel(monkeypatch):
    """Rows kernel in three stages: queries beyond 40 expansions are set aside (open list flushed into the heap, scalars
    in a SuspendRec) and continued by a second pass of the rows kernel in the same slot; beyond 150 expansions the
    one-query-per-wave kernel finishes them.  Only 6 spare slots first, so most long queries stay in the first pass.
    Expansion logs, counters and paths must not notice."""
    monkeypatch.setenv("PP_SEARCH_ROWS", "1")
    monkeypatch.setenv("PP_SEARCH_SUSPEND_AFTER", "40")    # first pass of the rows kernel
    monkeypatch.setenv("PP_SEARCH_SUSPEND_AFTER2", "150")  # second pass; beyond: one query per wave
    monkeypatch.setenv("PP_SEARCH_EXTRA_SLOTS", "6")
    w, ms, val, ctx = make_pair(256, 6, 3)
    rng = np.random.RandomState(41)
    n = 30
    starts = valid_random_poses(rng, w, n)
    goals = valid_random_poses(rng, w, n)
    seeds = np.arange(n, dtype=np.uint64) + 500
    planner, res, h = run_pair(w, ms, val, {}, starts, goals, seeds, search_rows=8)
    assert sum(1 for r in res if r.n_expanded > 40) > 10 and sum(1 for r in res if r.n_expanded > 150) > 3
    assert compare(planner, res, h, starts, goals, seeds) >= n // 2
    monkeypatch.setenv("PP_SEARCH_EXTRA_SLOTS", "64")  # every long query handed over
    planner, res, h = run_pair(w, ms, val, {}, starts, goals, seeds, search_rows=8)
    assert compare(planner, res, h, starts, goals, seeds) >= n // 2
    # compaction: waves with an empty queue and <= 2 busy rows re-queue their queries for the second pass
    monkeypatch.setenv("PP_SEARCH_COMPACT", "2")
    monkeypatch.setenv("PP_SEARCH_SUSPEND_AFTER", "100000")
    monkeypatch.setenv("PP_SEARCH_SUSPEND_AFTER2", "0")
    planner, res, h = run_pair(w, ms, val, {}, starts, goals, seeds, search_rows=8)
    assert compare(planner, res, h, starts, goals, seeds) >= n // 2


def test_handles_may_be_destroyed_in_any_order():
    """A map keeps its context alive and a planner its map: destroying the context and the map first (what a garbage
    collector may do with a reference cycle) must leave the planner usable."""
    w, ms, val, ctx = make_pair(256, 6, 3)
    import pathplanning_amd as pa
    planner = pa.HybridAStarBatch(val, pa.HybridAStarSearchParameters(), max_batch=2, max_nodes=16384)
    planner.initialize()
    lib = planner.lib
    lib.pp_ctx_destroy(ctx.h)
    ctx.h = None
    lib.pp_map_destroy(ms.h)
    ms.h = None
    starts = np.array([[-10.0, -10.0, 0.0], [9.0, -9.0, 1.0]])
    goals = np.array([[10.0, 10.0, 0.0], [-9.0, 8.0, -2.0]])
    res = planner.search_batch(starts, goals, np.array([7, 8], dtype=np.uint64))
    assert res[0].n_expanded > 0 and res[1].n_expanded > 0
    planner.close()


def test_context_idle_query_is_non_blocking():
    """pp_ctx_is_idle: false while a batch enqueued with search_batch_dev is still running (or at the latest true once
    fetch_results has synchronised), true on an idle stream; the batch's results are unaffected by the polling."""
    import torch
    w, ms, val, ctx = make_pair(256, 6, 3)
    import pathplanning_amd as pa
    assert ctx.is_idle()
    n = 40
    rng = np.random.RandomState(5)
    starts = valid_random_poses(rng, w, n)
    goals = valid_random_poses(rng, w, n)
    seeds = np.arange(n, dtype=np.uint64) + 9
    planner = pa.HybridAStarBatch(val, pa.HybridAStarSearchParameters(), max_batch=n, max_nodes=32768)
    planner.initialize()
    ref = planner.search_batch(starts, goals, seeds)
    dev = torch.device("cuda", 0)
    ds, dg = torch.from_numpy(starts).to(dev), torch.from_numpy(goals).to(dev)
    dseed = torch.from_numpy(seeds.astype(np.int64)).to(dev)
    planner.search_batch_dev(ds, dg, dseed)
    polls = 0
    while not ctx.is_idle():
        polls += 1
        assert polls < 10_000_000
    res = planner.fetch_results()
    assert ctx.is_idle()
    assert [r.n_expanded for r in res] == [r.n_expanded for r in ref]
    planner.close()


@pytest.mark.parametrize("rows_kernel", ["0", "1"])
def test_config2_query_at_74_primitives_both_kernels(monkeypatch, rows_kernel):
    """BASELINE config 2 at the reference-reachable primitive count next to "72": numGeneratedMotion = 37 -> P = 74
    (hybrid_a_star.cpp:21-28).  The one-query kernel walks the children in two passes of 64 lanes (`base += 64`), the rows
    kernel in five batches of 16; both must produce the oracle's expansion sequence, counters, cost and path."""
    monkeypatch.setenv("PP_SEARCH_ROWS", rows_kernel)
    w, ms, val, ctx = make_pair(512, 12, 1)
    starts = np.array([[-23.04, -23.04, 0.0], [20.3, -21.7, 2.0]])
    goals = np.array([[23.04, 23.04, 0.0], [-20.1, 19.4, -1.0]])
    assert w.is_state_valid(starts).all() and w.is_state_valid(goals).all()
    seeds = np.array([12345, 12346], dtype=np.uint64)
    planner, res, h = run_pair(w, ms, val, dict(num_generated_motion=37), starts, goals, seeds, max_nodes=131072)
    assert planner.num_primitives == 74 and h.P == 74
    assert compare(planner, res, h, starts, goals, seeds) >= 1
    assert res[0].status == 0 and res[0].n_expanded > 200


@pytest.mark.parametrize("rows_kernel", ["0", "1"])
def test_config2_query_with_an_explicit_table_of_72_primitives(monkeypatch, rows_kernel):
    """BASELINE config 2 names 72 motion primitives; hybrid_a_star.cpp:21-28 can only generate 2 * odd.  pp_planner_set_primitives
    takes the steering-angle list itself (36 angles, forward + backward each); the oracle gets the same list in place of m_deltas.
    Expansion sequence, counters, cost and path as for every other primitive count, through both search kernels."""
    monkeypatch.setenv("PP_SEARCH_ROWS", rows_kernel)
    import pathplanning_amd as pa
    w, ms, val, ctx = make_pair(512, 12, 1)
    starts = np.array([[-23.04, -23.04, 0.0], [20.3, -21.7, 2.0]])
    goals = np.array([[23.04, 23.04, 0.0], [-20.1, 19.4, -1.0]])
    seeds = np.array([12345, 12346], dtype=np.uint64)
    P = pa.HybridAStarSearchParameters()
    delta_max = math.atan(P.wheelbase / P.min_turning_radius)
    deltas = np.linspace(-delta_max, delta_max, 36)
    planner = pa.HybridAStarBatch(val, P, max_batch=2, max_nodes=131072)
    planner.set_primitives(deltas)
    assert planner.num_primitives == 72
    planner.initialize()
    res = planner.search_batch(starts, goals, seeds)
    h = O.Hybrid(w, O.params_array(), table=planner.nonholo_table())
    h.set_deltas(deltas)
    assert h.P == 72
    assert compare(planner, res, h, starts, goals, seeds) >= 1
    assert res[0].status == 0 and res[0].n_expanded > 200
    with pytest.raises(Exception):
        planner.set_primitives(np.zeros(65))  # 130 primitives: beyond the table


def test_planners_until_capacity_error_not_an_abort():
    """Round 1's abort (gpurun_out/b_b2048.log): a planner took the last byte of HBM and the runtime could not allocate the
    scratch k_wavefront needs at its first dispatch -> HSA_STATUS_ERROR_OUT_OF_RESOURCES, core dump.  Now the kernels are
    dispatched once (empty) before the planner's large allocations and the planner leaves a reserve: creating planners
    until the device is full must end in PP_ERR_CAPACITY (-4) with a message, and every planner created must still search."""
    import pathplanning_amd as pa
    from pathplanning_amd._lib import PPError
    w, ms, val, ctx = make_pair(256, 6, 3)
    planners = []
    err = None
    for k in range(64):
        try:
            p = pa.HybridAStarBatch(val, pa.HybridAStarSearchParameters(), max_batch=4096, max_nodes=81920, search_rows=2048)
        except PPError as e:
            err = e
            break
        planners.append(p)
    assert err is not None, "64 throughput-sized planners cannot fit in 288 GB"
    assert err.code == -4 and "reserved" in str(err), (err.code, str(err))
    assert len(planners) >= 2
    rng = np.random.RandomState(3)
    starts = valid_random_poses(rng, w, 80)
    goals = valid_random_poses(rng, w, 80)
    seeds = np.arange(80, dtype=np.uint64)
    first = None
    for p in (planners[0], planners[-1]):  # the last one was created with the least memory left
        p.initialize()
        res = p.search_batch(starts, goals, seeds)
        got = [(r.status, r.n_expanded) for r in res]
        first = first or got
        assert got == first
    for p in planners:
        p.close()


def test_lattice_guard_band_counter():
    """the result of every query says how many of its poses were discretised within 1e-9 cells of a lattice
    boundary (pp_query_result::n_lattice_boundary_hits) -- 0 for random queries (bit-exactness by construction), > 0 when a start
    sits exactly on a multiple of the resolution (where it holds as observed); the count itself matches the oracle's, through
    both search kernels."""
    w, ms, val, ctx = make_pair(192, 4, 31)
    rng = np.random.RandomState(9)
    starts = valid_random_poses(rng, w, 12)
    goals = valid_random_poses(rng, w, 12)
    on_lattice = 0
    for q in range(0, 12, 2):  # every other start: x on a lattice line (1.0 = 1 cell), heading on a bin edge
        cand = np.array([np.round(starts[q][0]), starts[q][1], 0.0])
        if w.is_state_valid(cand[None, :])[0]:
            starts[q] = cand
            on_lattice += 1
    assert on_lattice >= 2
    seeds = np.arange(12, dtype=np.uint64) + 900
    for rows in (0, 8):
        planner, res, h = run_pair(w, ms, val, {}, starts, goals, seeds, search_rows=rows)
        compare(planner, res, h, starts, goals, seeds)
        hits = np.array([r.n_lattice_boundary_hits for r in res])
        assert (hits[1::2] == 0).all(), hits
        assert (hits[0::2] > 0).sum() >= on_lattice, hits


def test_lattice_certification_on_boundary_poses(monkeypatch):
    """SURVEY 7.3 H2 as a contract: starts AND goals on exact multiples of the spatial resolution (x or y an integer number of metres,
    heading a multiple of the angular resolution) -- the case where a last-bit difference between this libm and glibc can put a child on
    the other side of a lattice line (y + (cos t - cos t') / kappa lands on an integer +- 1e-16).  Every such query reports
    n_lattice_boundary_hits > 0; pp_planner_certify_lattice recomputes every created constant-steer node and every logged lattice-line
    child (created or dropped) on the host with glibc.  The contract: EVERY query that deviates from the oracle is caught (a
    mismatching cell, or a flagged event that cannot be recomputed: hand it to the CPU reference); the certified ones equal the oracle
    in every discrete output.  Starts on lattice lines but goals off them, so that not every query ends in an unverifiable
    Reeds-Shepp child."""
    w, ms, val, ctx = make_pair(192, 4, 31)
    rng = np.random.RandomState(21)
    cand_s, cand_g = valid_random_poses(rng, w, 400), valid_random_poses(rng, w, 400)
    ares = 0.0872
    starts, goals = [], []
    for s_, g_ in zip(cand_s, cand_g):
        s2 = np.array([np.round(s_[0]), np.round(s_[1]), np.round(s_[2] / ares) * ares])
        g2 = g_ if len(starts) % 2 else np.array([np.round(g_[0]), g_[1], np.round(g_[2] / ares) * ares])  # every other goal off the lines
        if w.is_state_valid(np.array([s2, g2])).all():
            starts.append(s2)
            goals.append(g2)
        if len(starts) == 24:
            break
    assert len(starts) >= 12
    starts, goals = np.array(starts), np.array(goals)
    seeds = np.arange(len(starts), dtype=np.uint64) + 400
    planner, res, h = run_pair(w, ms, val, {}, starts, goals, seeds, search_rows=0)  # one query per wave: the tree is kept
    clean, caught, total_checked = 0, 0, 0
    for q in range(len(starts)):
        assert res[q].n_lattice_boundary_hits > 0
        r = h.search(starts[q], goals[q], int(seeds[q]))
        g = res[q]
        same = (g.status == r["status"] and g.n_expanded == len(r["expanded"]) and np.array_equal(planner.get_expanded_of(q), r["expanded"]) and g.n_nodes == r["n_nodes"]
                and g.n_rng_draws == r["n_rng_draws"] and g.n_rs_attempts == r["n_rs_attempts"] and g.n_state_checks == r["n_state_checks"] and g.n_path_checks == r["n_path_checks"]
                and (r["status"] != 0 or abs(g.cost - r["cost"]) < 1e-5))
        checked, mismatches, unverified, worst = planner.certify_lattice(q)
        assert worst < 1e-9, (q, worst)  # the poses themselves agree far below the parity tolerance either way
        assert same or mismatches + unverified > 0, (q, "deviates from the oracle, not caught", checked, mismatches, unverified)
        clean += mismatches + unverified == 0
        caught += mismatches + unverified > 0
        total_checked += checked
    print("lattice certification: %d queries on lattice lines, %d certified, %d flagged for the CPU reference, %d nodes recomputed" % (len(starts), clean, caught, total_checked))
    assert total_checked > 1000 and clean >= 1 and clean + caught == len(starts)
    # a throughput planner keeps no tree: the call says so instead of certifying nothing
    monkeypatch.setenv("PP_SEARCH_ROWS", "1")
    rows, res_rows, _ = run_pair(w, ms, val, {}, starts, goals, seeds, search_rows=8)
    assert rows.search_rows == 8
    with pytest.raises(Exception):
        rows.certify_lattice(0)


def test_chained_planners_give_the_same_results():
    """pp_planner_start_after_fields_of is scheduling only: two planners on their own streams, the second one's batch held back until
    the first one's heuristic fields are built, return what they return unchained; chaining a planner to itself is refused."""
    import pathplanning_amd as pa
    from pathplanning_amd import _lib, synthetic
    w, ms, val, ctx = make_pair(192, 4, 31)
    rng = np.random.RandomState(12)
    starts = valid_random_poses(rng, w, 24)
    goals = valid_random_poses(rng, w, 24)
    seeds = np.arange(24, dtype=np.uint64) + 77
    base = pa.HybridAStarBatch(val, pa.HybridAStarSearchParameters(), max_batch=24, max_nodes=32768, search_rows=8)
    base.initialize()
    want = base.search_batch(starts, goals, seeds)
    ctx2 = pa.Context(0)
    m = dict(lower=w.lb, upper=w.ub, resolution=0.1, occ=w.occ(), d2=w.d2(), path_cost=w.pathcost())
    ms2, val2 = synthetic.upload(ctx2, m)
    other = pa.HybridAStarBatch(val2, pa.HybridAStarSearchParameters(), max_batch=24, max_nodes=32768, search_rows=8)
    other.initialize(base.nonholo_table())
    other.start_after_fields_of(base)  # base has a recorded "fields built" event from the batch above
    got = other.search_batch(starts, goals, seeds)
    for a, b in zip(want, got):
        assert (a.status, a.n_expanded, a.n_nodes, a.n_path) == (b.status, b.n_expanded, b.n_nodes, b.n_path) and (a.cost == b.cost or a.status != 0)
    with pytest.raises(_lib.PPError):
        other.start_after_fields_of(other)
