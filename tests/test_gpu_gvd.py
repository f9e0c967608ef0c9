"""-m gpu: map authoring and field construction on the device against the oracle's restatement of
state_validator/obstacle.cpp, obstacle_list_occupancy_map.cpp and gvd.cpp (the reference's dynamic brushfire with the same
libstdc++ priority queue).  Exact: outline rasterisation, PathCostMap::Update, and -- in the reference-order mode of
pp_map_update_gvd_ex -- every grid of GVD::Update (d2, nearest obstacle cell, Voronoi edges, Voronoi d2, nearest edge cell, path
cost), also after incremental AddObstacle / RemoveObstacle.  The throughput mode (exact Euclidean transform on the device) is
compared with scipy's EDT (equal) and with the brushfire (bounded: the brushfire over-estimates a few tie cells)."""
import json
import math
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rect_vertices(dx, dy):
    return [(dx / 2.0, dy / 2.0), (-dx / 2.0, dy / 2.0), (-dx / 2.0, -dy / 2.0), (dx / 2.0, -dy / 2.0)]  # RectangleShape, obstacle.cpp:106-110


def circle_vertices(radius, count):
    radius *= 1.0 / math.cos(math.pi / count)  # CircleShape, obstacle.cpp:112-122 (the angle goes through a float division)
    return [(radius * math.cos(2 * math.pi * i / float(np.float32(count))), radius * math.sin(2 * math.pi * i / float(np.float32(count)))) for i in range(count)]


def build_pair(n_cells, shapes, resolution=0.1):
    """The same obstacles in the oracle world (AddObstacle) and on the device (host vertices, device Bresenham)."""
    import pathplanning_amd as pa
    half = n_cells * resolution / 2.0
    w = O.World(half, half, resolution)
    ctx = pa.Context(0)
    ms = pa.OccupancyMapSet.from_bounds(ctx, w.lb, w.ub, resolution)
    for k, (kind, a, b, pose) in enumerate(shapes):
        if kind == "rect":
            w.add_rectangle(a, b, pose)
            ms.add_polygon(rect_vertices(a, b), pose, k)
        else:
            w.add_circle(a, b, pose)
            ms.add_polygon(circle_vertices(a, b), pose, k)
    return w, ms, ctx


def seeded_shapes(n_cells, n, seed, resolution=0.1):
    half = n_cells * resolution / 2.0
    rng = np.random.RandomState(seed)
    out = []
    for k in range(n):
        x, y = rng.uniform(-0.7 * half, 0.7 * half, 2)
        th = rng.uniform(-math.pi, math.pi)
        out.append(("rect", 0.3 * half, 0.04 * half, [x, y, th]) if k % 3 else ("circle", 0.08 * half, 10, [x, y, th]))
    return out


def test_outline_rasterisation_is_exact_and_removable():
    w, ms, ctx = build_pair(256, seeded_shapes(256, 9, 5) + [("rect", 8.0, 1.0, [12.0, 12.0, 0.3])])  # the last one sticks out of the map
    occ = ms.download_occupancy()
    assert np.array_equal(occ, w.occ())
    assert (occ >= 0).sum() > 500 and occ.max() == 9
    # RemoveObstacle: the boundary cells go back to -1 (also where another outline crossed them, as in the reference)
    ms.add_polygon(rect_vertices(8.0, 1.0), [12.0, 12.0, 0.3], -1)
    assert (ms.download_occupancy() == 9).sum() == 0


def test_path_cost_update_is_bit_exact():
    import pathplanning_amd as pa
    for cells, nobs, seed in ((256, 6, 3), (512, 12, 1)):
        w = O.synthetic_world(cells, nobs, seed)
        ctx = pa.Context(0)
        ms = pa.OccupancyMapSet.from_bounds(ctx, w.lb, w.ub, 0.1)
        got = ms.path_cost_update(w.d2(), w.voro_d2())
        want = w.pathcost()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert (want > 0).mean() > 0.5


def assert_fields_equal_the_brushfire(ms, w, where):
    g = ms.download_gvd()
    no, ne = O.world_nearest(w)
    assert np.array_equal(g["d2"], w.d2()), where
    assert np.array_equal(g["nearest_obstacle"], no), where
    assert np.array_equal(g["voronoi_d2"], w.voro_d2()), where
    assert np.array_equal(g["nearest_edge"], ne), where
    assert np.array_equal(g["voronoi_edge"].astype(bool), w.voro_d2() == 0), where
    assert np.array_equal(g["path_cost"].view(np.uint32), w.pathcost().view(np.uint32)), where
    return g


@pytest.mark.parametrize("cells,n,seed", [(256, 6, 3), (512, 12, 1), (1024, 24, 1)])
def test_reference_order_mode_equals_the_brushfire_bit_for_bit(cells, n, seed):
    """SURVEY 8f rank 1 ('needs a parity mode vs Lau brushfire'): PP_GVD_REFERENCE_ORDER returns the reference's grids exactly --
    the same std::priority_queue over the same sequence of SetObstacle calls (gvd.cpp:30-89, 105-131, 200-255)."""
    w, ms, ctx = build_pair(cells, seeded_shapes(cells, n, seed))
    w.update()
    pops = ms.update_gvd(mode=ms.GVD_REFERENCE_ORDER)
    g = assert_fields_equal_the_brushfire(ms, w, "first build")
    assert pops > cells * cells and (g["voronoi_edge"] != 0).sum() > cells
    # the validator reads the freshly built grids
    import pathplanning_amd as pa
    val = pa.StateValidatorOccupancyMap(ms)
    rng = np.random.RandomState(1)
    half = float(w.ub[0])
    poses = np.column_stack([rng.uniform(-half, half, 20000), rng.uniform(-half, half, 20000), rng.uniform(-3.1, 3.1, 20000)])
    assert np.array_equal(val.is_state_valid(poses), w.is_state_valid(poses).astype(bool))


def test_reference_order_mode_incremental_add_and_remove():
    """gvd.cpp:74-89: SetObstacle / UnsetObstacle after the first build run the brushfire's raise and lower waves from the state the
    previous Update left -- both sides keep that state, so the grids stay identical through a sequence of edits, and an edit
    costs far fewer heap pops than the first build."""
    cells = 512
    shapes = seeded_shapes(cells, 10, 7)
    w, ms, ctx = build_pair(cells, shapes)
    w.update()
    first = ms.update_gvd(mode=ms.GVD_REFERENCE_ORDER)
    assert_fields_equal_the_brushfire(ms, w, "first build")
    half = cells * 0.1 / 2.0
    extra = [(0.25 * half, 0.05 * half, [0.31 * half, -0.22 * half, 0.4]), (0.2 * half, 0.03 * half, [-0.4 * half, 0.35 * half, -1.1])]
    ids, total = [], first
    for k, (dx, dy, pose) in enumerate(extra):  # AddObstacle
        ident = w.add_rectangle(dx, dy, pose)
        ids.append(ident)
        ms.add_polygon(rect_vertices(dx, dy), pose, ident)
        w.update()
        pops = ms.update_gvd(mode=ms.GVD_REFERENCE_ORDER)
        assert_fields_equal_the_brushfire(ms, w, "after adding %d" % k)
        assert pops - total < first // 2  # incremental: a fraction of the first sweep
        total = pops
    # RemoveObstacle of the first added one (its outline crosses nothing else here), then of an original rectangle
    w.remove_rectangle(ids[0], extra[0][0], extra[0][1], extra[0][2])
    ms.add_polygon(rect_vertices(extra[0][0], extra[0][1]), extra[0][2], -1)
    w.update()
    ms.update_gvd(mode=ms.GVD_REFERENCE_ORDER)
    assert_fields_equal_the_brushfire(ms, w, "after removing the added rectangle")
    k = next(i for i, sh in enumerate(shapes) if sh[0] == "rect")
    w.remove_rectangle(k, shapes[k][1], shapes[k][2], shapes[k][3])
    ms.add_polygon(rect_vertices(shapes[k][1], shapes[k][2]), shapes[k][3], -1)
    w.update()
    ms.update_gvd(mode=ms.GVD_REFERENCE_ORDER)
    assert_fields_equal_the_brushfire(ms, w, "after removing an original rectangle")
    assert np.array_equal(ms.download_occupancy(), w.occ())


def test_gvd_update_against_the_brushfire():
    from scipy import ndimage
    report = {}
    for cells, n, seed in ((256, 6, 3), (512, 12, 1), (1024, 24, 1)):
        w, ms, ctx = build_pair(cells, seeded_shapes(cells, n, seed))
        w.update()
        assert np.array_equal(ms.download_occupancy(), w.occ())
        steps = ms.update_gvd()
        g = ms.download_gvd()
        d2, ref = g["d2"].astype(np.int64), w.d2().astype(np.int64)
        edt = ndimage.distance_transform_edt(w.occ() < 0)
        assert np.array_equal(d2, np.rint(edt * edt).astype(np.int64))  # an exact transform
        assert (d2 <= ref).all()  # the brushfire's values are distances to real obstacle cells: never below the minimum
        diff = d2 != ref
        assert diff.mean() < 2e-3 and (np.sqrt(ref) - np.sqrt(d2)).max() < 0.05  # a twentieth of a cell at most, on < 0.2 % of the cells
        # the label is a nearest obstacle cell
        lab = g["nearest_obstacle"]
        rr, cc = np.meshgrid(np.arange(cells), np.arange(cells), indexing="ij")
        assert np.array_equal((lab[..., 0] - rr) ** 2 + (lab[..., 1] - cc) ** 2, d2)
        assert (w.occ()[lab[..., 0], lab[..., 1]] >= 0).all()
        # Voronoi edges: CheckVoro on final labels vs the brushfire's incremental marks
        edge, ref_edge = g["voronoi_edge"].astype(bool), w.voro_d2() == 0
        both = (edge & ref_edge).sum()
        iou = both / max(1, (edge | ref_edge).sum())
        assert iou > 0.97, (cells, iou)
        vd = np.sqrt(g["voronoi_d2"].astype(np.float64))
        vr = np.sqrt(w.voro_d2().astype(np.float64))
        assert np.abs(vd - vr).mean() < 0.1  # cells
        # path cost: identical bits wherever both of its inputs agree with the brushfire's, close elsewhere
        pc, ref_pc = g["path_cost"], w.pathcost()
        same_in = (~diff) & (g["voronoi_d2"] == w.voro_d2())
        assert np.array_equal(pc[same_in].view(np.uint32), ref_pc[same_in].view(np.uint32))
        # a Voronoi mark that differs moves the potential of the cells around it (voroDist / (obstDist + voroDist) jumps at an
        # edge cell): few cells, bounded on average
        assert np.abs(pc - ref_pc).mean() < 2e-3 and (np.abs(pc - ref_pc) > 0.02).mean() < 0.02
        report[str(cells)] = dict(device_passes=steps, d2_cells_differing=int(diff.sum()), d2_max_diff=int((ref - d2).max()), cells=cells * cells,
                                  voronoi_edges_device=int(edge.sum()), voronoi_edges_brushfire=int(ref_edge.sum()), voronoi_edges_common=int(both),
                                  voronoi_d2_equal_fraction=float((g["voronoi_d2"] == w.voro_d2()).mean()), path_cost_bits_equal_fraction=float((pc.view(np.uint32) == ref_pc.view(np.uint32)).mean()),
                                  path_cost_max_abs_diff=float(np.abs(pc - ref_pc).max()), path_cost_mean_abs_diff=float(np.abs(pc - ref_pc).mean()))
        # the validator reads the freshly built grids: same verdicts as the oracle wherever the distance cell agrees
        import pathplanning_amd as pa
        val = pa.StateValidatorOccupancyMap(ms)
        rng = np.random.RandomState(1)
        half = float(w.ub[0])
        poses = np.column_stack([rng.uniform(-half, half, 20000), rng.uniform(-half, half, 20000), rng.uniform(-3.1, 3.1, 20000)])
        cell = w.to_cell(poses[:, :2])
        agree = ~diff[np.clip(cell[:, 0], 0, cells - 1), np.clip(cell[:, 1], 0, cells - 1)]
        assert np.array_equal(val.is_state_valid(poses)[agree], w.is_state_valid(poses).astype(bool)[agree])
    print("gvd parity:", json.dumps(report))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(report, open(os.path.join(ROOT, "gpurun_out", "gvd_parity.json"), "w"), indent=1)


def test_planning_on_product_built_fields_against_the_reference_built_fields():
    """End to end without any host-built grid: outlines -> device occupancy -> GVD::Update -> Hybrid A*, against the oracle planning on
    ITS OWN brushfire fields (not on a download of the product's).  Reference-order mode: every query identical (expansion sequence,
    cost) -- the fields are the reference's.  Exact-transform mode: the queries whose outcome changes because a few cells in ten
    thousand differ are counted (profiles/r03_fields_mode_vs_outcomes.json) and must be few."""
    import pathplanning_amd as pa
    from gpu_common import valid_random_poses
    cells = 512
    w, ms, ctx = build_pair(cells, seeded_shapes(cells, 12, 11))
    w.update()
    rng = np.random.RandomState(2)
    n = 96
    starts, goals = valid_random_poses(rng, w, n), valid_random_poses(rng, w, n)
    seeds = np.arange(n, dtype=np.uint64) + 5
    table, _ = O.nonholo_build(w.lb, w.ub, O.params_array())
    h = O.Hybrid(w, table=table)
    want = [h.search(starts[q], goals[q], int(seeds[q])) for q in range(n)]
    assert sum(r["status"] == 0 for r in want) >= n // 2

    def differing(mode):
        ms.update_gvd(mode=mode)
        val = pa.StateValidatorOccupancyMap(ms)
        planner = pa.HybridAStarBatch(val, pa.HybridAStarSearchParameters(), max_batch=n, max_nodes=65536)
        planner.initialize(table)
        res = planner.search_batch(starts, goals, seeds)
        bad = [q for q in range(n) if not (res[q].status == want[q]["status"] and np.array_equal(planner.get_expanded_of(q), want[q]["expanded"])
                                           and (want[q]["status"] != 0 or abs(res[q].cost - want[q]["cost"]) < 1e-5))]
        planner.close()
        return bad

    bad_ref = differing(ms.GVD_REFERENCE_ORDER)
    g = ms.download_gvd()
    bad_edt = differing(ms.GVD_EXACT_EDT)
    g2 = ms.download_gvd()
    line = dict(map="%d^2, 12 outlines" % cells, queries=n, queries_differing_on_reference_order_fields=len(bad_ref), queries_differing_on_exact_transform_fields=len(bad_edt),
                d2_cells_differing_between_modes=int((g["d2"] != g2["d2"]).sum()), path_cost_cells_differing_between_modes=int((g["path_cost"].view(np.uint32) != g2["path_cost"].view(np.uint32)).sum()),
                first_differing=bad_edt[:8])
    print("fields mode vs outcomes:", json.dumps(line))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(line, open(os.path.join(ROOT, "gpurun_out", "fields_mode_vs_outcomes.json"), "w"))
    assert bad_ref == []
    assert len(bad_edt) <= n // 4, line
