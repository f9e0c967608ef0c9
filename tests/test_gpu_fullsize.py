"""-m gpu: BASELINE configs[3] at full size (4096 queries on one 1024 x 1024 map) through properties that do not need
the oracle for every query: permutation invariance, path geometry and validity, cost bookkeeping -- plus exact oracle
agreement on a random sample.  Exercises the throughput configuration of the planner (rows kernel, queries taken
longest-first, hand-over of very long queries)."""
import math

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def arc_end(p, kappa, d, backward):
    """KinematicBicycleModel::ConstantSteer with rearToCenter = 0 (models/kinematic_bicycle_model.cpp:5-32)."""
    x, y, t = p
    dd = -d if backward else d
    if abs(kappa) <= 1e-9:
        return np.array([x + dd * math.cos(t), y + dd * math.sin(t), t])
    t1 = t + dd * kappa
    return np.array([x + (math.sin(t1) - math.sin(t)) / kappa, y - (math.cos(t1) - math.cos(t)) / kappa, t1])


def test_full_size_batch_properties():
    import pathplanning_amd as pa
    from pathplanning_amd import synthetic
    B = 4096
    ctx = pa.Context(0)
    m = synthetic.make_map(1024, 24, seed=1)
    ms, val = synthetic.upload(ctx, m)
    params = pa.HybridAStarSearchParameters()
    planner = pa.HybridAStarBatch(val, params, max_batch=B, max_nodes=81920, search_rows=1024)
    assert planner.search_rows == 1024  # the four-queries-per-wave kernel, four queries per row on average
    planner.initialize()
    reach = synthetic.reachable_mask(val, m)
    starts = synthetic.sample_valid_poses(val, m, B, seed=1000, reachable=reach)
    goals = synthetic.sample_valid_poses(val, m, B, seed=2000, reachable=reach)
    seeds = np.arange(B, dtype=np.uint64)
    res = planner.search_batch(starts, goals, seeds)
    status = np.array([r.status for r in res])
    nexp = np.array([r.n_expanded for r in res])
    cost = np.array([r.cost for r in res])
    assert set(np.unique(status)) <= {0, -1}
    assert (status == 0).mean() > 0.95
    assert nexp.max() > 32768  # some queries went through the hand-over to the one-query-per-wave kernel

    # ---- paths: geometry, validity, cost bookkeeping (every solved query)
    half_pi_tol = 1e-3 * math.pi / 180.0
    # primitive curvatures: p = 2 * deltaIndex + direction (hybrid_a_star.cpp:13-29,65-77)
    delta_max = math.atan(params.wheelbase / params.min_turning_radius)
    deltas = [0.0]
    for i in range(params.num_generated_motion // 2):
        d = (i + 1) / 2.0 * delta_max
        deltas += [d, -d]
    arc_len = 1.5 * params.spatial_resolution
    checked_edges = 0
    all_poses = []
    for q in np.nonzero(status == 0)[0]:
        path = planner.get_path_of(int(q))
        poses, kind, prim, length = path["poses"], path["kind"], path["prim"], path["length"]
        assert len(poses) == res[q].n_path >= 1
        assert np.abs(poses[0, :2] - starts[q, :2]).max() < 1e-12 and kind[0] == 0
        assert math.hypot(*(poses[-1, :2] - goals[q, :2])) < 1e-3  # IsSolution: IdenticalPoses with the goal
        dth = (poses[-1, 2] - goals[q, 2] + math.pi) % (2 * math.pi) - math.pi
        assert abs(dth) < half_pi_tol
        total = 0.0
        for i in range(1, len(poses)):
            assert length[i] > 0
            total += length[i]  # forward / reverse multipliers are 1, the Voronoi term is >= 0
            if kind[i] == 1:
                p = int(prim[i])
                assert 0 <= p < planner.num_primitives and length[i] <= arc_len + 1e-12
                kappa = math.tan(deltas[p // 2]) / params.wheelbase
                want = arc_end(poses[i - 1], kappa, length[i], backward=(p % 2 == 1))
                assert np.abs(want[:2] - poses[i, :2]).max() < 1e-9
                dt = (want[2] - poses[i, 2] + math.pi) % (2 * math.pi) - math.pi
                assert abs(dt) < 1e-9
                checked_edges += 1
            else:
                assert kind[i] == 2 and i == len(poses) - 1  # the Reeds-Shepp child is the goal: last edge
        assert cost[q] >= total * (1 - 1e-6)  # (a Reeds-Shepp edge's cost is computed in float, reeds_shepp.cpp:654)
        all_poses.append(poses)
    assert checked_edges > 50000
    assert val.is_state_valid(np.concatenate(all_poses)).all()

    # ---- permutation invariance: the same queries in another order (other rows, other hand-out order)
    perm = np.random.RandomState(3).permutation(B)
    res2 = planner.search_batch(starts[perm].copy(), goals[perm].copy(), seeds[perm].copy())
    for f in ("status", "n_expanded", "n_nodes", "n_rng_draws", "n_rs_attempts", "n_state_checks", "n_path_checks", "n_path"):
        a = np.array([getattr(r, f) for r in res])[perm]
        b = np.array([getattr(r, f) for r in res2])
        assert np.array_equal(a, b), f
    assert np.array_equal(cost[perm], np.array([r.cost for r in res2]))

    # ---- exact agreement with the oracle on a random sample (expansion order included)
    ow = O.World(float(m["upper"][0]), float(m["upper"][1]), m["resolution"])
    ow.set_occ(m["occ"])
    ow.set_d2(m["d2"])
    ow.set_pathcost(m["path_cost"])
    h = O.Hybrid(ow, O.params_array(), table=planner.nonholo_table())
    h.set_max_expansions(20000)
    sample = [int(q) for q in np.random.RandomState(4).permutation(B) if nexp[q] < 20000][:48]
    planner.search_batch(starts, goals, seeds)  # results of the original order again
    for q in sample:
        r = h.search(starts[q], goals[q], int(seeds[q]))
        assert res[q].status == r["status"] and res[q].n_expanded == len(r["expanded"]) and res[q].n_nodes == r["n_nodes"]
        assert np.array_equal(planner.get_expanded_of(q), r["expanded"])
        if r["status"] == 0:
            assert abs(res[q].cost - r["cost"]) < 1e-5
    # ---- the hard cases, at full size: every query that exhausts the lattice (status -1 after ~65 k expansions: the ones the hand-over and
    # the pipeline's urgent ring exist for) and the 16 longest successes -- the whole expansion sequence against the oracle (~1 s of CPU each)
    status = np.array([r.status for r in res])
    failures = [int(q) for q in np.flatnonzero((status != 0) & (nexp > 20000))]
    longest = [int(q) for q in np.argsort(np.where(status == 0, nexp, -1))[-16:]]
    assert 4 <= len(failures) <= 64 and min(nexp[longest]) > 10000
    h.set_max_expansions(1 << 30)
    for q in failures + longest:
        r = h.search(starts[q], goals[q], int(seeds[q]))
        assert res[q].status == r["status"] and res[q].n_expanded == len(r["expanded"]) and res[q].n_nodes == r["n_nodes"], q
        assert np.array_equal(planner.get_expanded_of(q), r["expanded"]), q
        assert res[q].n_rng_draws == r["n_rng_draws"] and res[q].n_rs_attempts == r["n_rs_attempts"], q
        if r["status"] == 0:
            assert abs(res[q].cost - r["cost"]) < 1e-5
            path = planner.get_path_of(q)
            assert len(path["poses"]) == len(r["path_poses"]) and np.abs(path["poses"] - r["path_poses"]).max() < 1e-5


def test_own_table_on_the_benchmark_map():
    """Round 2's own-table test ran on a 256^2 map where device and glibc tables were identical and proved nothing.  Here: the
    benchmark's own 1024^2 / 24-outline map and the first 512 of its queries, the oracle with ITS table (glibc) against the planner
    with its default table (k_nonholo_build, OCML libm) -- what the reference computes is 'own table + own search'.
    Measured: the two tables are IDENTICAL on this map (774 457 entries; asserted, so a future libm change that breaks it shows up
    here) and so are all 512 outcomes.  The failure mode the round-1 verdict named -- entries that differ in their last bit -- is
    then exercised on purpose: one entry in a thousand of the uploaded table is moved by one float ulp (the entries are float-valued
    costs, reeds_shepp.cpp:659) and the outcomes that change are counted and bounded.  Counts go to gpurun_out/own_table_parity_1024.json
    (copied to profiles/)."""
    import json
    import os
    import pathplanning_amd as pa
    from pathplanning_amd import synthetic
    n = 512
    ctx = pa.Context(0)
    m = synthetic.make_map(1024, 24, seed=1)
    ms, val = synthetic.upload(ctx, m)
    reach = synthetic.reachable_mask(val, m)
    starts = synthetic.sample_valid_poses(val, m, 4096, seed=1000, reachable=reach)[:n]
    goals = synthetic.sample_valid_poses(val, m, 4096, seed=2000, reachable=reach)[:n]
    seeds = np.arange(n, dtype=np.uint64)
    ow = O.World(float(m["upper"][0]), float(m["upper"][1]), m["resolution"])
    ow.set_occ(m["occ"])
    ow.set_d2(m["d2"])
    ow.set_pathcost(m["path_cost"])
    own_table, _ = O.nonholo_build(ow.lb, ow.ub, O.params_array())
    _, st, cost, nexp = O.hybrid_batch(ow, own_table, starts, goals, seeds, threads=min(os.cpu_count() or 1, 16))

    def differing(table):
        planner = pa.HybridAStarBatch(val, pa.HybridAStarSearchParameters(), max_batch=n, max_nodes=81920, search_rows=256)
        planner.initialize(table)  # None: the planner's default, built on the device
        used = planner.nonholo_table()
        res = planner.search_batch(starts, goals, seeds)
        planner.close()
        return used, [q for q in range(n) if not (res[q].status == st[q] and res[q].n_expanded == nexp[q] and (st[q] != 0 or abs(res[q].cost - cost[q]) < 1e-5))]

    default_table, bad_default = differing(None)
    n_diff = int((default_table != own_table).sum())
    # last-bit differences on purpose: every 1000th entry one float ulp up or down
    pert = own_table.copy().reshape(-1)
    idx = np.arange(7, pert.size, 1000)
    f = pert[idx].astype(np.float32)
    assert np.array_equal(f.astype(np.float64), pert[idx])  # the entries are float-valued
    pert[idx] = np.nextafter(f, np.where(idx % 2000 < 1000, np.float32(np.inf), np.float32(-np.inf))).astype(np.float64)
    _, bad_pert = differing(pert.reshape(own_table.shape))
    line = dict(map="bench 1024^2, 24 outlines, seed 1", queries=n, table_entries=int(own_table.size), device_table_entries_differing_from_glibc=n_diff,
                queries_differing_with_device_table=len(bad_default), perturbed_entries=int(len(idx)), queries_differing_with_perturbed_table=len(bad_pert),
                first_differing_perturbed=bad_pert[:8])
    print("own-table parity on the bench map:", json.dumps(line))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(line, open(os.path.join(out, "own_table_parity_1024.json"), "w"))
    assert n_diff == 0, "device-built and glibc-built tables differ on the benchmark map: make the host-built table bench.py's default"
    assert bad_default == [], line
    assert len(bad_pert) <= n // 20, line  # one-ulp entries reorder a tie now and then; they must stay rare


def test_full_size_pipeline_replay_equals_the_batch_planner():
    """bench.py's workload through the streaming pipeline, as bench.py drives it: the 4096 benchmark queries submitted eight times over
    with three submissions' worth of slots, so that launches queue, urgent slots are built ahead of their launch and every
    slot is refilled several times.  All 8 x 4096 results must be the batch planner's (status and every counter; cost bit for bit)."""
    import time
    import pathplanning_amd as pa
    from pathplanning_amd import synthetic
    B, rounds = 4096, 8
    ctx = pa.Context(0)
    m = synthetic.make_map(1024, 24, seed=1)
    ms, val = synthetic.upload(ctx, m)
    reach = synthetic.reachable_mask(val, m)
    starts = synthetic.sample_valid_poses(val, m, B, seed=1000, reachable=reach)
    goals = synthetic.sample_valid_poses(val, m, B, seed=2000, reachable=reach)
    seeds = np.arange(B, dtype=np.uint64)
    batch = pa.HybridAStarBatch(val, max_batch=B, max_nodes=81920, search_rows=1024)
    batch.initialize()
    want = batch.search_batch(starts, goals, seeds)
    fields = ("status", "n_expanded", "n_nodes", "n_path", "n_rng_draws", "n_rs_attempts", "n_state_checks", "n_path_checks")
    want_a = np.array([[getattr(r, f) for f in fields] for r in want], dtype=np.int64)
    want_cost = np.array([r.cost for r in want])
    table = batch.nonholo_table()
    batch.close()
    pipe = pa.HybridAStarPipeline(val, capacity=3 * B, max_nodes=81920)
    pipe.initialize(table)
    first, submitted, done, bad = None, 0, 0, []
    t0 = time.time()
    while done < rounds * B:
        if submitted < rounds * B and pipe.free_slots() >= B:
            tickets = pipe.submit(starts, goals, seeds)
            assert len(tickets) == B
            first = int(tickets[0]) if first is None else first
            submitted += B
        tickets, res = pipe.poll_array(8192)
        if len(tickets):
            q = (tickets.astype(np.int64) - first) % B
            got = np.column_stack([res[f].astype(np.int64) for f in fields])
            wrong = (got != want_a[q]).any(axis=1) | ((res["status"] == 0) & (res["cost"] != want_cost[q]))
            bad.extend(int(x) for x in q[wrong])
            done += len(tickets)
        else:
            time.sleep(0.0005)
        assert time.time() - t0 < 300, "pipeline stalled: %d of %d" % (done, rounds * B)
    assert not bad, "%d results differ from the batch planner's, queries %s" % (len(bad), sorted(set(bad))[:20])
    pipe.close()
