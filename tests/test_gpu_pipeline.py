"""-m gpu: the streaming pipeline (include/pp_hip.h: pp_pipeline_*; pathplanning_amd/csrc/pp_pipeline.hpp) against the CPU oracle and
against the batch planner: every query's status, expansion sequence, counters, cost and path are what pp_planner_search_batch
returns -- with field slots recycled several times (a stale cache line of a slot's previous field, start or goal would show here),
across submissions that arrive while the grid is busy, and after the grid has gone idle and been started again."""
import time

import numpy as np
import pytest

import oracle_lib as O
from gpu_common import make_pair, valid_random_poses

pytestmark = pytest.mark.gpu


def drain(pipe, want, release=True, timeout=120.0):
    """polls until `want` results have arrived; returns {ticket: QueryResult}"""
    got = {}
    t0 = time.time()
    while len(got) < want:
        tickets, res = pipe.poll(4096, release=release)
        for i, t in enumerate(tickets):
            got[int(t)] = res[i]
        if not len(tickets):
            time.sleep(0.001)
        assert time.time() - t0 < timeout, "pipeline stalled: %d of %d results" % (len(got), want)
    return got


def check_against_oracle(pipe, ticket, r, h, start, goal, seed):
    o = h.search(start, goal, int(seed))
    assert r.status == o["status"], (ticket, r.status, o["status"])
    assert r.n_expanded == len(o["expanded"]) and r.n_nodes == o["n_nodes"]
    assert np.array_equal(pipe.get_expanded_of(ticket), o["expanded"])
    assert r.n_rng_draws == o["n_rng_draws"] and r.n_rs_attempts == o["n_rs_attempts"]
    assert r.n_state_checks == o["n_state_checks"] and r.n_path_checks == o["n_path_checks"]
    if o["status"] == 0:
        assert abs(r.cost - o["cost"]) < 1e-5
        path = pipe.get_path_of(ticket)
        assert len(path["poses"]) == len(o["path_poses"]) and np.abs(path["poses"] - o["path_poses"]).max() < 1e-5
        assert np.array_equal(path["kind"], o["path_kind"])
    # the plan as it left the GPU with its completion record (pp_pipeline_get_paths: the ring in pinned host memory)
    poses, n_poses = pipe.get_paths([ticket], max_poses=512, release=False)
    assert n_poses[0] == (len(o["path_poses"]) if o["status"] == 0 else 0)
    if n_poses[0]:
        assert np.abs(poses[0, :n_poses[0]] - o["path_poses"]).max() < 1e-5
    return o["status"] == 0


def test_pipeline_matches_the_oracle_with_recycled_slots():
    import pathplanning_amd as pa
    w, ms, val, ctx = make_pair(256, 6, 3)
    rng = np.random.RandomState(11)
    n = 200
    starts, goals = valid_random_poses(rng, w, n), valid_random_poses(rng, w, n)
    seeds = np.arange(n, dtype=np.uint64) + 1000
    pipe = pa.HybridAStarPipeline(val, capacity=48, max_nodes=32768, search_rows=16, log_expansions=True)
    assert pipe.capacity == 48 and pipe.search_rows == 16
    pipe.initialize()
    h = O.Hybrid(w, O.params_array(), table=pipe.nonholo_table())
    index_of = {}
    nxt, done, solved = 0, 0, 0
    t0 = time.time()
    while done < n:
        if nxt < n and pipe.free_slots() > 0:
            k = min(n - nxt, 17)  # odd-sized submissions, most of them while the grid is busy
            tickets = pipe.submit(starts[nxt:nxt + k], goals[nxt:nxt + k], seeds[nxt:nxt + k])
            for i, t in enumerate(tickets):
                index_of[int(t)] = nxt + i
            nxt += len(tickets)
        tickets, res = pipe.poll(64, release=False)
        for i, t in enumerate(tickets):
            q = index_of[int(t)]
            solved += check_against_oracle(pipe, int(t), res[i], h, starts[q], goals[q], seeds[q])
            done += 1
        if len(tickets):
            pipe.release(tickets)
        assert time.time() - t0 < 300
    assert nxt == n and pipe.in_flight() == 0 and pipe.free_slots() == 48
    assert solved >= n // 2
    pipe.close()


@pytest.mark.parametrize("urgent_clearance", [None, "0", "1000"])
def test_pipeline_equals_the_batch_planner_and_restarts_after_idling(monkeypatch, urgent_clearance):
    """urgent_clearance: the order-of-work knob of pp_pipeline.hpp (queries with a pose near an obstacle have their fields built ahead of
    the submissions queued before them, through a ring every running wavefront launch serves): the default (2 m), off, and "every query is
    urgent" -- each slot then sits in the ring AND in its launch's list and must be built exactly once; results are the batch planner's whatever it is"""
    import pathplanning_amd as pa
    if urgent_clearance is not None:
        monkeypatch.setenv("PP_PIPE_URGENT_CLEARANCE", urgent_clearance)
    w, ms, val, ctx = make_pair(512, 12, 1)
    rng = np.random.RandomState(5)
    n = 1500
    starts, goals = valid_random_poses(rng, w, n), valid_random_poses(rng, w, n)
    seeds = np.arange(n, dtype=np.uint64) + 7
    batch = pa.HybridAStarBatch(val, max_batch=n, max_nodes=65536, search_rows=256)
    batch.initialize()
    want = batch.search_batch(starts, goals, seeds)
    table = batch.nonholo_table()
    pipe = pa.HybridAStarPipeline(val, capacity=400, max_nodes=65536, search_rows=128)
    pipe.initialize(table)
    fields = ("status", "n_expanded", "n_nodes", "n_path", "n_rng_draws", "n_rs_attempts", "n_state_checks", "n_path_checks", "n_lattice_boundary_hits")

    def run(lo, hi):
        index_of, nxt, got = {}, lo, {}
        t0 = time.time()
        while len(got) < hi - lo:
            if nxt < hi and pipe.free_slots() > 0:
                tickets = pipe.submit(starts[nxt:hi], goals[nxt:hi], seeds[nxt:hi])
                for i, t in enumerate(tickets):
                    index_of[int(t)] = nxt + i
                nxt += len(tickets)
            tickets, res = pipe.poll(512)
            for i, t in enumerate(tickets):
                got[index_of[int(t)]] = res[i]
            if not len(tickets):
                time.sleep(0.0005)
            assert time.time() - t0 < 300
        for q, r in got.items():
            for f in fields:
                assert getattr(r, f) == getattr(want[q], f), (q, f, getattr(r, f), getattr(want[q], f))
            assert r.cost == want[q].cost or r.status != 0

    run(0, 1000)
    assert pipe.in_flight() == 0
    time.sleep(0.3)  # every wave of the grid has left by now (nothing submitted is unclaimed): the next submission starts it again
    run(1000, n)
    pipe.close()
    batch.close()


@pytest.mark.parametrize("solo", [("50", "1000000"), ("2000", "8")])
def test_waves_with_a_long_query_stop_claiming_and_results_stay_the_batch_planners(monkeypatch, solo):
    """PP_PIPE_SOLO_AFTER / PP_PIPE_SOLO_BACKLOG (off by default): a wave one of whose rows has passed that many expansions takes nothing new while
    the ready ring holds fewer fields than the backlog.  ("50", huge): nearly every wave goes solo at once and never claims beside a long query --
    everything must still complete (a wave whose long query ends claims again) and equal the batch planner's results; ("2000", "8"): the policy as
    it would be used, switching on and off with the ring's length."""
    import pathplanning_amd as pa
    monkeypatch.setenv("PP_PIPE_SOLO_AFTER", solo[0])
    monkeypatch.setenv("PP_PIPE_SOLO_BACKLOG", solo[1])
    w, ms, val, ctx = make_pair(512, 12, 1)
    rng = np.random.RandomState(9)
    n = 600
    starts, goals = valid_random_poses(rng, w, n), valid_random_poses(rng, w, n)
    seeds = np.arange(n, dtype=np.uint64) + 77
    batch = pa.HybridAStarBatch(val, max_batch=n, max_nodes=65536, search_rows=256)
    batch.initialize()
    want = batch.search_batch(starts, goals, seeds)
    pipe = pa.HybridAStarPipeline(val, capacity=256, max_nodes=65536, search_rows=64)
    pipe.initialize(batch.nonholo_table())
    fields = ("status", "n_expanded", "n_nodes", "n_path", "n_rng_draws", "n_rs_attempts", "n_state_checks", "n_path_checks", "n_lattice_boundary_hits")
    index_of, nxt, got = {}, 0, {}
    t0 = time.time()
    while len(got) < n:
        if nxt < n and pipe.free_slots() > 0:
            tickets = pipe.submit(starts[nxt:n], goals[nxt:n], seeds[nxt:n])
            for i, t in enumerate(tickets):
                index_of[int(t)] = nxt + i
            nxt += len(tickets)
        tickets, res = pipe.poll(512)
        for i, t in enumerate(tickets):
            got[index_of[int(t)]] = res[i]
        if not len(tickets):
            time.sleep(0.0005)
        assert time.time() - t0 < 300, "pipeline stalled: %d of %d results" % (len(got), n)
    for q, r in got.items():
        for f in fields:
            assert getattr(r, f) == getattr(want[q], f), (q, f, getattr(r, f), getattr(want[q], f))
        assert r.cost == want[q].cost or r.status != 0
    assert max(r.n_expanded for r in got.values()) > int(solo[0])  # (the policy had something to act on)
    pipe.close()
    batch.close()


def test_pipeline_under_dribbling_submissions_and_short_idle_timeout(monkeypatch):
    """The grid's waves leave when nothing is left to claim (or after PP_PIPE_IDLE_MS without work) and every submission launches the grid
    again: here with a 1 ms idle time-out, 8 slots, 2 waves and queries arriving one to three at a time with pauses, so that waves leave
    and re-enter constantly (the hand-back of a wave index races with the next submission's launch).  Every result must still be the
    batch planner's, and nothing may stall."""
    import pathplanning_amd as pa
    monkeypatch.setenv("PP_PIPE_IDLE_MS", "1")
    w, ms, val, ctx = make_pair(256, 6, 3)
    rng = np.random.RandomState(77)
    n = 240
    starts, goals = valid_random_poses(rng, w, n), valid_random_poses(rng, w, n)
    seeds = np.arange(n, dtype=np.uint64) + 5000
    batch = pa.HybridAStarBatch(val, max_batch=n, max_nodes=32768, search_rows=64)
    batch.initialize()
    want = batch.search_batch(starts, goals, seeds)
    pipe = pa.HybridAStarPipeline(val, capacity=8, max_nodes=32768, search_rows=8)
    pipe.initialize(batch.nonholo_table())
    index_of, nxt, got = {}, 0, {}
    t0 = time.time()
    while len(got) < n:
        if nxt < n and pipe.free_slots() > 0:
            k = min(n - nxt, int(rng.randint(1, 4)), pipe.free_slots())
            for i, t in enumerate(pipe.submit(starts[nxt:nxt + k], goals[nxt:nxt + k], seeds[nxt:nxt + k])):
                index_of[int(t)] = nxt + i
            nxt = len(index_of)
            if rng.rand() < 0.3:
                time.sleep(float(rng.uniform(0.0, 0.004)))  # longer than the idle time-out now and then
        tickets, res = pipe.poll(16)
        for i, t in enumerate(tickets):
            got[index_of[int(t)]] = (res[i].status, res[i].n_expanded, res[i].n_nodes, res[i].cost)
        assert time.time() - t0 < 240, "pipeline stalled: %d of %d" % (len(got), n)
    for q in range(n):
        s_, e_, nn_, c_ = got[q]
        assert (s_, e_, nn_) == (want[q].status, want[q].n_expanded, want[q].n_nodes) and (s_ != 0 or c_ == want[q].cost), q
    assert pipe.in_flight() == 0
    pipe.close()
    batch.close()


def test_pipeline_replay_with_queued_launches_and_urgent_slots_recycled_early(monkeypatch):
    """Order of work must not touch results.  With the urgent ring an urgent slot is built by whatever wavefront launch is running, searched,
    polled and refilled before its own launch -- queued behind several others -- reaches its list entry: that stale entry must not claim
    the refilled slot (entries and claim words carry the slot's generation), and a launch reads goal poses written after it began
    (agent-scope loads).  A first version without these two guards built about one field in 4096 for a slot's previous goal at
    bench.py's scale (bench.py checks every run: `replay_consistent`; tests/test_gpu_fullsize.py repeats its workload).  Here: six
    submissions stay queued, a third of the queries are urgent, and 40 rounds replay the same 384 queries."""
    import pathplanning_amd as pa
    monkeypatch.setenv("PP_PIPE_URGENT_CLEARANCE", "3.0")
    w, ms, val, ctx = make_pair(512, 12, 1)
    rng = np.random.RandomState(9)
    n, rounds = 384, 40
    starts, goals = valid_random_poses(rng, w, n), valid_random_poses(rng, w, n)
    seeds = np.arange(n, dtype=np.uint64) + 31
    batch = pa.HybridAStarBatch(val, max_batch=n, max_nodes=65536, search_rows=256)
    batch.initialize()
    want = batch.search_batch(starts, goals, seeds)
    want_a = np.array([(r.status, r.n_expanded, r.n_nodes, r.n_rng_draws, r.n_state_checks, r.n_path_checks) for r in want], dtype=np.int64)
    pipe = pa.HybridAStarPipeline(val, capacity=6 * n, max_nodes=65536, search_rows=512)
    pipe.initialize(batch.nonholo_table())
    first, submitted, done, bad = None, 0, 0, []
    t0 = time.time()
    while done < rounds * n:
        if submitted < rounds * n and pipe.free_slots() >= n:
            tickets = pipe.submit(starts, goals, seeds)
            assert len(tickets) == n
            first = int(tickets[0]) if first is None else first
            submitted += n
        tickets, res = pipe.poll_array(4096)
        if len(tickets):
            q = (tickets.astype(np.int64) - first) % n
            got = np.column_stack([res[f].astype(np.int64) for f in ("status", "n_expanded", "n_nodes", "n_rng_draws", "n_state_checks", "n_path_checks")])
            bad.extend(int(x) for x in q[(got != want_a[q]).any(axis=1)])
            done += len(tickets)
        else:
            time.sleep(0.0005)
        assert time.time() - t0 < 300, "pipeline stalled: %d of %d" % (done, rounds * n)
    assert not bad, "queries whose result differs from the batch planner's: %s" % sorted(set(bad))[:20]
    assert pipe.in_flight() == 0
    pipe.close()
    batch.close()


def test_paths_longer_than_the_host_ring_and_cut_requests(monkeypatch):
    """PP_PIPE_PATH_POSES=6: every plan of more than six nodes has its start end fetched from the device records; a request for fewer
    poses than the path has returns its first poses and the full count."""
    import pathplanning_amd as pa
    monkeypatch.setenv("PP_PIPE_PATH_POSES", "6")
    w, ms, val, ctx = make_pair(256, 6, 3)
    rng = np.random.RandomState(21)
    n = 24
    starts, goals = valid_random_poses(rng, w, n), valid_random_poses(rng, w, n)
    seeds = np.arange(n, dtype=np.uint64) + 5
    pipe = pa.HybridAStarPipeline(val, capacity=32, max_nodes=32768, search_rows=16)
    pipe.initialize()
    h = O.Hybrid(w, O.params_array(), table=pipe.nonholo_table())
    tickets = pipe.submit(starts, goals, seeds)
    assert len(tickets) == n
    got = drain(pipe, n, release=False)
    long_paths = 0
    for i, t in enumerate(tickets):
        o = h.search(starts[i], goals[i], int(seeds[i]))
        poses, n_poses = pipe.get_paths([t], max_poses=512, release=False)
        want = len(o["path_poses"]) if o["status"] == 0 else 0
        assert n_poses[0] == want and got[int(t)].status == o["status"]
        if want:
            assert np.abs(poses[0, :want] - o["path_poses"]).max() < 1e-5
            long_paths += want > 6
            cut, n_cut = pipe.get_paths([t], max_poses=4, release=False)
            assert n_cut[0] == want and np.abs(cut[0, :min(4, want)] - o["path_poses"][:4]).max() < 1e-5
    assert long_paths >= 5
    poses, n_poses = pipe.get_paths(tickets, max_poses=64, release=True)  # the batched form, slots returned
    assert pipe.free_slots() == 32 and pipe.in_flight() == 0
    pipe.close()


def test_held_slots_post_process_like_the_batch_planner_and_the_buffer_set_refuses_batches():
    """pp_pipeline_planner() is the handle of the pipeline's buffer set: post-processing its held slots (the rows write the slot's result
    record on the device too) gives what the batch planner gives for the same queries; batch entry points on it are refused."""
    import pathplanning_amd as pa
    PP_ERR_INVALID = -1  # include/pp_hip.h
    w, ms, val, ctx = make_pair(256, 6, 3)
    ms.upload_nearest_cells(*O.world_nearest(w))
    rng = np.random.RandomState(22)
    n = 12
    starts, goals = valid_random_poses(rng, w, n), valid_random_poses(rng, w, n)
    seeds = np.arange(n, dtype=np.uint64) + 77
    pipe = pa.HybridAStarPipeline(val, capacity=16, max_nodes=32768, search_rows=8)
    pipe.initialize()
    batch = pa.HybridAStarBatch(val, max_batch=n, max_nodes=32768)
    batch.initialize(pipe.nonholo_table())
    ref = batch.search_batch(starts, goals, seeds)
    batch.postprocess(n, path_interpolation=0.8)
    tickets = pipe.submit(starts, goals, seeds)
    got = drain(pipe, n, release=False)
    pipe.postprocess_held(16, path_interpolation=0.8)
    for i, t in enumerate(tickets):
        assert got[int(t)].status == ref[i].status and got[int(t)].n_expanded == ref[i].n_expanded
        a, b = pipe.get_processed_path_of(t), batch.get_processed_path(i)
        assert a["status"] == b["status"] and len(a["sampled"]) == len(b["sampled"])
        if len(a["sampled"]):
            assert np.array_equal(a["sampled"], b["sampled"]) and np.array_equal(a["cusp"], b["cusp"]) and np.array_equal(a["path"], b["path"])
    # the buffer set is the pipeline's: no batches, no result fetch, no expansion log without log_expansions
    lib = pipe.lib
    import ctypes as C
    from pathplanning_amd._lib import QueryResult, ptr
    res = (QueryResult * n)()
    sd = np.ascontiguousarray(seeds)
    assert lib.pp_planner_search_batch(pipe.planner_h, n, ptr(starts), ptr(goals), ptr(sd), C.cast(res, C.c_void_p)) == PP_ERR_INVALID
    assert lib.pp_planner_fetch_results(pipe.planner_h, n, C.cast(res, C.c_void_p)) == PP_ERR_INVALID
    cells = np.zeros((4, 3), dtype=np.int32)
    assert lib.pp_planner_get_expanded(pipe.planner_h, 0, ptr(cells)) == PP_ERR_INVALID
    pipe.release(tickets)
    batch.close()
    pipe.close()


def test_a_changed_validator_meets_no_query_in_flight():
    """The map as the kernels see it (bounds, grids, the validator's tunables) is a launch argument, and the persistent grid's waves keep the one they were
    launched with: a submission after `min_safe_radius` changed is refused while queries are in flight (some would be searched under the old radius, some
    under the new one) and accepted once they are polled -- the old grid's waves are waited for first -- with the results of a batch planner made after
    the change."""
    import pathplanning_amd as pa
    w, ms, val, ctx = make_pair(256, 6, 3)
    rng = np.random.RandomState(21)
    n = 96
    starts, goals = valid_random_poses(rng, w, n), valid_random_poses(rng, w, n)
    seeds = np.arange(n, dtype=np.uint64) + 5
    pipe = pa.HybridAStarPipeline(val, capacity=128, max_nodes=32768, search_rows=16)
    pipe.initialize()
    table = pipe.nonholo_table()
    t1 = pipe.submit(starts[:48], goals[:48], seeds[:48])
    assert len(t1) == 48
    old_radius = val.min_safe_radius
    val.min_safe_radius = old_radius * 0.8
    with pytest.raises(pa.PPError, match="in flight"):
        pipe.submit(starts[48:], goals[48:], seeds[48:])
    assert pipe.in_flight() == 48  # (nothing was taken by the refused call)
    val.min_safe_radius = old_radius
    drain(pipe, 48)
    val.min_safe_radius = old_radius * 0.8
    t2 = pipe.submit(starts[48:], goals[48:], seeds[48:])
    assert len(t2) == 48
    got = drain(pipe, 48)
    batch = pa.HybridAStarBatch(val, max_batch=48, max_nodes=32768, search_rows=16)
    batch.initialize(table)
    want = batch.search_batch(starts[48:], goals[48:], seeds[48:])
    for i, t in enumerate(t2):
        r = got[int(t)]
        assert (r.status, r.n_expanded, r.n_nodes, r.n_state_checks) == (want[i].status, want[i].n_expanded, want[i].n_nodes, want[i].n_state_checks), i
    val.min_safe_radius = old_radius
    pipe.close()
    batch.close()
