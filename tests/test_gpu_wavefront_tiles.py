"""-m gpu: the tile form of the obstacle-heuristic wavefront (pp_wavefront_tiles.hip) against the oracle's sequential
ObstaclesHeuristic::Update (algo/heuristics.cpp:106-153) -- bit for bit -- where the other wavefront tests do not reach: grids that
are not multiples of the 64-cell tile, occupied / outside / corner goals, many goals per launch (waves taking several goals), the
hand-over to the ordered kernel, and the evidence that the tile form (not the ordered kernel) built the fields."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def world_with(occ, hx, hy):
    w = O.World(hx, hy, 0.1)
    assert (w.rows, w.cols) == occ.shape
    w.set_occ(occ)
    w.set_d2(np.full(occ.shape, 100, dtype=np.int32))
    return w


def device_map(w, occ):
    import pathplanning_amd as pa
    ctx = pa.Context(0)
    ms = pa.OccupancyMapSet.from_bounds(ctx, w.lb, w.ub, 0.1)
    assert (ms.rows, ms.cols) == occ.shape
    ms.upload_occupancy(occ)
    return ctx, ms


def random_occ(rng, rows, cols, density):
    occ = np.full((rows, cols), -1, dtype=np.int32)
    occ[rng.rand(rows, cols) < density] = 0
    return occ


@pytest.mark.parametrize("rows,cols,density", [(200, 333, 0.05), (65, 64, 0.0), (63, 130, 0.2), (129, 127, 0.33), (320, 64, 0.1)])
def test_ragged_grids_and_special_goals(rows, cols, density):
    import pathplanning_amd as pa
    rng = np.random.RandomState(rows * 1000 + cols)
    occ = random_occ(rng, rows, cols, density)
    w = world_with(occ, rows * 0.05, cols * 0.05)
    ctx, ms = device_map(w, occ)
    hx, hy = rows * 0.05, cols * 0.05
    goals = [(-hx + 0.01, -hy + 0.01), (hx - 0.01, hy - 0.01), (hx - 0.01, -hy + 0.01), (0.0, 0.0), (hx + 5.0, 0.0)]  # corners, centre, outside the map
    occ_cells = np.argwhere(occ >= 0)
    if len(occ_cells):  # a goal ON an occupied cell: the reference pushes it all the same (heuristics.cpp:119-121)
        r, c = occ_cells[len(occ_cells) // 2]
        goals.append((-hx + (r + 0.5) * 0.1, -hy + (c + 0.5) * 0.1))
    goals += [tuple(x) for x in rng.uniform([-hx, -hy], [hx, hy], (6, 2))]
    got = pa.ObstaclesHeuristic(ms).update(goals)
    for i, g in enumerate(goals):
        cost, explored = w.obstacle_heuristic(g)
        assert np.array_equal(got[i].view(np.uint32), cost.view(np.uint32)), (i, g)
        assert np.array_equal(np.isfinite(got[i]), explored.astype(bool)), (i, g)


def test_many_goals_per_wave_and_counters():
    """600 goals on a 256 x 256 map: every goal's field equals the oracle's; the counters show the tile form did the work and
    handed nothing over."""
    import torch
    import pathplanning_amd as pa
    from gpu_common import make_pair
    w, ms, val, ctx = make_pair(256, 6, 3)
    rng = np.random.RandomState(4)
    goals = rng.uniform(-12.8, 12.8, (600, 2))
    out = torch.empty((len(goals), ms.rows * ms.cols), dtype=torch.float32, device="cuda:0")
    st, ms_launch = pa.ObstaclesHeuristic(ms).update_dev_tile_stats(goals, out)
    assert st["goals"] == len(goals) and st["handed_over"] == 0
    assert st["tile_visits"] >= 16 * len(goals) * 0.9 and st["cells"] > 0.8 * 256 * 256 * len(goals)
    got = out.cpu().numpy().reshape(len(goals), ms.rows, ms.cols)
    for i in range(0, len(goals), 7):
        cost, _ = w.obstacle_heuristic(goals[i])
        assert np.array_equal(got[i].view(np.uint32), cost.view(np.uint32)), i


_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import torch, pathplanning_amd as pa
from gpu_common import make_pair
w, ms, val, ctx = make_pair(256, 6, 3)
rng = np.random.RandomState(5)
goals = rng.uniform(-12.8, 12.8, (40, 2))
mode = sys.argv[2]
if mode == "stats":
    out = torch.empty((len(goals), ms.rows * ms.cols), dtype=torch.float32, device="cuda:0")
    st, _ = pa.ObstaclesHeuristic(ms).update_dev_tile_stats(goals, out)
    got = out.cpu().numpy().reshape(len(goals), ms.rows, ms.cols)
    print("HANDED", st["handed_over"], st["goals"])
else:
    got = pa.ObstaclesHeuristic(ms).update(goals)
bad = 0
for i, g in enumerate(goals):
    cost, _ = w.obstacle_heuristic(g)
    bad += not np.array_equal(got[i].view(np.uint32), cost.view(np.uint32))
print("BAD", bad)
"""


def _child(env_extra, mode):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", _CHILD, ROOT, mode], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_hand_over_to_the_ordered_kernel():
    """PP_WF_TILES_FORCE_FALLBACK=3: every third goal is handed to the ordered kernel (as a tie of the fixed-point equation would
    be); all 40 fields still equal the oracle's, and the counters show 13 goals handed over."""
    out = _child({"PP_WF_TILES_FORCE_FALLBACK": "3"}, "stats")
    assert "BAD 0" in out, out
    assert "HANDED 13 40" in out, out


def test_tile_form_switched_off():
    """PP_WF_TILES=0: every goal through the ordered kernel, same fields."""
    out = _child({"PP_WF_TILES": "0"}, "plain")
    assert "BAD 0" in out, out


def test_64_column_tiles():
    """PP_WF_TILE_WIDTH=64 (one 64-bit mask word per row; the default is 32): the other instantiation of the kernel -- its blocked-move test
    shifts a 64-bit mask where the 32-column one uses v_bfe_i32 -- gives the same fields, plain and through the counters' instantiation."""
    out = _child({"PP_WF_TILE_WIDTH": "64"}, "plain")
    assert "BAD 0" in out, out
    out = _child({"PP_WF_TILE_WIDTH": "64"}, "stats")
    assert "BAD 0" in out and "HANDED 0 40" in out, out


def test_tile_queue_in_global_memory_through_a_pipeline():
    """PP_WF_TILES_QUEUE=global: a pipeline brings per-wave regions for the tile queue (what it does by itself above 2048 x 2048 cells,
    where 16 KB of queue per wave would halve the tile waves per CU) and its launches run the instantiation that keeps the queue there.
    The pipeline's oracle test (200 queries through 48 recycled slots: every field, every search) runs in a child process with the
    variable set; PP_WF_TILES_QUEUE is read once per process."""
    env = dict(os.environ)
    env["PP_WF_TILES_QUEUE"] = "global"
    env["PP_WF_TILES_TRACE"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_pipeline.py"), "-q", "-x", "-s", "-k",
                        "matches_the_oracle_with_recycled_slots or replay_with_queued_launches", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-1000:]
    assert "tile queue in global memory" in r.stderr, r.stderr[-1000:]  # (the pipeline's launches; a batch planner beside it keeps the queue in LDS)
