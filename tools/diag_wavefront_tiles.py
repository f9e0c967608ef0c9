"""Stand-alone rate and work counters of the tile form of the obstacle-heuristic wavefront (pp_wavefront_tiles.hip).

    python tools/diag_wavefront_tiles.py [n_goals] [cells] [obstacles] [reps]

Builds the benchmark's map (K rectangle outlines), draws goals uniformly, runs the wavefront for all of them with
the chip to itself and prints milliseconds per launch, the 9 B/cell algorithmic rate against 8 TB/s, and the kernel's own
counters: tile visits per tile, bucket rounds per visit, candidate passes per round, cells per pass, wave time per goal.
"""
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import pathplanning_amd as pa
from pathplanning_amd import synthetic
from pathplanning_amd.planner import ObstaclesHeuristic


def main():
    n_goals = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    cells = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    obstacles = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    ctx = pa.Context(0)
    m = synthetic.make_map(cells, obstacles, 12345)
    ms, val = synthetic.upload(ctx, m)
    rng = np.random.RandomState(3)
    half = cells * 0.1 / 2
    goals = rng.uniform(-half, half, (n_goals, 2))
    out = torch.empty((n_goals, ms.rows * ms.cols), dtype=torch.float32, device="cuda:0")
    oh = ObstaclesHeuristic(ms)
    if os.environ.get("PP_WF_TILES") == "0":  # the ordered kernel alone, for comparison (no counters)
        t = []
        for _ in range(reps):
            ctx.synchronize()
            ctx.timer_start()
            oh.update_dev(goals, out)
            t.append(float(ctx.timer_stop()))
        print(json.dumps(dict(n_goals=n_goals, cells=cells, ordered_kernel_ms=t)))
        return
    runs = []
    for _ in range(reps):
        st, msl = oh.update_dev_tile_stats(goals, out)
        runs.append((st, msl))
    st, msl = min(runs, key=lambda r: r[1])
    tiles = st["tiles_per_goal"] * st["goals"]
    gbs = n_goals * cells * cells * 9 / (msl * 1e-3) / 1e9
    rec = dict(n_goals=n_goals, cells=cells, obstacles=obstacles, ms_per_launch=msl, all_runs_ms=[r[1] for r in runs], algorithmic_GBps=gbs, frac_of_8TBps=gbs / 8000.0,
               visits_per_tile=st["tile_visits"] / max(1, tiles), rounds_per_visit=st["rounds"] / max(1, st["tile_visits"]),
               passes_per_round=st["candidate_passes"] / max(1, st["rounds"]), cells_per_pass=st["cells"] / max(1, st["candidate_passes"]),
               handed_over=st["handed_over"], handed_over_goals=[(i, [float(x) for x in goals[i]]) for i in st["handed_over_goals"]],
               wave_cycles_per_goal=st["wave_cycles"] / max(1, st["goals"]),
               phase_share={k[7:]: st[k] / max(1, st["wave_cycles"]) for k in st if k.startswith("cycles_")}, counters=st)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
