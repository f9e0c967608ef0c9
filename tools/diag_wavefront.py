#!/usr/bin/env python3
"""Phase breakdown of the exact-order wavefront kernel (stamped diagnostic build)."""
import os
import sys

import numpy as np
import torch

torch.cuda.init()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathplanning_amd as pa  # noqa: E402
from pathplanning_amd import synthetic  # noqa: E402
from pathplanning_amd._lib import check, ptr  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 768
CELLS = int(sys.argv[3]) if len(sys.argv) > 3 else 1024  # map side in cells
ctx = pa.Context(0)
m = synthetic.make_map(CELLS, 24 * (CELLS // 1024) ** 2 if CELLS >= 1024 else 12, seed=1)
ms, val = synthetic.upload(ctx, m)
goals = synthetic.sample_valid_poses(val, m, G, seed=2000)[:, :2].copy()
cnt = np.zeros((G, 20), dtype=np.uint64)
for it in range(2):
    ctx.timer_start()
    check(ctx.lib.pp_obstacle_heuristic_profile(ms.h, G, ptr(np.ascontiguousarray(goals)), ptr(cnt)))
    ms_ = ctx.timer_stop()
names = ["init", "min", "partition", "sort", "offer", "push", "tail"]
c = cnt.astype(np.float64)
tot = c[:, :7].sum()
print("goals %d  wall %.1f ms (incl. alloc/copies)" % (G, ms_))
print("rounds/goal %.0f  mean window %.0f  rounds with the open list in HBM %.1f %%" % (c[:, 7].mean(), c[:, 8].sum() / c[:, 7].sum(), 100 * c[:, 9].sum() / c[:, 7].sum()))
for i, nm in enumerate(names):
    print("  %-9s %5.1f %%  %.0f cycles/round" % (nm, 100 * c[:, i].sum() / tot, c[:, i].sum() / c[:, 7].sum()))
print("cycles/goal %.3g" % (c[:, :7].sum(1).mean()))
print("fallback rounds %.1f %%, their push phase %.1f %% of all cycles" % (100 * c[:, 10].sum() / c[:, 7].sum(), 100 * c[:, 11].sum() / tot))
print("offer: store-wait %.0f  loads %.0f  count+barrier %.0f  (whole offer to end of insert %.0f) cycles/round" % tuple(c[:, k].sum() / c[:, 7].sum() for k in (12, 13, 14, 15)))
print("push (cumulative): look-up %.0f  slot scan %.0f  stores %.0f cycles/round" % tuple(c[:, k].sum() / c[:, 7].sum() for k in (16, 17, 18)))
G2 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if G2:
    goals2 = synthetic.sample_valid_poses(val, m, G2, seed=2001)[:, :2].copy()
    out = torch.empty((G2, CELLS * CELLS), dtype=torch.float32, device="cuda")
    for it in range(3):
        ctx.timer_start()
        check(ctx.lib.pp_obstacle_heuristic_dev(ms.h, G2, ptr(np.ascontiguousarray(goals2)), out.data_ptr()))
        t = ctx.timer_stop()
    print("standalone: %d goals in %.1f ms" % (G2, t))
