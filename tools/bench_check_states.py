#!/usr/bin/env python3
"""k_check_states alone: 2^26 streamed poses (device-resident), HIP-event time on the library's stream, algorithmic 29 B/pose
 and actually moved 25 B/pose (24 B pose + 1 B flag; the validity bitmap is cache-resident) against the 8 TB/s roof.
PP_CS_STAGED=1 python tools/bench_check_states.py  -> the LDS-staged kernel of round 1 for comparison."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathplanning_amd as pa  # noqa: E402
from pathplanning_amd import synthetic  # noqa: E402
from pathplanning_amd._lib import check  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
dev = torch.device("cuda", 0)
ctx = pa.Context(0)
m = synthetic.make_map(1024, 24, seed=1)
ms, val = synthetic.upload(ctx, m)
half = float(m["upper"][0])
g = torch.Generator(device=dev)
g.manual_seed(42)
poses = torch.empty(n, 3, dtype=torch.float64, device=dev)
poses[:, 0].uniform_(-half, half, generator=g)
poses[:, 1].uniform_(-half, half, generator=g)
poses[:, 2].uniform_(-3.141592653589793, 3.141592653589793, generator=g)
out = torch.empty(n, dtype=torch.uint8, device=dev)
torch.cuda.synchronize(dev)
for _ in range(3):
    check(ctx.lib.pp_check_states_dev(ms.h, n, C.c_void_p(poses.data_ptr()), C.c_void_p(out.data_ptr())))
ctx.synchronize()
best = 1e9
for rep in range(5):
    ctx.timer_start()
    for _ in range(10):
        check(ctx.lib.pp_check_states_dev(ms.h, n, C.c_void_p(poses.data_ptr()), C.c_void_p(out.data_ptr())))
    best = min(best, ctx.timer_stop() / 10)
valid = int(out.sum().item())
print(json.dumps(dict(kernel="k_check_states (round 1, LDS-staged tiles, bitmap gathered through L1)" if os.environ.get("PP_CS_STAGED") == "1" else ("k_check_states_pipe (pipelined tiles, bitmap gathered through L1)" if os.environ.get("PP_CS_LDS") == "0" else "k_check_states_lds (bitmap resident in LDS)"), poses=n, ms=best, checks_per_s=n / (best * 1e-3),
                      algorithmic_GBs=n * 29 / (best * 1e-3) / 1e9, moved_GBs=n * 25 / (best * 1e-3) / 1e9, frac_of_8TBs_algorithmic=n * 29 / (best * 1e-3) / 8e12,
                      frac_of_8TBs_moved=n * 25 / (best * 1e-3) / 8e12, valid=valid)))

# calibration on the same buffers: what a read-only stream and a copy of this size reach on this GPU (torch kernels, torch events)
def _time(fn, reps=10):
    fn()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best_ = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        best_ = min(best_, e0.elapsed_time(e1) / reps)
    return best_


flat = poses.view(-1)
dst = torch.empty_like(flat)
t_sum = _time(lambda: flat.sum())
t_copy = _time(lambda: dst.copy_(flat))
t_max = _time(lambda: torch.amax(flat))
print(json.dumps(dict(calibration="torch kernels on the same 1.6 GB pose buffer", read_only_sum_GBs=flat.numel() * 8 / (t_sum * 1e-3) / 1e9,
                      read_only_amax_GBs=flat.numel() * 8 / (t_max * 1e-3) / 1e9, copy_read_plus_write_GBs=2 * flat.numel() * 8 / (t_copy * 1e-3) / 1e9)))
