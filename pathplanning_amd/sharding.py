"""Query sharding for multi-GPU runs: one process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" in CPU
tests).  Queries are independent given the read-only map set, so there is NO data-path collective: each rank plans
its own queries.  Communication happens twice: one broadcast of the map set when it is loaded on one rank only, and one gather
of fixed-size result records {status, cost, nExpanded, nPoses, poses[max_poses] x 3} at the end of a batch."""
import ctypes as C

import numpy as np

RECORD_FIELDS = ("status", "cost", "n_expanded", "n_path")
MAP_GRIDS = (("occ", np.int32), ("d2", np.int32), ("path_cost", np.float32))


def shard_indices(n_queries, rank, world):
    """Block-cyclic assignment: rank r owns queries {i : i mod world == r} (keeps per-query seeds stable and
    mixes easy/hard queries across ranks)."""
    return np.arange(rank, n_queries, world, dtype=np.int64)


def record_width(max_poses=0):
    return len(RECORD_FIELDS) + 3 * int(max_poses)


def records_from_results(results, n, planner=None, max_poses=0):
    """One float64 row per local query: status, cost, n_expanded, n_path, then (when max_poses > 0) the first max_poses path
    poses (x, y, theta), zero-padded.  `planner` needs get_path_of(q) -> {"poses": (k, 3)} (HybridAStarBatch has it)."""
    rec = np.zeros((n, record_width(max_poses)), dtype=np.float64)
    if isinstance(results, C.Array) and n > 0:  # the planner's own result array: one vectorised pass (4096 records per batch and lane)
        a = np.frombuffer(results, dtype=np.dtype(type(results[0])), count=n)
        rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3] = a["status"], a["cost"], a["n_expanded"], a["n_path"]
        if not (max_poses and planner is not None):
            return rec
    for i in range(n):
        r = results[i]
        rec[i, :4] = (r.status, r.cost, r.n_expanded, r.n_path)
        if max_poses and planner is not None and r.status == 0 and r.n_path > 0:
            poses = np.asarray(planner.get_path_of(i)["poses"], dtype=np.float64).reshape(-1, 3)[:max_poses]
            rec[i, 4:4 + 3 * len(poses)] = poses.reshape(-1)
    return rec


def poses_of_record(row):
    """(status, cost, n_expanded, poses[k, 3]) of one gathered row"""
    n = int(row[3])
    k = min(n, (len(row) - 4) // 3)
    return int(row[0]), float(row[1]), int(row[2]), row[4:4 + 3 * k].reshape(k, 3)


def gather_records(local_records, n_queries, rank, world, device=None, force_collective=False):
    """all_gather of the per-rank record blocks (padded to equal length) -> [n_queries, fields] in query order.
    force_collective: run the collective even with one rank (exercises the backend's device-tensor path on a one-GPU box)."""
    import torch
    import torch.distributed as dist
    if world == 1 and not force_collective:
        return local_records
    per = (n_queries + world - 1) // world
    buf = torch.zeros(per, local_records.shape[1], dtype=torch.float64)
    buf[: len(local_records)] = torch.from_numpy(np.ascontiguousarray(local_records))
    if device is not None:
        buf = buf.to(device)
    parts = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = np.zeros((n_queries, local_records.shape[1]), dtype=np.float64)
    for r in range(world):
        idx = shard_indices(n_queries, r, world)
        out[idx] = parts[r].cpu().numpy()[: len(idx)]
    return out


def broadcast_map_set(m, src, rank, world, device=None, force_collective=False):
    """The map set of rank `src` on every rank: {lower, upper, resolution, occ, d2, path_cost} (the dict of synthetic.make_map, or
    grids loaded from elsewhere on one rank only).  A 16-double header carries the sizes, then one broadcast per grid
    (13 MB at 1024^2, 200 MB at 4096^2: xGMI-trivial, done once per map).  Ranks other than `src` pass m=None."""
    import torch
    import torch.distributed as dist
    if world == 1 and not force_collective:
        return m
    head = torch.zeros(16, dtype=torch.float64)
    if rank == src:
        rows, cols = m["occ"].shape
        head[:9] = torch.tensor([rows, cols, float(m["resolution"]), *[float(v) for v in m["lower"]], *[float(v) for v in m["upper"]]], dtype=torch.float64)
    if device is not None:
        head = head.to(device)
    dist.broadcast(head, src)
    h = head.cpu().numpy()
    rows, cols = int(h[0]), int(h[1])
    out = dict(resolution=float(h[2]), lower=h[3:6].copy(), upper=h[6:9].copy())
    for name, dtype in MAP_GRIDS:
        t = torch.from_numpy(np.ascontiguousarray(m[name], dtype=dtype)) if rank == src else torch.empty((rows, cols), dtype=torch.from_numpy(np.empty(0, dtype)).dtype)
        if device is not None:
            t = t.to(device)
        dist.broadcast(t, src)
        out[name] = t.cpu().numpy()
    return out
