"""Query sharding for multi-GPU runs: one process per GPU (torch.distributed; backend "nccl" = RCCL
on ROCm, "gloo" in CPU tests).  Queries are independent given the read-only map set, so
there is NO data-path collective: each rank plans its own queries; the only communication is the
gather of fixed-size result records at the end of a batch."""
import numpy as np

RECORD_FIELDS = ("status", "cost", "n_expanded", "n_path")


def shard_indices(n_queries, rank, world):
    """Block-cyclic assignment: rank r owns queries {i : i mod world == r} (keeps per-query seeds stable and
    mixes easy/hard queries across ranks)."""
    return np.arange(rank, n_queries, world, dtype=np.int64)


def records_from_results(results, n):
    rec = np.zeros((n, len(RECORD_FIELDS)), dtype=np.float64)
    for i in range(n):
        r = results[i]
        rec[i] = (r.status, r.cost, r.n_expanded, r.n_path)
    return rec


def gather_records(local_records, n_queries, rank, world, device=None):
    """all_gather of the per-rank record blocks (padded to equal length) -> [n_queries, fields] in query order."""
    import torch
    import torch.distributed as dist
    if world == 1:
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
