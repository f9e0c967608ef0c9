"""N>1 path on CPU: two gloo ranks shard a query batch block-cyclically, plan their shards (CPU oracle
stands in for the GPU planner here -- this test is about the sharding and the result gather), gather the
fixed-size records, and must reproduce the single-process result query for query."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class _Rec:
    def __init__(self, status, cost, n_expanded, n_path):
        self.status, self.cost, self.n_expanded, self.n_path = status, cost, n_expanded, n_path


def _plan(indices, starts, goals, seeds):
    import oracle_lib as O
    w = O.synthetic_world(128, 3, 5)
    h = O.Hybrid(w)
    out = []
    for i in indices:
        r = h.search(starts[i], goals[i], int(seeds[i]))
        out.append(_Rec(r["status"], r["cost"], len(r["expanded"]), len(r["path_poses"])))
    return out


def _queries(n):
    rng = np.random.RandomState(0)
    starts = np.column_stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(-3, 3, n)])
    goals = np.column_stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(-3, 3, n)])
    return starts, goals, np.arange(n, dtype=np.uint64) + 50


def _worker(rank, world, port, n, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pathplanning_amd import sharding
    starts, goals, seeds = _queries(n)
    idx = sharding.shard_indices(n, rank, world)
    res = _plan(idx, starts, goals, seeds)
    rec = sharding.records_from_results(res, len(idx))
    dist.barrier()
    full = sharding.gather_records(rec, n, rank, world)
    if rank == 0:
        ret.put(full)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_matches_single_process():
    from pathplanning_amd import sharding
    n = 7  # ragged: 4 + 3
    assert list(sharding.shard_indices(n, 0, 2)) == [0, 2, 4, 6]
    assert list(sharding.shard_indices(n, 1, 2)) == [1, 3, 5]
    starts, goals, seeds = _queries(n)
    single = sharding.records_from_results(_plan(range(n), starts, goals, seeds), n)
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, ret)) for r in range(2)]
    for p in procs:
        p.start()
    full = ret.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert full.shape == single.shape
    assert np.array_equal(np.isfinite(full), np.isfinite(single))
    m = np.isfinite(single)
    assert np.array_equal(full[m], single[m])
    assert (single[:, 0] == 0).sum() >= 3
