"""N > 1 path: two ranks shard a query batch block-cyclically, plan their shards, gather the fixed-size records (status, cost,
counters AND the path poses) and must reproduce the single-process result query for query; the map set travels by
broadcast from rank 0.  On CPU the ranks plan with the oracle (this part is about sharding, broadcast and gather, gloo); with a
GPU present the same test body runs the real planner in both ranks on device 0 (-m gpu), the collectives still over gloo since
RCCL refuses two ranks on one device."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
MAX_POSES = 96


class _Rec:
    def __init__(self, status, cost, n_expanded, n_path):
        self.status, self.cost, self.n_expanded, self.n_path = status, cost, n_expanded, n_path


class OraclePlanner:
    """the CPU oracle behind the two calls the sharded driver needs: search_batch + get_path_of"""

    def __init__(self, m):
        import oracle_lib as O
        half = float(m["upper"][0])
        self.w = O.World(half, half, m["resolution"])
        self.w.set_occ(m["occ"])
        self.w.set_d2(m["d2"])
        self.w.set_pathcost(m["path_cost"])
        self.h = O.Hybrid(self.w)
        self.paths = []

    def search_batch(self, starts, goals, seeds):
        out, self.paths = [], []
        for s, g, sd in zip(starts, goals, seeds):
            r = self.h.search(s, g, int(sd))
            out.append(_Rec(r["status"], r["cost"], len(r["expanded"]), len(r["path_poses"])))
            self.paths.append(r["path_poses"])
        return out

    def get_path_of(self, q):
        return dict(poses=self.paths[q])


class GpuPlanner:
    def __init__(self, m):
        import pathplanning_amd as pa
        from pathplanning_amd import synthetic
        self.ctx = pa.Context(0)
        self.ms, self.val = synthetic.upload(self.ctx, m)
        self.pl = pa.HybridAStarBatch(self.val, pa.HybridAStarSearchParameters(), max_batch=16, max_nodes=32768)
        self.pl.initialize()

    def search_batch(self, starts, goals, seeds):
        return self.pl.search_batch(starts, goals, np.asarray(seeds, dtype=np.uint64))

    def get_path_of(self, q):
        return self.pl.get_path_of(q)


def _map():
    import oracle_lib as O
    w = O.synthetic_world(128, 3, 5)
    return dict(lower=w.lb.copy(), upper=w.ub.copy(), resolution=0.1, occ=w.occ(), d2=w.d2(), path_cost=w.pathcost())


def _queries(n):
    rng = np.random.RandomState(0)
    starts = np.column_stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(-3, 3, n)])
    goals = np.column_stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(-3, 3, n)])
    return starts, goals, np.arange(n, dtype=np.uint64) + 50


def _plan(planner_cls, m, idx, starts, goals, seeds):
    from pathplanning_amd import sharding
    planner = planner_cls(m)
    res = planner.search_batch(starts[idx], goals[idx], seeds[idx])
    return sharding.records_from_results(res, len(idx), planner, MAX_POSES)


def _worker(rank, world, port, n, backend, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pathplanning_amd import sharding
    # only rank 0 has the map; the others receive it
    m = sharding.broadcast_map_set(_map() if rank == 0 else None, 0, rank, world)
    starts, goals, seeds = _queries(n)
    idx = sharding.shard_indices(n, rank, world)
    rec = _plan(GpuPlanner if backend == "gpu" else OraclePlanner, m, idx, starts, goals, seeds)
    dist.barrier()
    full = sharding.gather_records(rec, n, rank, world)
    if rank == 0:
        ret.put((full, {k: np.asarray(v) for k, v in m.items()}))
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(backend):
    from pathplanning_amd import sharding
    n = 7  # ragged: 4 + 3
    assert list(sharding.shard_indices(n, 0, 2)) == [0, 2, 4, 6]
    assert list(sharding.shard_indices(n, 1, 2)) == [1, 3, 5]
    starts, goals, seeds = _queries(n)
    m0 = _map()
    single = _plan(OraclePlanner, m0, np.arange(n), starts, goals, seeds)  # the reference result: one process, CPU oracle
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, backend, ret)) for r in range(2)]
    for p in procs:
        p.start()
    full, m_recv = ret.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for k in ("occ", "d2", "path_cost", "lower", "upper"):
        assert np.array_equal(m_recv[k], np.asarray(m0[k])), k
    assert full.shape == single.shape == (n, sharding.record_width(MAX_POSES))
    assert np.array_equal(full[:, [0, 2, 3]], single[:, [0, 2, 3]])  # status, expansions, path length: exact
    ok = single[:, 0] == 0
    assert ok.sum() >= 3
    tol = 0.0 if backend == "oracle" else 1e-5
    assert np.abs(full[ok][:, 1] - single[ok][:, 1]).max() <= tol
    assert np.abs(full[ok][:, 4:] - single[ok][:, 4:]).max() <= tol
    st, cost, nexp, poses = sharding.poses_of_record(full[np.argmax(ok)])
    assert st == 0 and len(poses) >= 2 and np.abs(poses[0] - starts[np.argmax(ok)]).max() < 1e-9


def test_two_rank_gloo_sharding_matches_single_process():
    _run_two_ranks("oracle")


@pytest.mark.gpu
def test_two_ranks_plan_on_the_gpu_and_match_the_oracle():
    _run_two_ranks("gpu")


def _rccl_one_rank(port, ret):
    """child process: RCCL (backend "nccl") with world_size 1 on cuda:0, device tensors through both collectives of the sharding path"""
    try:
        import torch
        import torch.distributed as dist
        from pathplanning_amd import sharding, synthetic
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        dev = torch.device("cuda", 0)
        m = synthetic.make_map(256, 6, seed=3)
        got = sharding.broadcast_map_set(m, 0, 0, 1, device=dev, force_collective=True)  # header + three grids (int32, int32, float32) as device tensors
        ok_map = all(np.array_equal(got[k], np.asarray(m[k])) for k in ("occ", "d2", "path_cost", "lower", "upper")) and got["resolution"] == float(m["resolution"])
        rec = np.random.RandomState(1).uniform(-5, 5, (37, sharding.record_width(8)))
        out = sharding.gather_records(rec, 37, 0, 1, device=dev, force_collective=True)  # all_gather of float64 device tensors
        t = torch.ones(4, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py's elapsed-time reduction
        dist.barrier()
        dist.destroy_process_group()
        ret.put((ok_map, bool(np.array_equal(out, rec)), float(t.sum().item())))
    except Exception as e:  # the parent reports it
        ret.put(("error", repr(e), 0.0))


@pytest.mark.gpu
def test_rccl_backend_carries_the_sharding_collectives_on_device_tensors():
    """Every multi-rank run so far used gloo.  RCCL refuses two ranks on one device, so on a one-GPU box it is exercised with
    world_size 1: the `nccl` process group is initialised and the map-set broadcast (header + int32 / float32 grids), the record
    all_gather (float64) and bench.py's all_reduce run as real collectives on device tensors (force_collective skips the
    single-rank short cuts)."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank, args=(29500 + (os.getpid() % 2000) + 3, ret))
    p.start()
    got = ret.get(timeout=600)
    p.join(timeout=120)
    assert got[0] is True and got[1] is True and got[2] == 4.0, got
    assert p.exitcode == 0
