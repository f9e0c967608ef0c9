#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X planner core.

Workload (BASELINE.json configs[3], the config the metric is quoted on): batches of independent
Hybrid-A* (start, goal) queries on ONE 1024x1024 occupancy map (resolution 0.1 m, 24 rectangle
outlines), 4096 queries per GPU per step.  A step = obstacle-heuristic wavefront for every
query's goal + the graph search of every query.  Inputs (map set, tables, starts/goals/seeds) are
resident in HBM before the timed region.  With N GPUs every rank runs its own 4096 queries (weak
scaling, no data-path collective); RCCL gathers the fixed-size result records at the end of each step.

Prints ONE JSON line (rank 0).  Secondary metric in the same line: collision checks/s
(IsStateValid over 2^26 streamed poses).  `roofline` is for the dominant kernel of the step;
`cpu_baseline` is the CPU oracle (a quirk-exact port of the reference) timed on this box's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The batches in flight run on separate HIP streams; the runtime maps streams onto 4 hardware queues by default, and two
# streams that share a queue serialise (a 1-2 s persistent search kernel then blocks the other batch's wavefront).
# Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
WAVEFRONT_BYTES_PER_CELL = 9.0     # 4 B occ read + 4 B cost write + 1 B explored
CHECK_BYTES_PER_POSE = 29.0        # 24 B pose + 4 B distance gather + 1 B result
CHILD_BYTES = 143.0                # per expansion child


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=8, help="untimed steps; at least one per batch lane is always run (a lane's first batch is cold), whatever is asked")
    ap.add_argument("--batch", type=int, default=4096, help="queries per GPU per step (weak scaling) / per step in total (strong scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank runs --batch queries per step (the default the driver's 1/2/4/8 curve uses); strong: BASELINE configs[3] "
                         "literally -- --batch queries per step in total, block-cyclic over the ranks (512 per GPU at 8 GPUs)")
    ap.add_argument("--cells", type=int, default=1024)
    ap.add_argument("--obstacles", type=int, default=24)
    ap.add_argument("--max-nodes", type=int, default=81920, help=">= number of (x, y, aliased heading) cells: no query can run out of nodes")
    ap.add_argument("--check-poses", type=int, default=1 << 26)
    ap.add_argument("--cpu-sample", type=int, default=192, help="queries timed on the CPU oracle (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--search-rows", type=int, default=8192, help="search rows shared by the batches in flight (8192 = every resident wave slot)")
    ap.add_argument("--debug-skip", type=int, default=0, choices=(0, 1, 2, 3),
                    help="timing experiments only (the line is then marked invalid): after the warm-up drop the wavefront (1) or the search (2) from every step; "
                         "with 1 the searches reuse the cost fields the warm-up left in each lane (same goals every step)")
    ap.add_argument("--stagger-ms", type=float, default=-1.0,
                    help="delay between the first launches of the lanes; < 0 (default): the wavefront time of one batch alone, measured in the warm-up; 0: all lanes start together")
    ap.add_argument("--chain", choices=["none", "first", "always"], default="none",
                    help="pp_planner_start_after_fields_of between consecutively launched lanes (event-based phasing): in the first round of launches / always / never")
    ap.add_argument("--streams", type=int, default=8, help="--mode lanes: independent batches kept in flight (one planner + HIP stream each)")
    ap.add_argument("--mode", choices=("pipeline", "lanes"), default="pipeline",
                    help="pipeline (default): the library's streaming pipeline (pp_pipeline_*): one persistent search grid fed by the wavefront kernel through a "
                         "device-side queue, field slots recycled; lanes: round 2's scheduling, --streams batch planners refilled by this script")
    ap.add_argument("--capacity", type=int, default=0, help="--mode pipeline: queries in flight (field slots); 0 = 6 steps' worth")
    ap.add_argument("--pipe-rows", type=int, default=4096, help="--mode pipeline: rows of the persistent search grid")
    ap.add_argument("--map-source", choices=("product", "synthetic"), default="product",
                    help="product: outlines -> pp_map_set_cells -> pp_map_update_gvd_ex(REFERENCE_ORDER) (the library's own map pipeline); synthetic: numpy / scipy generator of rounds 1-2")
    ap.add_argument("--sample-alive", action="store_true", help="diagnostics: sample the number of live search waves four times a second (a blocking copy each)")
    ap.add_argument("--submit-chunk", type=int, default=4096, help="--mode pipeline: queries per submission (= per wavefront launch)")
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal on a one-GPU box: PP_BENCH_BACKEND=gloo PP_BENCH_DEVICE=0 runs the N-rank code path with every rank on device 0
        dist.init_process_group(backend=os.environ.get("PP_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    n_gpus = max(world, 1)
    if args.gpus != n_gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d: launch N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 "
                         "--master-port P bench.py --gpus N ...` (one rank per GPU)" % (args.gpus, n_gpus))
    local_rank = int(os.environ.get("PP_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import pathplanning_amd as pa
    from pathplanning_amd import synthetic

    # the map set is made on rank 0 and broadcast (RCCL; 13 MB at 1024^2) -- what a deployment does with a map that exists on one
    # rank only; the broadcast is outside the timed region, like the upload that follows it
    from pathplanning_amd import sharding as _sh
    # the map is built THROUGH THE PRODUCT (outlines rasterised on the device, GVD::Update in reference order), outside the timed region;
    # --map-source synthetic: round 1-2's numpy / scipy generator (dense-sampled outlines, scipy's EDT)
    map_info = None
    if rank == 0 or world == 1:
        if args.map_source == "product":
            m, map_info = synthetic.make_map_product(pa.Context(local_rank), args.cells, args.obstacles, seed=1)
        else:
            m = synthetic.make_map(args.cells, args.obstacles, seed=1)
    else:
        m = None
    if world > 1:
        m = _sh.broadcast_map_set(m, 0, rank, world, device=dev)
    params = pa.HybridAStarSearchParameters()
    from pathplanning_amd import sharding
    B = args.batch if args.scaling == "weak" else len(sharding.shard_indices(args.batch, rank, max(world, 1)))
    # Batches are independent, and one batch alone cannot fill the GPU (its search kernel is bound by the longest
    # query): keep `streams` batches in flight, each on its own HIP stream with its own planner workspace.
    pipeline_mode = args.mode == "pipeline"
    n_streams = 1 if pipeline_mode else max(1, args.streams)
    lanes = []
    pipe = None
    if pipeline_mode:
        c = pa.Context(local_rank)
        ms_i, val_i = synthetic.upload(c, m)
        cap = args.capacity if args.capacity > 0 else max(6 * B, 3 * args.submit_chunk)  # (a submission must fit: steps smaller than a submission share one)
        if args.submit_chunk > cap // 3:
            # a submission waits for as many free slots as it has queries: one of a third of the capacity at most, or the pipeline runs empty
            # before every submission (config 5 with --capacity 1536 and the default chunk of 4096: 470 plans/s instead of 680)
            args.submit_chunk = max(B, cap // 3 // B * B)
        pipe = pa.HybridAStarPipeline(val_i, params, capacity=cap, max_nodes=args.max_nodes, search_rows=args.pipe_rows)
        pipe.initialize()  # non-holonomic table built on the device

        class _PipeAsPlanner:  # what the reporting below reads from lanes[0][3]
            num_primitives = pipe.num_primitives
            search_rows = pipe.search_rows
            nonholo_table = staticmethod(pipe.nonholo_table)
        lanes.append((c, ms_i, val_i, _PipeAsPlanner))
    for si in range(0 if pipeline_mode else n_streams):
        c = pa.Context(local_rank)
        ms_i, val_i = synthetic.upload(c, m)
        # the batches in flight share the GPU: each planner gets its share of the resident search rows (8 waves x 4 rows per CU)
        pl = pa.HybridAStarBatch(val_i, params, max_batch=B, max_nodes=args.max_nodes, search_rows=max(4, (args.search_rows // n_streams) // 4 * 4))
        pl.initialize()  # non-holonomic table built on the device
        lanes.append((c, ms_i, val_i, pl))
    ctx, ms, val, planner = lanes[0]

    # queries: uniform over valid poses; every rank its own slice of the seed space
    reach = synthetic.reachable_mask(val, m)  # drop the pockets enclosed by outline obstacles
    from pathplanning_amd import sharding
    total_queries = B * max(world, 1) if args.scaling == "weak" else args.batch
    my_ids = sharding.shard_indices(total_queries, rank, max(world, 1))  # global query ids owned by this rank (block-cyclic)
    if args.scaling == "weak":  # every rank its own slice of the seed space
        starts = synthetic.sample_valid_poses(val, m, B, seed=1000 + rank, reachable=reach)
        goals = synthetic.sample_valid_poses(val, m, B, seed=2000 + rank, reachable=reach)
    else:  # one global query set, the same on every rank; each rank plans the queries it owns
        starts = np.ascontiguousarray(synthetic.sample_valid_poses(val, m, args.batch, seed=1000, reachable=reach)[my_ids])
        goals = np.ascontiguousarray(synthetic.sample_valid_poses(val, m, args.batch, seed=2000, reachable=reach)[my_ids])
    seeds = my_ids.astype(np.uint64)
    d_starts = torch.from_numpy(starts).to(dev)
    d_goals = torch.from_numpy(goals).to(dev)
    d_seeds = torch.from_numpy(seeds.astype(np.int64)).to(dev)
    step_records = []

    def finish(pl):
        res = pl.fetch_results()  # synchronises that planner's stream
        if world > 1:
            step_records.append(sharding.records_from_results(res, B))
        return res

    def gather_all():
        # the only collective: ONE gather of the fixed-size result records of all the steps just run (RCCL all_gather).
        # One launch per run rather than per step: a collective's kernel queues behind the persistent search grids like
        # any other launch.
        if world > 1 and step_records:
            rec = np.concatenate(step_records)  # [k * B, fields] (lanes: step-major; pipeline: completion order)
            k = len(rec) // B
            rec = rec.reshape(k, B, -1).transpose(1, 0, 2).reshape(B, -1)  # one row per local query slot, k records wide
            out = sharding.gather_records(rec, total_queries, rank, world, device=dev)
            step_records.clear()
            return out
        return None

    stagger_ms = [0.0]

    def run_steps(k):
        """k steps = k batches of B queries; up to n_streams of them in flight.  A lane is refilled as soon as ITS batch is
        done (non-blocking stream query), whichever lane that is: batches differ in length, and a lane that waits for the
        host to finish with a slower one leaves its 17 GB of fields idle."""
        out, timings = None, []
        busy, free = {}, list(range(n_streams))
        started = finished = 0
        t_begin = time.perf_counter()
        prev = None
        while finished < k:
            while free and started < k:
                if started < n_streams and (time.perf_counter() - t_begin) * 1e3 < started * stagger_ms[0]:
                    break  # first round of launches: one wavefront time apart (see `stagger_ms` below)
                li = free.pop(0)
                # lanes launched together would run in phase (all wavefronts, all searches, then all tails with the GPU nearly
                # empty): each launch of the first round / every launch waits for the fields of the batch launched before it
                if prev is not None and prev != li and (args.chain == "always" or (args.chain == "first" and started < n_streams)):
                    lanes[li][3].start_after_fields_of(lanes[prev][3])
                prev = li
                lanes[li][3].search_batch_dev(d_starts, d_goals, d_seeds)  # asynchronous: wavefront + search enqueued on the lane's stream
                busy[li] = lanes[li][3]
                started += 1
            ready = [li for li in busy if lanes[li][0].is_idle()]
            if not ready:
                time.sleep(0.0005)
                continue
            for li in ready:
                pl = busy.pop(li)
                out = finish(pl)
                timings.append(pl.last_timings())
                free.append(li)
                finished += 1
        gather_all()
        return out, timings

    # steps are accounting, not barriers: when a step is smaller than a submission (--batch 512 = one GPU's share of configs[3] at
    # N = 8), one submission carries several steps' queries -- the arrays are tiled so that a submission stays one contiguous range
    rep = max(1, args.submit_chunk // B) if pipeline_mode else 1
    d_starts_rep, d_goals_rep, d_seeds_rep = (d_starts.repeat(rep, 1), d_goals.repeat(rep, 1), d_seeds.repeat(rep)) if rep > 1 else (d_starts, d_goals, d_seeds)

    backlog = []  # (ready, searching) samples of the last run_steps_pipeline
    tail_marks, tail_info = [], {}
    PATH_POSES = 256  # poses fetched per plan (the longest path of the benchmark's queries has ~150 nodes)
    path_buf = np.empty((8192, PATH_POSES, 3))
    path_n = np.zeros(8192, dtype=np.int32)
    last_paths = np.zeros((B, PATH_POSES, 3))  # the last step's plans, by query index
    last_path_n = np.zeros(B, dtype=np.int32)
    path_stats = {}
    alive_samples = []

    def run_steps_pipeline(k):
        """k steps = k x B queries through the library's pipeline: submitted as slots are free, polled in completion order.  Returns the
        last step's results (by query index) and the sums over all k steps."""
        from pathplanning_amd._lib import QUERY_RESULT_DTYPE
        total, submitted, done = k * B, 0, 0
        last = np.zeros(B, dtype=QUERY_RESULT_DTYPE)
        sums = dict(success=0, expansions=0, rs_attempts=0, rng_draws=0, state_checks=0, path_checks=0)
        base = None
        backlog.clear()
        tail_marks.clear()
        late = []
        path_stats.update(paths=0, poses=0, longest=0, truncated=0, with_solution=0)
        t_run0 = time.perf_counter()
        last_submit = [0.0]
        alive_samples.clear()
        t_alive = time.perf_counter()
        while done < total:
            if args.sample_alive and time.perf_counter() - t_alive > 0.25:  # diagnostics: is the search grid still whole?
                t_alive = time.perf_counter()
                alive_samples.append((round(t_alive - t_run0, 2), pipe.alive_waves()))
            if submitted < total:
                free = pipe.free_slots()
                off = submitted % (rep * B)
                want = min(rep * B - off, args.submit_chunk, total - submitted, pipe.capacity)  # (never more than the pipeline can hold: the loop would wait for ever)
                if free >= want:  # a submission is one wavefront launch: never a handful of goals (a launch lasts at least one goal's 20 ms)
                    first, kk = pipe.submit_dev(d_starts_rep, d_goals_rep, d_seeds_rep, n=want, offset=off)
                    if base is None:
                        base = first
                    submitted += kk
                    last_submit[0] = time.perf_counter() - t_run0
            # every plan leaves the GPU inside the measured region: the slots are held until their paths (GetGraphSearchPath: the poses of the
            # solution's nodes, start first) have been copied out of the pipeline's ring in pinned host memory, then released
            tickets, res = pipe.poll_array_held(8192)
            backlog.append(pipe.backlog())
            if len(tickets):
                poses, n_poses = pipe.get_paths(tickets, max_poses=PATH_POSES, release=True, out=path_buf, n_out=path_n)
                path_stats["paths"] += len(tickets)
                path_stats["with_solution"] = path_stats.get("with_solution", 0) + int((n_poses > 0).sum())
                path_stats["poses"] += int(np.minimum(n_poses, PATH_POSES).sum())
                path_stats["longest"] = max(path_stats["longest"], int(n_poses.max()))
                path_stats["truncated"] += int((n_poses > PATH_POSES).sum())
                idx = (tickets - np.uint64(base)).astype(np.int64)
                in_last = idx >= (k - 1) * B
                last[idx[in_last] - (k - 1) * B] = res[in_last]
                last_paths[idx[in_last] - (k - 1) * B] = poses[in_last]
                last_path_n[idx[in_last] - (k - 1) * B] = n_poses[in_last]
                sums["success"] += int((res["status"] == 0).sum())
                sums["expansions"] += int(res["n_expanded"].sum())
                sums["rs_attempts"] += int(res["n_rs_attempts"].sum())
                sums["rng_draws"] += int(res["n_rng_draws"].sum())
                sums["state_checks"] += int(res["n_state_checks"].sum())
                sums["path_checks"] += int(res["n_path_checks"].sum())
                done += len(tickets)
                tail_marks.append((time.perf_counter(), done))
                if done > 0.98 * total:  # what the run ends with: arrival time, submission index, expansions, status of the late results
                    late.extend((time.perf_counter() - t_run0, int(i), int(r["n_expanded"]), int(r["status"])) for i, r in zip(idx, res))
                if world > 1:
                    step_records.append(np.column_stack([res["status"].astype(np.float64), res["cost"], res["n_expanded"].astype(np.float64), res["n_path"].astype(np.float64)]))
            else:
                time.sleep(0.0002)
        # how the run ends: when the last submission went in, and when 90 / 99 / 99.9 / 100 % of the results had arrived
        tm = np.array([(t - t_run0, d) for t, d in tail_marks])
        tail_info.clear()
        tail_info.update(last_submission_s=last_submit[0], **{"done_%s_s" % str(f).replace(".", "_"): float(tm[np.searchsorted(tm[:, 1], f * total / 100.0), 0]) for f in (50, 90, 99, 99.9, 100)})
        # the last results to arrive: [arrival s, position of the query in the run as a fraction, expansions, status, min clearance of start / goal in m]
        qi = np.array([x[1] for x in late[-12:]], dtype=np.int64) % B
        clr = np.minimum(query_clearance(starts[qi]), query_clearance(goals[qi])) if len(qi) else []
        tail_info["last_results"] = [[round(t, 3), round(i / total, 3), ne, st, round(float(c), 2)] for (t, i, ne, st), c in zip(late[-12:], clr)]
        return last, sums

    def query_clearance(p):
        res_m = float(ms.resolution)
        r = np.clip(((p[:, 0] - ms.grid_origin[0]) / res_m).astype(np.int64), 0, ms.rows - 1)
        c = np.clip(((p[:, 1] - ms.grid_origin[1]) / res_m).astype(np.int64), 0, ms.cols - 1)
        return np.sqrt(m["d2"][r, c].astype(np.float64)) * res_m

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # every lane's first batch is cold (first touch of its 17 GB of fields, first dispatch of its kernels): the warm-up runs
    # at least one batch through each lane, so no cold batch falls into the timed region whatever --warmup says
    # (round 1's driver run, --warmup 5 with 8 lanes, timed three cold batches)
    # Lanes launched together run in phase -- eight wavefronts, then eight searches, then the tails of eight batches' longest
    # queries with the GPU nearly empty, every cycle.  Started one wavefront apart their phases interleave from the first cycle on
    # (measured with the driver's --steps 20 --warmup 5: 13.2 k plans/s together, 14.5 k staggered; 150-200 ms all within 3 %).
    # The delay is measured, not assumed: the first warm-up batch runs alone and its wavefront time is the stagger.
    pipe_kernel = None
    if pipeline_mode:
        # the pipeline has no lanes to warm one by one: the warm-up steps touch every field slot once (capacity = 6 steps' worth)
        warm = max(args.warmup, 1)
        run_steps_pipeline(warm)
        sync_all()
        pipe.timings()  # reset
    elif args.stagger_ms < 0:
        run_steps(1)
        sync_all()
        stagger_ms[0] = float(lanes[0][3].last_timings()[0])
    else:
        stagger_ms[0] = args.stagger_ms
    if not pipeline_mode:
        warm = max(args.warmup, n_streams)
        run_steps(warm)
        sync_all()
    if args.debug_skip:
        if "PP_HIP_LIB" not in os.environ:
            raise SystemExit("--debug-skip needs a diagnostic build of the library: python tools/build_variant.py skip -DPP_ENABLE_DEBUG_SKIP=1, "
                             "then PP_HIP_LIB=pathplanning_amd/lib/variants/skip.so (the shipped library has no work-skipping path)")
        os.environ["PP_DEBUG_SKIP"] = str(args.debug_skip)
    t0 = time.perf_counter()
    if pipeline_mode:
        res, run_sums = run_steps_pipeline(args.steps)
        gather_all()
    else:
        res, timings = run_steps(args.steps)
    sync_all()
    elapsed = time.perf_counter() - t0
    if pipeline_mode:
        res = res.view(np.recarray)
        pipe_kernel = pipe.timings()
        # wavefront: one launch per submission, timed by HIP events on its stream.  Search grid: ONE persistent grid that is alive from
        # the first submission to the last result -- its "launch" is the timed region itself (the launches that only top up a full
        # grid last microseconds and say nothing); what it processed in that time is the whole run's expansions.
        wf_ms = [pipe_kernel["wavefront_ms_total"] / max(1, pipe_kernel["wavefront_launches"])]
        se_ms = [elapsed * 1e3]
    else:
        wf_ms = [t[0] for t in timings]
        se_ms = [t[1] for t in timings]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_per_step = B * n_gpus if args.scaling == "weak" else args.batch
    plans_per_s = total_per_step * args.steps / elapsed
    n_success = sum(1 for r in res if r.status == 0)
    n_expanded = sum(r.n_expanded for r in res)
    n_children = n_expanded * planner.num_primitives
    state_checks = sum(r.n_state_checks for r in res)

    # ---- secondary metric: collision checks / s on streamed poses (device-resident input)
    n_chk = args.check_poses
    g = torch.Generator(device=dev)
    g.manual_seed(42 + rank)
    half = float(m["upper"][0])
    poses = torch.empty(n_chk, 3, dtype=torch.float64, device=dev)
    poses[:, 0].uniform_(-half, half, generator=g)
    poses[:, 1].uniform_(-half, half, generator=g)
    poses[:, 2].uniform_(-3.141592653589793, 3.141592653589793, generator=g)
    out = torch.empty(n_chk, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)  # the poses are filled on torch's stream, the checks run on the library's own
    lib = ctx.lib
    from pathplanning_amd._lib import check
    import ctypes as C
    for _ in range(2):
        check(lib.pp_check_states_dev(ms.h, n_chk, C.c_void_p(poses.data_ptr()), C.c_void_p(out.data_ptr())))
    ctx.synchronize()
    reps = 5
    ctx.timer_start()
    for _ in range(reps):
        check(lib.pp_check_states_dev(ms.h, n_chk, C.c_void_p(poses.data_ptr()), C.c_void_p(out.data_ptr())))
    chk_ms = ctx.timer_stop() / reps
    checks_per_s = n_chk / (chk_ms * 1e-3) * n_gpus
    chk_gbs = n_chk * CHECK_BYTES_PER_POSE / (chk_ms * 1e-3) / 1e9
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    val.count_valid_fused(n_chk, 7, cnt)
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        val.count_valid_fused(n_chk, 7, cnt)
    fused_ms = ctx.timer_stop() / reps
    del poses, out

    # ---- the wavefront kernel with the chip to itself (one launch, the step's 4096 goals, row-major fields): the kernel's own rate, next to
    # the per-launch figure of the timed region, where launches overlap and share the chip with the search grid
    wf_alone_ms = None
    if pipeline_mode and rank == 0 and not args.no_cpu_baseline:
        try:
            from pathplanning_amd.planner import ObstaclesHeuristic
            out_f = torch.empty((B, ms.rows * ms.cols), dtype=torch.float32, device=dev)
            oh = ObstaclesHeuristic(ms)
            for _ in range(2):
                oh.update_dev(goals[:, :2], out_f)
            ctx.synchronize()
            ctx.timer_start()
            oh.update_dev(goals[:, :2], out_f)
            wf_alone_ms = float(ctx.timer_stop())
            del out_f
        except Exception as e:  # (memory: the pipeline's slots are still allocated)
            wf_alone_ms = None
            print("stand-alone wavefront measurement skipped: %s" % e, file=sys.stderr)

    # ---- roofline of the dominant kernel of the step
    search_kernel = "k_hybrid_search_rows" if lanes[0][3].search_rows else "k_hybrid_search"
    wf = float(np.mean(wf_ms))
    se = float(np.mean(se_ms))
    cells = ms.rows * ms.cols
    if pipeline_mode:
        # per launch: the goals of an average wavefront launch; the search grid's one "launch" processed every expansion of the run
        wf_goals = pipe_kernel["wavefront_goals"] / max(1, pipe_kernel["wavefront_launches"])
        wf_bytes = wf_goals * cells * WAVEFRONT_BYTES_PER_CELL
        se_bytes = run_sums["expansions"] * planner.num_primitives * CHILD_BYTES
        search_kernel = "k_hybrid_search_rows<true> (persistent grid)"
    else:
        wf_bytes = B * cells * WAVEFRONT_BYTES_PER_CELL
        se_bytes = n_children * CHILD_BYTES
    wf_gbs = wf_bytes / (wf * 1e-3) / 1e9
    se_gbs = se_bytes / (se * 1e-3) / 1e9
    # (the obstacle-heuristic fields are built by the tile form of the wavefront, k_wavefront_tiles, unless PP_WF_TILES=0 sends every goal through
    # the ordered kernel k_wavefront; the key of this entry stays "k_wavefront" for the earlier rounds' records)
    WF_KERNEL = "k_wavefront" if os.environ.get("PP_WF_TILES") == "0" else "k_wavefront_tiles"
    # one roofline entry per kernel, each with ITS OWN time; `roofline` is the one that is busy longest per step
    roofs = {
        "k_wavefront": dict(kernel=WF_KERNEL, bound="hbm", achieved=wf_gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=wf_gbs / HBM_PEAK_GBS, traffic=None,
                            ms_per_launch=wf, algorithmic_bytes_per_launch=wf_bytes),
        "search": dict(kernel=search_kernel, bound="hbm", achieved=se_gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=se_gbs / HBM_PEAK_GBS, traffic=None,
                       ms_per_launch=se, algorithmic_bytes_per_launch=se_bytes),
    }
    if pipeline_mode:
        # both kernels are busy for the whole timed region; the wavefront kernel's launches cover it several times over (two streams):
        # the dominant one is the one with more busy time per step
        wf_busy_per_step = pipe_kernel["wavefront_ms_total"] / args.steps
        roofs["k_wavefront"]["busy_ms_per_step"] = wf_busy_per_step
        # launches on the two wavefront streams overlap and share the chip: `achieved` is per launch as the contract defines it (bytes of a
        # launch / its own duration); all launches in flight together move `achieved_all_launches`
        in_flight = pipe_kernel["wavefront_ms_total"] / (elapsed * 1e3)
        roofs["k_wavefront"]["launches_in_flight"] = in_flight
        roofs["k_wavefront"]["achieved_all_launches"] = wf_gbs * max(1.0, in_flight)
        if wf_alone_ms:
            alone_gbs = B * cells * WAVEFRONT_BYTES_PER_CELL / (wf_alone_ms * 1e-3) / 1e9
            roofs["k_wavefront"]["alone"] = dict(ms_per_launch=wf_alone_ms, goals=B, achieved=alone_gbs, frac=alone_gbs / HBM_PEAK_GBS)
        roofs["search"]["busy_ms_per_step"] = elapsed * 1e3 / args.steps
        # per step: the bytes of everything the kernel did in the run / the timed region (whatever its launches' overlap)
        for key, nbytes in (("k_wavefront", pipe_kernel["wavefront_goals"] * cells * WAVEFRONT_BYTES_PER_CELL), ("search", se_bytes)):
            gbs = nbytes / elapsed / 1e9
            roofs[key]["per_step"] = dict(algorithmic_bytes_per_step=nbytes / args.steps, ms_per_step=elapsed * 1e3 / args.steps, achieved=gbs, frac=gbs / HBM_PEAK_GBS)
        roofs["both_kernels_per_step"] = dict(achieved=sum(roofs[k]["per_step"]["achieved"] for k in ("k_wavefront", "search")),
                                              frac=sum(roofs[k]["per_step"]["frac"] for k in ("k_wavefront", "search")))
        roof = roofs["k_wavefront"] if wf_busy_per_step >= elapsed * 1e3 / args.steps else roofs["search"]
    else:
        roof = roofs["k_wavefront"] if wf >= se else roofs["search"]
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    # the committed PMC passes were made on the default workload: for any other map / batch the counter figure does not apply
    if os.path.exists(tj) and (args.cells, args.batch, args.obstacles) == (1024, 4096, 24):
        try:
            tr = json.load(open(tj))
            for r_ in roofs.values():
                if "kernel" in r_:
                    r_["traffic"] = tr.get(r_["kernel"].split("<")[0].split(" ")[0])
                    r_["l2_hit"] = tr.get("l2_hit", {}).get(r_["kernel"].split("<")[0].split(" ")[0])
        except Exception:
            pass

    # ---- CPU baseline: the oracle (port of the reference) on this box's cores, rank 0, N=1 only
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            ow = O.World(half, half, m["resolution"])
            ow.set_occ(m["occ"])
            ow.set_d2(m["d2"])
            ow.set_pathcost(m["path_cost"])
            # the oracle plans with ITS OWN non-holonomic table (glibc), as the reference would -- not with the device-built one
            host_cpus = os.cpu_count() or 1
            usable = host_cpus  # the hardware threads this job may actually use: affinity mask and cgroup CPU quota (a GPU box grants a share of its host)
            try:
                usable = min(usable, len(os.sched_getaffinity(0)))
            except (AttributeError, OSError):
                pass
            for cg in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
                try:
                    txt = open(cg).read().split()
                    if cg.endswith("cpu.max"):
                        quota, period = txt[0], float(txt[1])
                    else:
                        quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    if quota not in ("max", "-1") and float(quota) > 0:
                        usable = max(1, min(usable, int(float(quota) / period + 0.5)))
                    break
                except (OSError, ValueError, IndexError):
                    continue
            table, _ = O.nonholo_build(ow.lb, ow.ub, O.params_array(), threads=min(usable, 64))
            table_equal = bool(np.array_equal(table, planner.nonholo_table()))
            model = ""
            try:
                for ln in open("/proc/cpuinfo"):
                    if ln.startswith("model name"):
                        model = ln.split(":", 1)[1].strip()
                        break
            except OSError:
                pass
            # (i) one thread, (ii) all host cores (one planner instance per thread, queries sharded).  Bounded samples: the first
            # --cpu-sample queries on one thread (~2 plans/s), the first 1024 on every hardware thread.
            n1 = min(args.cpu_sample // 4 if args.cpu_sample >= 8 else args.cpu_sample, B)
            secs1, st1, cost1, nexp1 = O.hybrid_batch(ow, table, starts[:n1], goals[:n1], seeds[:n1], threads=1)
            ns = min(max(args.cpu_sample, 1024) if args.cpu_sample >= 192 else args.cpu_sample, B)
            secs, st, cost, nexp, oposes, on = O.hybrid_batch_paths(ow, table, starts[:ns], goals[:ns], seeds[:ns], threads=usable, max_poses=PATH_POSES if pipeline_mode else 256)
            agree = sum(1 for i in range(ns) if st[i] == res[i].status and (st[i] != 0 or abs(cost[i] - res[i].cost) < 1e-5) and nexp[i] == res[i].n_expanded)
            agree1 = sum(1 for i in range(n1) if st1[i] == st[i] and nexp1[i] == nexp[i])
            paths_agree = None
            if pipeline_mode:  # the plans the GPU delivered in the last step against the oracle's: node counts equal, poses within 1e-5
                paths_agree = 0
                for i in range(ns):
                    k_ = min(int(on[i]), PATH_POSES)
                    if int(on[i]) == int(last_path_n[i]) and (k_ == 0 or float(np.abs(oposes[i, :k_] - last_paths[i, :k_]).max()) < 1e-5):
                        paths_agree += 1
            cpu = dict(value=ns / secs, unit="plans/s", cores=usable, cpu_model=model, host_cpus=host_cpus, usable_cpus=usable, kind="port",
                       sample="first %d of the %d benchmark queries on %d threads = every hardware thread this job may use (affinity mask and cgroup CPU quota; the host has %d), oracle HybridAStar::Search (heap wavefront + graph search), one planner per thread" % (ns, B, usable, host_cpus),
                       all_cores=dict(value=ns / secs, threads=usable, queries=ns, seconds=secs),
                       one_thread=dict(value=n1 / secs1, threads=1, queries=n1, seconds=secs1, same_outcome_as_all_cores_run="%d/%d" % (agree1, n1)),
                       agree_with_gpu="%d/%d" % (agree, ns), paths_agree_with_gpu=None if paths_agree is None else "%d/%d" % (paths_agree, ns),
                       oracle_table="own (glibc)", device_table_identical_to_oracle_table=table_equal)
        except Exception as e:  # the bench line must still be printed
            cpu = dict(value=None, unit="plans/s", cores=0, kind="port", sample="failed: %r" % (e,))

    if rank == 0:
        out = {
            "metric": "hybrid_astar_plans_per_sec_%dx%d" % (args.cells, args.cells),
            "value": plans_per_s,
            "unit": "plans/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "warmup_run": warm,
            **({"INVALID_debug_skip": args.debug_skip} if args.debug_skip else {}),
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "Hybrid A* batch of %d start/goal pairs per GPU per step on one %dx%d map (res 0.1 m, %d rectangle outlines), P=%d constant-steer primitives + RS analytic expansion, exact-order obstacle heuristic per query; every step replays one fixed set of %d queries (seeds 0..%d) and every plan's path is copied to the host inside the timed region" % (B, args.cells, args.cells, args.obstacles, planner.num_primitives, B, B - 1),
                       "queries_per_gpu": B, "grid": [ms.rows, ms.cols], "parallelism": "query-sharded x%d" % n_gpus,
                       **({"scheduler": "library pipeline (pp_pipeline_*)", "queries_in_flight": pipe.capacity, "search_rows": pipe.search_rows} if pipeline_mode
                          else {"scheduler": "bench.py lanes", "batches_in_flight": n_streams, "lane_stagger_ms": round(stagger_ms[0], 1)})},
            "secondary": {"metric": "collision_checks_per_sec", "value": checks_per_s, "unit": "checks/s", "poses": n_chk, "ms": chk_ms,
                          "achieved_GBs": chk_gbs, "hbm_frac": chk_gbs / HBM_PEAK_GBS, "bytes_per_pose_algorithmic": CHECK_BYTES_PER_POSE,
                          "moved_GBs": n_chk * 25.0 / (chk_ms * 1e-3) / 1e9, "moved_frac": n_chk * 25.0 / (chk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "fused_checks_per_sec": n_chk / (fused_ms * 1e-3) * n_gpus,
                          "in_search_state_checks_per_sec": state_checks * n_gpus * args.steps / elapsed},
            "kernels_ms": {"k_wavefront": wf, search_kernel: se},
            "kernel_GBs": {"k_wavefront": wf_gbs, search_kernel: se_gbs},
            "batch_stats": {"success": n_success, "queries": B, "expansions": n_expanded, "children": n_children,
                            "rs_attempts": sum(r.n_rs_attempts for r in res), "rng_draws": sum(r.n_rng_draws for r in res),
                            "state_checks": state_checks, "path_checks": sum(r.n_path_checks for r in res)},
            "roofline": roof,
            "roofline_per_kernel": roofs,
            "map_build": map_info,
            # every step replays the same B queries: the run's totals must be `steps` times the last step's (a field built for the wrong goal, or a
            # result delivered twice, would show here)
            **({"pipeline_kernel_timings": pipe_kernel, "run_totals": run_sums,
                "replay_consistent": bool(run_sums["expansions"] == args.steps * n_expanded and run_sums["success"] == args.steps * n_success and
                                          run_sums["rng_draws"] == args.steps * sum(r.n_rng_draws for r in res) and run_sums["state_checks"] == args.steps * state_checks),
                "paths_fetched": path_stats.get("paths"), "paths_with_solution": path_stats.get("with_solution"), "path_poses_fetched": path_stats.get("poses"), "longest_path_nodes": path_stats.get("longest"),
                "paths_longer_than_fetched": path_stats.get("truncated"),
                **({"alive_search_waves": alive_samples} if args.sample_alive else {}),
                "run_profile": dict(tail_info), "pipeline_backlog": dict(samples=len(backlog), ready_mean=float(np.mean([b[0] for b in backlog])), ready_p10=float(np.percentile([b[0] for b in backlog], 10)),
                                         ready_max=int(max(b[0] for b in backlog)), searching_mean=float(np.mean([b[1] for b in backlog])), rows=pipe.search_rows)} if pipeline_mode else {}),
            "cpu_baseline": cpu,
        }
        print(json.dumps(out, default=lambda o: o.item() if hasattr(o, "item") else str(o)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
