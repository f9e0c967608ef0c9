#!/bin/bash
# round 4: the solo policy WITHOUT the backlog condition (PP_PIPE_SOLO_BACKLOG huge): a wave one of whose queries has passed N expansions takes nothing new at all.
# In the drain of a run the ring is empty anyway, so the conditional form cannot matter there; unconditional, only queries past N (the lattice-exhausting ones for N >= 45 k:
# ~2 % of the rows' time) cost their waves three rows, and run their last 20 k expansions at ~14 us instead of ~28.
O=gpurun_out/r4solo2; mkdir -p $O; export TMPDIR=/tmp
run() { # name, steps, env...
	local name=$1 steps=$2; shift 2
	env "$@" timeout -k 10 300 python bench.py --steps $steps --warmup 5 --no-cpu-baseline > $O/$name.json 2> $O/$name.err
	python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    b = d.get("pipeline_backlog", {}); p = d.get("run_profile", {})
    print("%-16s %8.0f plans/s  ready %6.0f  searching %6.0f  last submission %.2f s, 90/99/100 %% done %.2f / %.2f / %.2f s  consistent %s" % (sys.argv[2], d["value"], b.get("ready_mean", -1), b.get("searching_mean", -1),
          p.get("last_submission_s", -1), p.get("done_90_s", -1), p.get("done_99_s", -1), p.get("done_100_s", -1), d.get("replay_consistent")), flush=True)
except Exception as e:
    print("%-16s failed: %s" % (sys.argv[2], e), flush=True)
PY
}
B=PP_PIPE_SOLO_BACKLOG=1000000000
run off_20 20 X=1 | tee -a $O/sweep.txt
run a45k_20 20 PP_PIPE_SOLO_AFTER=45000 $B | tee -a $O/sweep.txt
run a35k_20 20 PP_PIPE_SOLO_AFTER=35000 $B | tee -a $O/sweep.txt
run a25k_20 20 PP_PIPE_SOLO_AFTER=25000 $B | tee -a $O/sweep.txt
run a15k_20 20 PP_PIPE_SOLO_AFTER=15000 $B | tee -a $O/sweep.txt
run off_20b 20 X=1 | tee -a $O/sweep.txt
run a35k_20b 20 PP_PIPE_SOLO_AFTER=35000 $B | tee -a $O/sweep.txt
run off_64 64 X=1 | tee -a $O/sweep.txt
run a35k_64 64 PP_PIPE_SOLO_AFTER=35000 $B | tee -a $O/sweep.txt

