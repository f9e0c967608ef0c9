#!/bin/bash
# round 4: issue priority 3 for the waves that host a query past N expansions (PP_PIPE_BOOST_AFTER): do the run's longest chains get through a full chip faster?
O=gpurun_out/r4boost; mkdir -p $O; export TMPDIR=/tmp
run() { # name, steps, extra bench args, env...
	local name=$1 steps=$2 extra=$3; shift 3
	env "$@" timeout -k 10 300 python bench.py --steps $steps --warmup 5 --no-cpu-baseline $extra > $O/$name.json 2> $O/$name.err
	python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    b = d.get("pipeline_backlog", {}); p = d.get("run_profile", {})
    print("%-14s %8.0f plans/s  ready %6.0f  searching %6.0f  last submission %.2f s, 90/99/100 %% done %.2f / %.2f / %.2f s  consistent %s" % (sys.argv[2], d["value"], b.get("ready_mean", -1), b.get("searching_mean", -1),
          p.get("last_submission_s", -1), p.get("done_90_s", -1), p.get("done_99_s", -1), p.get("done_100_s", -1), d.get("replay_consistent")), flush=True)
except Exception as e:
    print("%-14s failed: %s" % (sys.argv[2], e), flush=True)
PY
}
run off_20 20 "" X=1 | tee -a $O/sweep.txt
run b20k_20 20 "" PP_PIPE_BOOST_AFTER=20000 | tee -a $O/sweep.txt
run b5k_20 20 "" PP_PIPE_BOOST_AFTER=5000 | tee -a $O/sweep.txt
run b40k_20 20 "" PP_PIPE_BOOST_AFTER=40000 | tee -a $O/sweep.txt
run off_20b 20 "" X=1 | tee -a $O/sweep.txt
run b20k_20b 20 "" PP_PIPE_BOOST_AFTER=20000 | tee -a $O/sweep.txt
run off_64 64 "" X=1 | tee -a $O/sweep.txt
run b20k_64 64 "" PP_PIPE_BOOST_AFTER=20000 | tee -a $O/sweep.txt
run off_share 20 "--batch 512" X=1 | tee -a $O/sweep.txt
run b20k_share 20 "--batch 512" PP_PIPE_BOOST_AFTER=20000 | tee -a $O/sweep.txt
run b5k_share 20 "--batch 512" PP_PIPE_BOOST_AFTER=5000 | tee -a $O/sweep.txt
