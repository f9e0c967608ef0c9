#!/bin/bash
# round 4, final build: the order-of-work knob (PP_PIPE_URGENT_CLEARANCE in metres; default: twice the validator's minimum safe radius = 2 m, tuned in round 3) again
O=gpurun_out/r4urg; mkdir -p $O; export TMPDIR=/tmp
run() { # name, env...
	local name=$1; shift
	env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/$name.json 2> $O/$name.err
	python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    b = d.get("pipeline_backlog", {}); p = d.get("run_profile", {})
    print("%-12s %8.0f plans/s  ready %6.0f  searching %6.0f  90/99/100 %% done %.2f / %.2f / %.2f s  consistent %s" % (sys.argv[2], d["value"], b.get("ready_mean", -1), b.get("searching_mean", -1),
          p.get("done_90_s", -1), p.get("done_99_s", -1), p.get("done_100_s", -1), d.get("replay_consistent")), flush=True)
except Exception as e:
    print("%-12s failed: %s" % (sys.argv[2], e), flush=True)
PY
}
run default X=1 | tee -a $O/sweep.txt
run off PP_PIPE_URGENT_CLEARANCE=0 | tee -a $O/sweep.txt
run c1.5 PP_PIPE_URGENT_CLEARANCE=1.5 | tee -a $O/sweep.txt
run c3 PP_PIPE_URGENT_CLEARANCE=3 | tee -a $O/sweep.txt
run c5 PP_PIPE_URGENT_CLEARANCE=5 | tee -a $O/sweep.txt
run default_b X=1 | tee -a $O/sweep.txt
