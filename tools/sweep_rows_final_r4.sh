#!/bin/bash
# round 4, final build (the tile stage has slack now: ready queue in the thousands): search rows beyond 4096
O=gpurun_out/r4rows; mkdir -p $O; export TMPDIR=/tmp
run() { # name, steps, rows
	local name=$1 steps=$2 rows=$3
	timeout -k 10 300 python bench.py --steps $steps --warmup 5 --no-cpu-baseline --pipe-rows $rows > $O/$name.json 2> $O/$name.err
	python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    b = d.get("pipeline_backlog", {}); p = d.get("run_profile", {})
    print("%-14s %8.0f plans/s  ready %6.0f  searching %6.0f / %d  100 %% done %.2f s  consistent %s" % (sys.argv[2], d["value"], b.get("ready_mean", -1), b.get("searching_mean", -1), b.get("rows", 0), p.get("done_100_s", -1), d.get("replay_consistent")), flush=True)
except Exception as e:
    print("%-14s failed: %s" % (sys.argv[2], e), flush=True)
PY
}
run r4096_64 64 4096 | tee -a $O/sweep.txt
run r4352_64 64 4352 | tee -a $O/sweep.txt
run r4608_64 64 4608 | tee -a $O/sweep.txt
run r5120_64 64 5120 | tee -a $O/sweep.txt
run r4096_20 20 4096 | tee -a $O/sweep.txt
run r4352_20 20 4352 | tee -a $O/sweep.txt
run r4608_20 20 4608 | tee -a $O/sweep.txt
