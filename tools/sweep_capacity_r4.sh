#!/bin/bash
# round 4: field slots in flight (bench.py --capacity; 4 MB each at 1024^2) against the end of a 20-step run: with more slots the last submission -- and with it
# the run's last lattice-exhausting queries -- starts earlier
O=gpurun_out/r4cap; mkdir -p $O; export TMPDIR=/tmp
run() { # name, steps, capacity
	local name=$1 steps=$2 cap=$3
	timeout -k 10 300 python bench.py --steps $steps --warmup 5 --no-cpu-baseline --capacity $cap > $O/$name.json 2> $O/$name.err
	python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    b = d.get("pipeline_backlog", {}); p = d.get("run_profile", {})
    print("%-16s %8.0f plans/s  ready %6.0f  searching %6.0f / %d  last submission %.2f s, 50 / 90 / 99 / 100 %% done %.2f / %.2f / %.2f / %.2f s  consistent %s" % (sys.argv[2], d["value"], b.get("ready_mean", -1), b.get("searching_mean", -1), b.get("rows", 0),
          p.get("last_submission_s", -1), p.get("done_50_s", -1), p.get("done_90_s", -1), p.get("done_99_s", -1), p.get("done_100_s", -1), d.get("replay_consistent")), flush=True)
except Exception as e:
    print("%-16s failed: %s %s" % (sys.argv[2], e, open(sys.argv[1].replace('.json', '.err')).read()[-300:]), flush=True)
PY
}
run cap24576_20 20 24576 | tee -a $O/sweep.txt
run cap32768_20 20 32768 | tee -a $O/sweep.txt
run cap40960_20 20 40960 | tee -a $O/sweep.txt
run cap49152_20 20 49152 | tee -a $O/sweep.txt
run cap40960_64 64 40960 | tee -a $O/sweep.txt
run cap24576_20b 20 24576 | tee -a $O/sweep.txt
run cap40960_20b 20 40960 | tee -a $O/sweep.txt
