#!/bin/bash
# round 4: waves with a long query take nothing new while the ready ring is short (PP_PIPE_SOLO_AFTER expansions, PP_PIPE_SOLO_BACKLOG fields)
O=gpurun_out/r4solo; mkdir -p $O; export TMPDIR=/tmp
(timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_hybrid.py -x -q > $O/gpu_tests.log 2>&1; echo "exit $?" >> $O/gpu_tests.log) &
P=$!
while kill -0 $P 2>/dev/null; do sleep 30; echo "tests: $(tail -c 100 $O/gpu_tests.log | tr '\n' ' ')"; done
tail -3 $O/gpu_tests.log
run() { # name, steps, extra bench args (quoted), env...
	local name=$1 steps=$2 extra=$3; shift 3
	env "$@" timeout -k 10 300 python bench.py --steps $steps --warmup 5 --no-cpu-baseline $extra > $O/$name.json 2> $O/$name.err
	python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    b = d.get("pipeline_backlog", {}); p = d.get("run_profile", {})
    print("%-26s %8.0f plans/s  ready %6.0f  searching %6.0f  last submission %.2f s, 90/99/100 %% done %.2f / %.2f / %.2f s  consistent %s" % (sys.argv[2], d["value"], b.get("ready_mean", -1), b.get("searching_mean", -1),
          p.get("last_submission_s", -1), p.get("done_90_s", -1), p.get("done_99_s", -1), p.get("done_100_s", -1), d.get("replay_consistent")), flush=True)
except Exception as e:
    print("%-26s failed: %s" % (sys.argv[2], e), flush=True)
PY
}
run off_20 20 "" PP_PIPE_SOLO_AFTER=0 | tee -a $O/sweep.txt
run a40k_b256_20 20 "" X=1 | tee -a $O/sweep.txt
run a20k_b256_20 20 "" PP_PIPE_SOLO_AFTER=20000 | tee -a $O/sweep.txt
run a30k_b1024_20 20 "" PP_PIPE_SOLO_AFTER=30000 PP_PIPE_SOLO_BACKLOG=1024 | tee -a $O/sweep.txt
run a10k_b64_20 20 "" PP_PIPE_SOLO_AFTER=10000 PP_PIPE_SOLO_BACKLOG=64 | tee -a $O/sweep.txt
run off_64 64 "" PP_PIPE_SOLO_AFTER=0 | tee -a $O/sweep.txt
run a40k_b256_64 64 "" X=1 | tee -a $O/sweep.txt
run a20k_b256_64 64 "" PP_PIPE_SOLO_AFTER=20000 | tee -a $O/sweep.txt
run off_share 20 "--batch 512" PP_PIPE_SOLO_AFTER=0 | tee -a $O/sweep.txt
run a40k_share 20 "--batch 512" X=1 | tee -a $O/sweep.txt
run a20k_share 20 "--batch 512" PP_PIPE_SOLO_AFTER=20000 | tee -a $O/sweep.txt
