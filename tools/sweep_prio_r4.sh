#!/bin/bash
# round 4 (final pipeline): issue priorities of the tile waves (PP_WF_TILES_PRIO 0..2) and of the search rows (variant -DPP_ROWS_PRIO=2) -- the steady state is
# search-bound now; plus this round's new pipeline tests
O=gpurun_out/r4prio; mkdir -p $O; export TMPDIR=/tmp
(timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_wavefront_tiles.py -x -q > $O/gpu_tests.log 2>&1; echo "exit $?" >> $O/gpu_tests.log) &
P=$!
while kill -0 $P 2>/dev/null; do sleep 30; echo "tests: $(tail -c 100 $O/gpu_tests.log | tr '\n' ' ')"; done
tail -3 $O/gpu_tests.log
run() { # name, steps, lib, env...
	local name=$1 steps=$2 lib=$3; shift 3
	env "$@" PP_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps $steps --warmup 5 --no-cpu-baseline > $O/$name.json 2> $O/$name.err
	python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    b = d.get("pipeline_backlog", {}); p = d.get("run_profile", {})
    print("%-22s %8.0f plans/s  ready %6.0f  searching %6.0f / %d  100 %% done %.2f s  consistent %s" % (sys.argv[2], d["value"], b.get("ready_mean", -1), b.get("searching_mean", -1), b.get("rows", 0), p.get("done_100_s", -1), d.get("replay_consistent")), flush=True)
except Exception as e:
    print("%-22s failed: %s" % (sys.argv[2], e), flush=True)
PY
}
N=pathplanning_amd/lib/libpphip.so; R2=pathplanning_amd/lib/variants/libpphip_rowsprio2.so
run tiles1_rows1_64 64 $N X=1 | tee -a $O/sweep.txt
run tiles0_rows1_64 64 $N PP_WF_TILES_PRIO=0 | tee -a $O/sweep.txt
run tiles2_rows1_64 64 $N PP_WF_TILES_PRIO=2 | tee -a $O/sweep.txt
run tiles1_rows2_64 64 $R2 X=1 | tee -a $O/sweep.txt
run tiles2_rows2_64 64 $R2 PP_WF_TILES_PRIO=2 | tee -a $O/sweep.txt
run tiles1_rows1_20 20 $N X=1 | tee -a $O/sweep.txt
run tiles0_rows1_20 20 $N PP_WF_TILES_PRIO=0 | tee -a $O/sweep.txt
run tiles1_rows2_20 20 $R2 X=1 | tee -a $O/sweep.txt
run tiles2_rows2_20 20 $R2 PP_WF_TILES_PRIO=2 | tee -a $O/sweep.txt
