#!/bin/bash
# round 4: tile-wave workgroups (waves per pack, LDS asked for) against search rows, now that the search grid's waves stay for the whole run;
# the tile queue in global memory (test + config 5)
O=gpurun_out/r4packs; mkdir -p $O; export TMPDIR=/tmp
(timeout -k 10 900 python -m pytest tests/test_gpu_wavefront_tiles.py tests/test_gpu_pipeline.py -x -q > $O/gpu_tests.log 2>&1; echo "exit $?" >> $O/gpu_tests.log) &
P=$!
while kill -0 $P 2>/dev/null; do sleep 30; echo "tests: $(tail -c 100 $O/gpu_tests.log | tr '\n' ' ')"; done
tail -3 $O/gpu_tests.log
run() { # name, steps, rows, env...
	local name=$1 steps=$2 rows=$3; shift 3
	env "$@" timeout -k 10 300 python bench.py --steps $steps --warmup 5 --no-cpu-baseline --pipe-rows $rows > $O/$name.json 2> $O/$name.err
	python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    b = d.get("pipeline_backlog", {}); p = d.get("run_profile", {})
    print("%-26s %8.0f plans/s  ready %6.0f  searching %6.0f / %d  100 %% done %.2f s  consistent %s" % (sys.argv[2], d["value"], b.get("ready_mean", -1), b.get("searching_mean", -1), b.get("rows", 0),
          p.get("done_100_s", -1), d.get("replay_consistent")), flush=True)
except Exception as e:
    print("%-26s failed: %s" % (sys.argv[2], e), flush=True)
PY
}
run p8_4096_20 20 4096 X=1 | tee -a $O/sweep.txt
run p1_4096_20 20 4096 PP_WF_TILES_PACK=1 | tee -a $O/sweep.txt
run p1_4608_20 20 4608 PP_WF_TILES_PACK=1 | tee -a $O/sweep.txt
run p1_5120_20 20 5120 PP_WF_TILES_PACK=1 | tee -a $O/sweep.txt
run p2kb24_4608_20 20 4608 PP_WF_TILES_PACK=2 PP_WF_TILES_PACK_KB=24 | tee -a $O/sweep.txt
run p4kb44_4608_20 20 4608 PP_WF_TILES_PACK=4 PP_WF_TILES_PACK_KB=44 | tee -a $O/sweep.txt
run p4kb44_5120_20 20 5120 PP_WF_TILES_PACK=4 PP_WF_TILES_PACK_KB=44 | tee -a $O/sweep.txt
run p8_4096_64 64 4096 X=1 | tee -a $O/sweep.txt
run p1_4608_64 64 4608 PP_WF_TILES_PACK=1 | tee -a $O/sweep.txt
run p1_5120_64 64 5120 PP_WF_TILES_PACK=1 | tee -a $O/sweep.txt
c5() { # name, env...
	local name=$1; shift
	(env "$@" timeout -k 10 400 python bench.py --cells 4096 --obstacles 384 --batch 512 --capacity 1536 --pipe-rows 1024 --max-nodes 262144 --steps 8 --warmup 2 --no-cpu-baseline > $O/$name.json 2> $O/$name.err &)
	for i in $(seq 1 45); do sleep 10; echo "tick $i"; if [ -s $O/$name.json ]; then break; fi; done
	python -c "
import json
d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['batch_stats']['success'], d['pipeline_backlog'])" | tee -a $O/sweep.txt
}
c5 config5_queue_global X=1
c5 config5_queue_lds PP_WF_TILES_QUEUE=lds
