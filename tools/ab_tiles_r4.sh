#!/bin/bash
# round 4: A/B on one box -- the library as built against the previous commit's (lib/variants/libpphip_prev.so), 20 and 64 steps, twice each;
# config 5 with more launches in flight (four wavefront streams, 3072 slots) for the two homes of the tile queue
O=gpurun_out/r4ab; mkdir -p $O; export TMPDIR=/tmp
run() { # name, steps, lib
	local name=$1 steps=$2 lib=$3
	PP_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps $steps --warmup 5 --no-cpu-baseline > $O/$name.json 2> $O/$name.err
	python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    b = d.get("pipeline_backlog", {}); p = d.get("run_profile", {}); a = d["roofline_per_kernel"]["k_wavefront"].get("alone", {})
    print("%-16s %8.0f plans/s  ready %6.0f  searching %6.0f / %d  100 %% done %.2f s  wavefront alone %.1f ms  consistent %s" % (sys.argv[2], d["value"], b.get("ready_mean", -1), b.get("searching_mean", -1), b.get("rows", 0),
          p.get("done_100_s", -1), a.get("ms_per_launch", -1), d.get("replay_consistent")), flush=True)
except Exception as e:
    print("%-16s failed: %s" % (sys.argv[2], e), flush=True)
PY
}
N=pathplanning_amd/lib/libpphip.so; V=pathplanning_amd/lib/variants/libpphip_prev.so
for i in 1 2; do
run new_20_$i 20 $N | tee -a $O/ab.txt
run prev_20_$i 20 $V | tee -a $O/ab.txt
run new_64_$i 64 $N | tee -a $O/ab.txt
run prev_64_$i 64 $V | tee -a $O/ab.txt
done
c5() { # name, capacity, env...
	local name=$1 cap=$2; shift 2
	(env "$@" timeout -k 10 400 python bench.py --cells 4096 --obstacles 384 --batch 512 --capacity $cap --pipe-rows 1024 --max-nodes 262144 --steps 8 --warmup 2 --no-cpu-baseline > $O/$name.json 2> $O/$name.err &)
	for i in $(seq 1 45); do sleep 10; echo "tick $i"; if [ -s $O/$name.json ]; then break; fi; done
	python -c "
import json
d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['batch_stats']['success'], d['pipeline_backlog'])" | tee -a $O/ab.txt
}
c5 config5_global_4streams_3072 3072 PP_PIPE_WF_STREAMS=4
c5 config5_lds_4streams_3072 3072 PP_PIPE_WF_STREAMS=4 PP_WF_TILES_QUEUE=lds

