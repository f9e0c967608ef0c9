#!/bin/bash
# round 4: the search grid at 3 / 4 waves per SIMD (168 / 128 VGPRs, variants built with -DPP_SEARCH_WAVES_PER_SIMD) against rows and the tile packs' LDS
# the variant libraries are not kept: build them with
#   PP_EXTRA_HIPCC_FLAGS=-DPP_SEARCH_WAVES_PER_SIMD=4 python tools/build_variant.py ...   (or hipcc with pathplanning_amd/build.py's FLAGS, -o pathplanning_amd/lib/variants/libpphip_w4.so;
#   -DPP_ROWS_STATS=1 for libpphip_stats.so); PP_HIP_LIB selects the library a process loads
O=gpurun_out/r4occ; mkdir -p $O; export TMPDIR=/tmp
run() { # name, lib, rows, extra env...
	local name=$1 lib=$2 rows=$3; shift 3
	env "$@" PP_HIP_LIB=$lib timeout -k 10 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipe-rows $rows > $O/$name.json 2> $O/$name.err
	python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    b = d.get("pipeline_backlog", {})
    print("%-28s %8.0f plans/s  ready %6.0f  searching %6.0f / %d  consistent %s" % (sys.argv[2], d["value"], b.get("ready_mean", -1), b.get("searching_mean", -1), b.get("rows", 0), d.get("replay_consistent")), flush=True)
except Exception as e:
    print("%-28s failed: %s" % (sys.argv[2], e), flush=True)
PY
}
V=pathplanning_amd/lib/variants
run base_4096 pathplanning_amd/lib/libpphip.so 4096 X=1 | tee -a $O/sweep.txt
run w4_4096 $V/libpphip_w4.so 4096 X=1 | tee -a $O/sweep.txt
run w4_6144_kb64 $V/libpphip_w4.so 6144 PP_WF_TILES_PACK_KB=64 | tee -a $O/sweep.txt
run w4_8192_kb46 $V/libpphip_w4.so 8192 PP_WF_TILES_PACK_KB=46 | tee -a $O/sweep.txt
run w4_8192_kb46_p4 $V/libpphip_w4.so 8192 PP_WF_TILES_PACK_KB=46 PP_WF_TILES_PACK=4 | tee -a $O/sweep.txt
run w4_6144_kb46_p4 $V/libpphip_w4.so 6144 PP_WF_TILES_PACK_KB=46 PP_WF_TILES_PACK=4 | tee -a $O/sweep.txt
run w3_4096 $V/libpphip_w3.so 4096 X=1 | tee -a $O/sweep.txt
run w3_6144_kb64 $V/libpphip_w3.so 6144 PP_WF_TILES_PACK_KB=64 | tee -a $O/sweep.txt
# phase clocks of the search wave inside the pipeline (diagnostic build -DPP_ROWS_STATS=1: printed when the planner is destroyed)
run stats_4096 $V/libpphip_stats.so 4096 X=1 | tee -a $O/sweep.txt
grep "rows stats" $O/stats_4096.err | tee $O/rows_phase_stats.txt
# config 5 (4096^2 map, 512 queries per step, three submissions in flight): ordered kernel (the default above 2048^2) and the tile form (PP_WF_TILES=2)
c5() { # name, env...
	local name=$1; shift
	(env "$@" timeout -k 10 400 python bench.py --cells 4096 --obstacles 384 --batch 512 --capacity 1536 --pipe-rows 1024 --max-nodes 262144 --steps 8 --warmup 2 --no-cpu-baseline > $O/$name.json 2> $O/$name.err &)
	for i in $(seq 1 45); do sleep 10; echo "tick $i"; if [ -s $O/$name.json ]; then break; fi; done
	python -c "
import json
d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['batch_stats']['success'], d['pipeline_backlog'])" | tee -a $O/sweep.txt
}
c5 config5_ordered X=1
c5 config5_tiles PP_WF_TILES=2
