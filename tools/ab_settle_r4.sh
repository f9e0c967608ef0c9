#!/bin/bash
# round 4: the tile kernel's diagonal-move test as v_bfe_i32 + v_or (pass loop 96 -> 78 instructions) against the build before it; tile + parity tests first
O=gpurun_out/r4settle; mkdir -p $O; export TMPDIR=/tmp
(timeout -k 10 900 python -m pytest tests/test_gpu_wavefront_tiles.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > $O/gpu_tests.log 2>&1; echo "exit $?" >> $O/gpu_tests.log) &
P=$!
while kill -0 $P 2>/dev/null; do sleep 30; echo "tests: $(tail -c 100 $O/gpu_tests.log | tr '\n' ' ')"; done
tail -3 $O/gpu_tests.log
run() { # name, steps, lib
	local name=$1 steps=$2 lib=$3
	PP_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps $steps --warmup 5 --no-cpu-baseline > $O/$name.json 2> $O/$name.err
	python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    b = d.get("pipeline_backlog", {}); p = d.get("run_profile", {}); a = d["roofline_per_kernel"]["k_wavefront"].get("alone", {})
    print("%-12s %8.0f plans/s  ready %6.0f  searching %6.0f / %d  100 %% done %.2f s  wavefront alone %.1f ms  consistent %s" % (sys.argv[2], d["value"], b.get("ready_mean", -1), b.get("searching_mean", -1), b.get("rows", 0),
          p.get("done_100_s", -1), a.get("ms_per_launch", -1), d.get("replay_consistent")), flush=True)
except Exception as e:
    print("%-12s failed: %s" % (sys.argv[2], e), flush=True)
PY
}
N=pathplanning_amd/lib/libpphip.so; V=pathplanning_amd/lib/variants/libpphip_prev.so
for i in 1 2; do
run new_20_$i 20 $N | tee -a $O/ab.txt
run prev_20_$i 20 $V | tee -a $O/ab.txt
run new_64_$i 64 $N | tee -a $O/ab.txt
run prev_64_$i 64 $V | tee -a $O/ab.txt
done
python tools/diag_wavefront_tiles.py > $O/tiles_diag_new.txt 2>&1; tail -6 $O/tiles_diag_new.txt
PP_HIP_LIB=$V python tools/diag_wavefront_tiles.py > $O/tiles_diag_prev.txt 2>&1; tail -6 $O/tiles_diag_prev.txt
