#!/bin/bash
# tools/sweep_env.sh TAG "ENV1=a ENV2=b" "ENV1=c" ...  -> one bench.py run per environment setting, summary lines on stdout
TAG=$1; shift
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 200 python bench.py --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/sw_${TAG}_$i.log 2>&1 || { echo "$e FAILED"; tail -3 gpurun_out/sw_${TAG}_$i.log; exit 1; }
  python3 - "$e" gpurun_out/sw_${TAG}_$i.log <<'PY'
import json, sys
for l in open(sys.argv[2]):
    if l.startswith("{"):
        d = json.loads(l)
        print("%-50s %8.0f plans/s %7.1f ms/step  wf %6.0f  search %6.0f" % (sys.argv[1], d["value"], d["ms_per_step"], d["kernels_ms"]["k_wavefront"], list(d["kernels_ms"].values())[1]))
PY
done
