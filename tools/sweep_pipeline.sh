#!/bin/bash
# usage: tools/sweep_pipeline.sh OUT "rows:wfblocks:chunk[:lib[:VAR=value]]" ...   -- one driver-style bench run per configuration (GPU box)
out=$1; shift
for cfg in "$@"; do
  IFS=: read rows blocks chunk lib extra <<< "$cfg"
  tag=${rows}_${blocks}_${chunk}_${lib:-default}_${extra//=/-}
  env PP_PIPE_WF_BLOCKS=$blocks ${extra:+$extra} ${lib:+PP_HIP_LIB=pathplanning_amd/lib/variants/$lib.so} timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --check-poses 1048576 \
    --pipe-rows $rows --submit-chunk $chunk > gpurun_out/sw_$tag.json 2> gpurun_out/sw_$tag.err || echo "FAILED $cfg" >> $out
  python - "$cfg" gpurun_out/sw_$tag.json >> $out <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    t = d["pipeline_kernel_timings"]
    print("%-28s plans/s %6.0f  ms/step %6.1f  wavefront busy %6.0f ms in %d launches" % (sys.argv[1], d["value"], d["ms_per_step"], t["wavefront_ms_total"], t["wavefront_launches"]))
except Exception as e:
    print(sys.argv[1], "no result:", e)
PY
done
cat $out
