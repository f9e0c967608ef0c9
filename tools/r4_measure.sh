#!/bin/bash
# round 4: the measurements the docs cite, in one go on the GPU box (outputs under gpurun_out/r4m/)
O=gpurun_out/r4m; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
for w in 32 64; do PP_WF_TILE_WIDTH=$w timeout -k 10 200 python tools/diag_wavefront_tiles.py 4096 1024 24 3 > $O/wavefront_tiles_width$w.json 2>/dev/null; done
PP_WF_TILES=0 timeout -k 10 200 python tools/diag_wavefront_tiles.py 4096 1024 24 3 > $O/wavefront_ordered.json 2>/dev/null
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
timeout -k 10 600 python bench.py --steps 64 --warmup 5 --no-cpu-baseline > $O/bench_64_steps.json 2> /dev/null
bash tools/pmc_traffic.sh r04b > $O/pmc.log 2>&1; cp gpurun_out/pmc_r04b/traffic.json $O/traffic_lanes_pass.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_trace.json 2> /dev/null
S=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp $S $O/bench_kernel_stats.csv
K=$(find $O/trace -name "*kernel_trace.csv" | head -1); python tools/trace_timeline.py $K --timeline > $O/bench_timeline.txt 2>&1; rm -rf $O/trace
python -c "
import json
d=json.loads(open('$O/bench_driver_style.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','paths_fetched','replay_consistent']}, d['cpu_baseline'].get('all_cores'), d['cpu_baseline'].get('one_thread'), d['cpu_baseline'].get('paths_agree_with_gpu'))
d=json.loads(open('$O/bench_64_steps.json').read().strip().splitlines()[-1]); print('64 steps', d['value'], d['ms_per_step'])
for w in (32,64):
    d=json.load(open('$O/wavefront_tiles_width%d.json'%w)); print('tiles width',w,d['ms_per_launch'],d['all_runs_ms'],d['visits_per_tile'],d['rounds_per_visit'],d['passes_per_round'],d['handed_over'])
print(open('$O/wavefront_ordered.json').read())
"
