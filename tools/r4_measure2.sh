#!/bin/bash
# round 4, final figures: driver-style line with the CPU leg, 64 steps, kernel stats + timeline of the driver-style run, config 5
O=gpurun_out/r4m; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "driver-style done"
timeout -k 10 300 python bench.py --steps 64 --warmup 5 --no-cpu-baseline > $O/bench_64_steps.json 2> /dev/null; echo "64 steps done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_trace.json 2> /dev/null; echo "trace done"
S=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp $S $O/bench_kernel_stats.csv
K=$(find $O/trace -name "*kernel_trace.csv" | head -1); python tools/trace_timeline.py $K --timeline > $O/bench_timeline.txt 2>&1; rm -rf $O/trace
(timeout -k 10 500 python bench.py --cells 4096 --obstacles 384 --batch 512 --capacity 1536 --pipe-rows 1024 --max-nodes 262144 --steps 8 --warmup 2 --cpu-sample 4 > $O/bench_config5.json 2> $O/bench_config5.err &)
for i in $(seq 1 50); do sleep 10; echo "tick $i"; if [ -s $O/bench_config5.json ]; then break; fi; done
python -c "
import json
def L(f): return json.loads(open('$O/%s.json'%f).read().strip().splitlines()[-1])
d=L('bench_driver_style'); print({k:d.get(k) for k in ['value','ms_per_step','paths_fetched','replay_consistent']}, d['cpu_baseline'].get('all_cores'), d['cpu_baseline'].get('one_thread'), d['cpu_baseline'].get('agree_with_gpu'), d['cpu_baseline'].get('paths_agree_with_gpu'), d['pipeline_backlog'])
print(d['roofline_per_kernel']['k_wavefront'])
d=L('bench_64_steps'); print('64 steps', d['value'], d['ms_per_step'], d['pipeline_backlog'])
d=L('bench_config5'); print('config5', d['value'], d['ms_per_step'], d['batch_stats']['success'], d['cpu_baseline'], d['map_build'])
"
tail -12 $O/bench_timeline.txt
