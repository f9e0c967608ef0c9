#!/bin/bash
# round 4, final figures (1/2): the whole GPU suite, the driver-style line with the CPU leg, 64 steps, kernel stats + timeline of a driver-style run
O=gpurun_out/r4f; mkdir -p $O; export TMPDIR=/tmp
(timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "exit $?" >> $O/gpu_tests.log) &
P=$!
while kill -0 $P 2>/dev/null; do sleep 30; echo "tests: $(tail -c 100 $O/gpu_tests.log | tr '\n' ' ')"; done
tail -3 $O/gpu_tests.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "driver-style done"
timeout -k 10 300 python bench.py --steps 64 --warmup 5 --no-cpu-baseline > $O/bench_64_steps.json 2> /dev/null; echo "64 steps done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_trace.json 2> /dev/null; echo "trace done"
S=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp $S $O/bench_kernel_stats.csv
K=$(find $O/trace -name "*kernel_trace.csv" | head -1); python tools/trace_timeline.py $K --timeline > $O/bench_timeline.txt 2>&1; rm -rf $O/trace
python -c "
import json
def L(f): return json.loads(open('$O/%s.json'%f).read().strip().splitlines()[-1])
d=L('bench_driver_style'); print({k:d.get(k) for k in ['value','ms_per_step','paths_fetched','replay_consistent']}, d['cpu_baseline'].get('all_cores'), d['cpu_baseline'].get('one_thread'), d['cpu_baseline'].get('agree_with_gpu'), d['cpu_baseline'].get('paths_agree_with_gpu'), d['pipeline_backlog'])
print(d['roofline_per_kernel']['k_wavefront'])
d=L('bench_64_steps'); print('64 steps', d['value'], d['ms_per_step'], d['pipeline_backlog'])
d=L('bench_under_trace'); print('under trace', d['value'])
"
tail -12 $O/bench_timeline.txt
