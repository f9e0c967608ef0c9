#!/bin/bash
# round 4: the whole GPU suite with the final build, then config 5 (ordered kernel above 2048^2) and the two strong-share runs
O=gpurun_out/r4n; mkdir -p $O; export TMPDIR=/tmp
(timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "exit $?" >> $O/gpu_tests.log) &
P=$!
while kill -0 $P 2>/dev/null; do sleep 30; echo "tests: $(tail -c 120 $O/gpu_tests.log | tr '\n' ' ')"; done
tail -5 $O/gpu_tests.log
(timeout -k 10 500 python bench.py --cells 4096 --obstacles 384 --batch 512 --capacity 1536 --pipe-rows 1024 --max-nodes 262144 --steps 8 --warmup 2 --cpu-sample 4 > $O/bench_config5.json 2> $O/bench_config5.err &)
for i in $(seq 1 50); do sleep 10; echo "tick $i"; if [ -s $O/bench_config5.json ]; then break; fi; done
python -c "
import json
def L(f): return json.loads(open('$O/%s.json'%f).read().strip().splitlines()[-1])
d=L('bench_config5'); print('config5', d['value'], d['ms_per_step'], d['batch_stats']['success'], d['cpu_baseline'], d['map_build'], d['pipeline_backlog'])
"
