#!/bin/bash
# round 4, final figures (2/2): counter traffic + L2 hit rate (three --pmc passes), the strong-scaling share, config 5
O=gpurun_out/r4f; mkdir -p $O; export TMPDIR=/tmp
bash tools/pmc_traffic.sh r4f > $O/pmc.log 2>&1; tail -3 $O/pmc.log; cp gpurun_out/pmc_r4f/traffic.json $O/traffic.json
timeout -k 10 200 python bench.py --batch 512 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_strong_share.json 2> /dev/null; echo "share 20 done"
timeout -k 10 200 python bench.py --batch 512 --steps 160 --warmup 5 --no-cpu-baseline > $O/bench_strong_share_160.json 2> /dev/null; echo "share 160 done"
c5() { # name, capacity, extra bench args, env...
	local name=$1 cap=$2 extra=$3; shift 3
	(env "$@" timeout -k 10 400 python bench.py --cells 4096 --obstacles 384 --batch 512 --capacity $cap --pipe-rows 1024 --max-nodes 262144 --steps 8 --warmup 2 $extra > $O/$name.json 2> $O/$name.err; echo finished >> $O/$name.err) &
	for i in $(seq 1 45); do sleep 10; echo "tick $i"; if grep -q finished $O/$name.err; then break; fi; done
	python -c "
import json
d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['batch_stats']['success'], d['pipeline_backlog'], d.get('cpu_baseline'))" | tee -a $O/config5.txt
}
c5 bench_config5 1536 "--cpu-sample 4" X=1
python -c "
import json
def L(f): return json.loads(open('$O/%s.json'%f).read().strip().splitlines()[-1])
d=L('bench_strong_share'); print('share 20', d['value'], d['run_profile'])
d=L('bench_strong_share_160'); print('share 160', d['value'])
print(open('$O/traffic.json').read()[:1500])
"
