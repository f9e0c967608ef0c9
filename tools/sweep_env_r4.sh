#!/bin/bash
# round 4: driver-style bench runs under environment settings, one line per run
# usage: tools/sweep_env_r4.sh out.txt "ROWS" "VAR=val VAR=val" "VAR=val" ...
out=$1; rows=$2; shift 2
for setting in "$@"; do
  for r in $rows; do
    env $setting timeout -k 10 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipe-rows $r --check-poses 1048576 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
b=d.get('pipeline_backlog',{})
print('rows $r [$setting] value %.0f ms/step %.1f ready_mean %.0f searching %.0f last %.2fs consistent %s' % (d['value'], d['ms_per_step'], b.get('ready_mean',-1), b.get('searching_mean',-1), d['run_profile']['done_100_s'], d.get('replay_consistent')))" >> $out 2>&1
    tail -1 $out
  done
done
