# the two order-of-work knobs of the pipeline against the end of a 20-step run (results are the same whatever they are set to)
for cfg in "2.0 32768" "2.0 16384" "2.0 8192" "2.0 0" "0 32768" "3.0 24576"; do
  set -- $cfg
  echo "urgent clearance $1 m, exclusive after $2 expansions"
  PP_PIPE_URGENT_CLEARANCE=$1 PP_PIPE_EXCLUSIVE_AFTER=$2 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ', round(d['value']), 'plans/s', round(d['ms_per_step'],1), 'ms/step', {k: round(v,2) for k,v in d['run_profile'].items()})"
done
