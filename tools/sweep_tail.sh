# the pipeline's order-of-work knob against the end of a 20-step run (results are the same whatever it is set to: replay_consistent)
for c in 0 1.2 1.5 2.0 3.0 5.0; do
  echo "urgent clearance $c m"
  PP_PIPE_URGENT_CLEARANCE=$c python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); rp=d['run_profile']
print('  ', round(d['value']), 'plans/s', round(d['ms_per_step'],1), 'ms/step; consistent with the last step:', d['replay_consistent'], {k: round(v,2) for k,v in rp.items() if k != 'last_results'}, 'last result [s, position, expansions, status, clearance]:', rp['last_results'][-1])"
done
