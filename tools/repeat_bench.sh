# N driver-style runs back to back: value, consistency, where the queries were (pipeline_backlog), launch counts -- to catch run-to-run modes
N=${1:-6}
for i in $(seq 1 $N); do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/repeat_$i.json
  python -c "
import json
d=json.loads(open('gpurun_out/repeat_$i.json').read().strip().splitlines()[-1]); rp=d['run_profile']; b=d['pipeline_backlog']; t=d['pipeline_kernel_timings']
print($i, round(d['value']), d['replay_consistent'], 'done 50/99/100 %%: %.2f %.2f %.2f' % (rp['done_50_s'], rp['done_99_s'], rp['done_100_s']), 'ready %.1f searching %.0f' % (b['ready_mean'], b['searching_mean']),
      'wf launches %d, %.0f ms each; search launches %d, %.0f ms total, longest %.0f' % (t['wavefront_launches'], t['wavefront_ms_total'] / max(1, t['wavefront_launches']), t['search_launches'], t['search_ms_total'], t['search_max_ms']))"
done
