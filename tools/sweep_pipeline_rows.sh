for rows in 1024 3072 4096 6144 8192; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --check-poses 1048576 --pipe-rows $rows > gpurun_out/sw_rows_$rows.json 2> gpurun_out/sw_rows_$rows.err || echo FAILED $rows
  python -c "
import json,sys
d=json.load(open('gpurun_out/sw_rows_$rows.json'))
t=d['pipeline_kernel_timings']
print('rows',$rows,'plans/s %.0f'%d['value'],'ms/step %.1f'%d['ms_per_step'],'wf busy ms %.0f'%t['wavefront_ms_total'],'launches',t['wavefront_launches'])
" >> gpurun_out/sweep1.txt
done
cat gpurun_out/sweep1.txt
