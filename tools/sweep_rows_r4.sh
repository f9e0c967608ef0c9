#!/bin/bash
# round 4: rows of the search grid x tile width / waves per launch of the tile wavefront (driver-style 20 steps), one line per run
# usage: tools/sweep_rows_r4.sh "<rows list>" "<tile widths>" "<tile grids, 0 = resident>"
out=gpurun_out/r4_sweep_rows.txt
for rows in $1; do
  for tw in $2; do
    for grid in $3; do
      export PP_WF_TILE_WIDTH=$tw
      if [ "$grid" = "0" ]; then unset PP_WF_TILES_GRID; else export PP_WF_TILES_GRID=$grid; fi
      timeout -k 10 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipe-rows $rows --check-poses 1048576 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
b=d.get('pipeline_backlog',{})
print('rows $rows tilewidth $tw tilegrid $grid value %.0f ms/step %.1f ready_mean %.0f searching %.0f last %.2fs consistent %s' % (d['value'], d['ms_per_step'], b.get('ready_mean',-1), b.get('searching_mean',-1), d['run_profile']['done_100_s'], d.get('replay_consistent')))" >> $out 2>&1
      tail -1 $out
    done
  done
done
