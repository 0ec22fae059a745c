#!/bin/bash
# r12m: ticks between TowerBuilding's draw launches (tower_draw_kernel, 47 us each on the copy stream; r12j: any kernel of another queue in flight costs the observation launches 9 %): 8 (the rule) / 16 / 32 / 64
set -u
TAG=${1:-r12m}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
for i in 1 2 3; do for P in 8 16 32 64; do
  MV_DRAW_PERIOD=$P $B > $OUT/tower_period${P}_$i.json 2> /dev/null
done; done
for P in 8 32; do MV_DRAW_PERIOD=$P $B --envs-per-gpu 4096 > $OUT/tower4096_period${P}.json 2> /dev/null; MV_DRAW_PERIOD=$P $B --gpus 1 --steps 20 --warmup 5 > $OUT/driver_period${P}.json 2> /dev/null; done
for f in $OUT/*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$f', 'failed', e)
"; done
