#!/bin/bash
# r08s: second sample of r08r's two states (normal / high-priority simulation stream), twice each, on the configurations where they differed; Empty with 8 ticks per call
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08s; mkdir -p $OUT; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0"
run() { local tag=$1; shift
  $B --scenario Mixed --obs 64 64 > $OUT/${tag}_mixed_64_bench.json 2> /dev/null
  $B --scenario Mixed4 --obs 64 64 > $OUT/${tag}_mixed4_64_bench.json 2> /dev/null
  $B --scenario Sokoban > $OUT/${tag}_Sokoban_bench.json 2> /dev/null
  $B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/${tag}_obstacles_hard_512_bench.json 2> /dev/null
  $B --scenario ObstaclesHard > $OUT/${tag}_obstacles_hard_1024_bench.json 2> /dev/null
  $B --scenario Empty --batch 8 > $OUT/${tag}_Empty_b8_bench.json 2> /dev/null
  $B --scenario Collect > $OUT/${tag}_Collect_bench.json 2> /dev/null
  $B --envs-per-gpu 512 --agents 4 > $OUT/${tag}_tower_512x4_bench.json 2> /dev/null
  $B --envs-per-gpu 4096 > $OUT/${tag}_tower_4096_bench.json 2> /dev/null
  $B > $OUT/${tag}_tower_bench.json 2> /dev/null
  $B --steps 20 --warmup 5 > $OUT/${tag}_tower_driver_style_bench.json 2> /dev/null
}
run n1
MV_X_SIM_PRIORITY=high run h1
run n2
MV_X_SIM_PRIORITY=high run h2
for f in $OUT/*_bench.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M', d['steps'], d['config'].get('ticks_per_call'))
except Exception as e: print('$f', 'failed', e)
"; done
