#!/bin/bash
# two rules of bench.py / the library checked on the final code: the batched group call at 2048 envs against tick-by-tick union launches; overlapped passes on / off for Sokoban and ObstaclesHard 1024
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08z_rules; mkdir -p $OUT; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0"
$B --scenario Mixed --obs 64 64 --envs-per-gpu 2048 > $OUT/mixed_2048_batched_bench.json 2> /dev/null
MV_STEP_TICKS=0 $B --scenario Mixed --obs 64 64 --envs-per-gpu 2048 > $OUT/mixed_2048_tick_by_tick_bench.json 2> /dev/null
MV_STEP_TICKS=0 $B --scenario Mixed --obs 64 64 > $OUT/mixed_1024_tick_by_tick_bench.json 2> /dev/null
$B --scenario Mixed --obs 64 64 --envs-per-gpu 512 > $OUT/mixed_512_batched_bench.json 2> /dev/null
MV_STEP_TICKS=0 $B --scenario Mixed --obs 64 64 --envs-per-gpu 512 > $OUT/mixed_512_tick_by_tick_bench.json 2> /dev/null
for s in Sokoban ObstaclesHard ObstaclesEasy; do for o in on off; do
  $B --scenario $s --pass-overlap $o > $OUT/${s}_overlap_${o}_bench.json 2> /dev/null
done; done
for f in $OUT/*_bench.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), d['config'].get('ticks_per_call'))
"; done
