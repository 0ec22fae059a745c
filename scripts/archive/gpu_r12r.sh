#!/bin/bash
# r12r: the feeder's thread rule leaves one core of the share to the caller's thread (r12p / r12q): the defaults on two cores (taskset -c 0,1) and on all, no overrides
set -u
TAG=${1:-r12r}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
for i in 1 2; do
for S in "ObstaclesHard 512 128" "ObstaclesHard 1024 128" "Collect 1024 128" "Mixed 1024 64" "Mixed4 1024 64" "HexExplore 1024 128"; do set -- $S
  taskset -c 0,1 $B --scenario $1 --envs-per-gpu $2 --obs $3 $3 > $OUT/$1_$2_2cores_$i.json 2> /dev/null
  $B --scenario $1 --envs-per-gpu $2 --obs $3 $3 > $OUT/$1_$2_all_$i.json 2> /dev/null
done; done
for f in $OUT/*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M', d['config'].get('host_generator_threads'))
except Exception as e: print('$f', 'failed', e)
"; done
