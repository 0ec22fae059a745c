#!/bin/bash
# r12o: the Obstacles family and the Hex scenarios on two host cores (taskset -c 0,1) against all cores, at their benchmarked sizes: is anything besides Collect bound by the host at eight ranks?
set -u
TAG=${1:-r12o}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
for i in 1 2; do
for S in "ObstaclesHard 512" "ObstaclesHard 1024" "ObstaclesEasy 1024" "HexMemory 1024" "HexExplore 1024" "Rearrange 1024" "Sokoban 1024"; do set -- $S
  taskset -c 0,1 $B --scenario $1 --envs-per-gpu $2 > $OUT/$1_$2_2cores_$i.json 2> /dev/null
  $B --scenario $1 --envs-per-gpu $2 > $OUT/$1_$2_all_$i.json 2> /dev/null
done; done
for f in $OUT/*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M')
except Exception as e: print('$f', 'failed', e)
"; done
