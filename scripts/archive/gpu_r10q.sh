#!/bin/bash
# r10q: 512 envs per GPU (configs[2]'s share of ObstaclesHard; TowerBuilding 512; 512 x 4 agents): 8 ticks per call (the rule below 1024 frames) against 16
set -u
TAG=${1:-r10q}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3), d['config'].get('ticks_per_call'), d['config'].get('overlapped_passes'))
except Exception as e: print('$name', 'failed', e)
PY
}
for i in 1 2; do
  for K in 8 16; do
    run oh512_k${K}_$i $B --scenario ObstaclesHard --envs-per-gpu 512 --batch $K
    run oh512_k${K}_no_overlap_$i $B --scenario ObstaclesHard --envs-per-gpu 512 --batch $K --pass-overlap off
    run tower512_k${K}_$i $B --envs-per-gpu 512 --batch $K
    run tower256_k${K}_$i $B --envs-per-gpu 256 --batch $K
    run oe512_k${K}_$i $B --scenario ObstaclesEasy --envs-per-gpu 512 --batch $K
  done
done
