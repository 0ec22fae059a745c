#!/bin/bash
# r10s: the two-wave pipelined step kernels (MV_STEP_PIPE=1) by env count: TowerBuilding, ObstaclesHard, ObstaclesEasy, Empty at 256 / 512 / 768 envs (r10r: ObstaclesHard
# 512 20.8 -> 23.1 M obs/s, 1024 28.6 -> 26.1)
set -u
TAG=${1:-r10s}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
for i in 1 2; do
  for S in TowerBuilding ObstaclesHard ObstaclesEasy Empty; do
    for E in 256 512 768; do
      for P in 0 1; do
        MV_STEP_PIPE=$P run ${S}_${E}_pipe${P}_$i $B --scenario $S --envs-per-gpu $E
      done
    done
  done
done
