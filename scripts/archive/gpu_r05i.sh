#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05i; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0"
run() { local N=$1; shift; echo "$N: $(env "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4))")"; }
for sc in Sokoban "ObstaclesHard --envs-per-gpu 512" Rearrange "ObstaclesEasy"; do
  run "$sc default" $B --scenario $sc
  run "$sc planar0" MV_PLANAR=0 $B --scenario $sc
  run "$sc planar0 notail" MV_PLANAR=0 MV_RASTER_TAIL_DIV=0 $B --scenario $sc
  run "$sc planar1 notail" MV_RASTER_TAIL_DIV=0 $B --scenario $sc
done
