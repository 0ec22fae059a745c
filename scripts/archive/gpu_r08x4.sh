#!/bin/bash
# r08x4: ObstaclesHard 512 (overlapped passes, 8 ticks per call) three times each: the settled scheme / without upload passes between read-backs / with the old 32-call bound
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08x4; mkdir -p $OUT; cd $R
B="timeout 300 python bench.py --no-cpu-baseline --profile-steps 0 --no-extra-legs"
for i in 1 2 3; do
  $B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/a_oh512_${i}_bench.json 2> /dev/null
  MV_X_NO_STALE_PASS=1 $B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/b_oh512_${i}_bench.json 2> /dev/null
  MV_X_NO_STALE_PASS=1 MV_X_BOUND_CALLS=32 $B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/c_oh512_${i}_bench.json 2> /dev/null
  MV_X_BOUND_CALLS=8 $B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/d_oh512_${i}_bench.json 2> /dev/null
done
for i in 1 2; do
  $B --scenario Empty > $OUT/a_Empty_${i}_bench.json 2> /dev/null
  MV_X_NO_STALE_PASS=1 MV_X_BOUND_CALLS=32 $B --scenario Empty > $OUT/c_Empty_${i}_bench.json 2> /dev/null
  MV_X_BOUND_CALLS=8 $B --scenario Empty > $OUT/d_Empty_${i}_bench.json 2> /dev/null
done
for f in $OUT/*_bench.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2))
"; done
