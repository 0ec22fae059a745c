#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06c; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --steps 800 --warmup 100 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "%.2f M"%(d["value"]/1e6))
PY
}
run oh512_def X=1 -- --scenario ObstaclesHard --envs-per-gpu 512
run oh512_mt MV_STEP_TICKS_OBST_MIN_ENVS=1 -- --scenario ObstaclesHard --envs-per-gpu 512
run oh512_mt_pp MV_STEP_TICKS_OBST_MIN_ENVS=1 MV_RASTER_BATCH_SPLIT_PER_PASS=1 -- --scenario ObstaclesHard --envs-per-gpu 512
run oh256_def X=1 -- --scenario ObstaclesHard --envs-per-gpu 256
run oh256_mt MV_STEP_TICKS_OBST_MIN_ENVS=1 -- --scenario ObstaclesHard --envs-per-gpu 256
run tw512 X=1 -- --envs-per-gpu 512
run tw512_pp MV_RASTER_BATCH_SPLIT_PER_PASS=1 -- --envs-per-gpu 512
run tw256 X=1 -- --envs-per-gpu 256
run tw256_pp MV_RASTER_BATCH_SPLIT_PER_PASS=1 -- --envs-per-gpu 256
run tw128 X=1 -- --envs-per-gpu 128
run tw128_pp MV_RASTER_BATCH_SPLIT_PER_PASS=1 -- --envs-per-gpu 128
run tw128x4 X=1 -- --envs-per-gpu 128 --agents 4
run tw128x4_pp MV_RASTER_BATCH_SPLIT_PER_PASS=1 -- --envs-per-gpu 128 --agents 4
run tw1024 X=1 -- --envs-per-gpu 1024
