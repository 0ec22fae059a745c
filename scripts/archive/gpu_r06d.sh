#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06d; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --steps 800 --warmup 100 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "%.2f M"%(d["value"]/1e6))
PY
}
for s in 0 1; do
run tw1024_s$s MV_RASTER_SPLIT=$s -- --envs-per-gpu 1024
run tw512_s$s MV_RASTER_SPLIT=$s -- --envs-per-gpu 512
run oh512_s$s MV_RASTER_SPLIT=$s -- --scenario ObstaclesHard --envs-per-gpu 512
run oh1024_s$s MV_RASTER_SPLIT=$s -- --scenario ObstaclesHard
run tw512x4_s$s MV_RASTER_SPLIT=$s -- --envs-per-gpu 512 --agents 4
done
