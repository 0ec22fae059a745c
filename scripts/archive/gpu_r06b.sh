#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06b; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --steps 400 --warmup 100 --no-cpu-baseline --profile-steps 100 --no-extra-legs > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d.get("roofline",{})
print(sys.argv[2], "%.2f M"%(d["value"]/1e6), "raster us", r.get("kernel_us"), "step us", (d.get("kernels") or {}).get("step_us"))
PY
}
for s in 0 2 4 8; do
  run oh512_s$s MV_RASTER_SPLIT=$s -- --scenario ObstaclesHard --envs-per-gpu 512
  run tw512_s$s MV_RASTER_SPLIT=$s -- --envs-per-gpu 512
done
run oh512_s2_b1 MV_RASTER_SPLIT=2 MV_STEP_TICKS=1 -- --scenario ObstaclesHard --envs-per-gpu 512
