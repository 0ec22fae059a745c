#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06k; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --steps 1600 --warmup 200 --no-cpu-baseline --profile-steps 100 --no-extra-legs > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "%.2f M"%(d["value"]/1e6), "raster/tick %.1f us"%(d["roofline"]["avg_launch_ms"]*1e3), "step/tick %.1f us"%(d["roofline_physics"]["avg_launch_ms"]*1e3))
PY
}
for v in def prio3; do
  L=X=1; [ $v = prio3 ] && L=MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_prio3.so
  run tw1024_$v $L -- --envs-per-gpu 1024
  run tw512_$v $L -- --envs-per-gpu 512
  run oh512_$v $L -- --scenario ObstaclesHard --envs-per-gpu 512
  run oh1024_$v $L -- --scenario ObstaclesHard
  run x4_$v $L -- --envs-per-gpu 512 --agents 4
done
