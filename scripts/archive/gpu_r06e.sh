#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06e; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --steps 800 --warmup 100 --no-cpu-baseline --profile-steps 100 --no-extra-legs > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "%.2f M"%(d["value"]/1e6), "raster/tick %.1f us"%(d["roofline"]["avg_launch_ms"]*1e3), "step/tick %.1f us"%(d["roofline_physics"]["avg_launch_ms"]*1e3))
PY
}
run tw1024 X=1 -- --envs-per-gpu 1024
run tw512 X=1 -- --envs-per-gpu 512
run oh512 X=1 -- --scenario ObstaclesHard --envs-per-gpu 512
run tw256 X=1 -- --envs-per-gpu 256
run tw128x4 X=1 -- --envs-per-gpu 128 --agents 4
# single-pass launches: the closed-loop shape (one tick per call), pipelined and alone
for cfg in "s2t8 MV_RASTER_SPLIT=2" "s1t0 MV_RASTER_SPLIT=1 MV_RASTER_TAIL_DIV=0" "s1t8x4 MV_RASTER_SPLIT=1 MV_RASTER_TAIL_SPLIT=4" "s1t8x8 MV_RASTER_SPLIT=1" "s1t4x4 MV_RASTER_SPLIT=1 MV_RASTER_TAIL_DIV=4 MV_RASTER_TAIL_SPLIT=4" "s1t4x2 MV_RASTER_SPLIT=1 MV_RASTER_TAIL_DIV=4 MV_RASTER_TAIL_SPLIT=2"; do
  set -- $cfg; N=$1; shift
  run b1_$N "$@" -- --batch 1
  run b1u_$N "$@" MV_PIPELINE=0 -- --batch 1
done
