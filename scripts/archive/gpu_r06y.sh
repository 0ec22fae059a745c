#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06y; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --steps 800 --warmup 200 --no-cpu-baseline --profile-steps 100 --no-extra-legs > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "%.2f M"%(d["value"]/1e6), "raster/tick %.1f us"%(d["roofline"]["avg_launch_ms"]*1e3), "step/tick %.1f us"%(d["roofline_physics"]["avg_launch_ms"]*1e3))
PY
}
# raster workgroups per CU capped by LDS padding: 17 KB each without padding (7 fit by VGPRs); pad p -> floor(160 / (17 + p)) workgroups
for pad in 0 6000 10000 15000 23000 36000; do
  run u_pad$pad MV_RASTER_LDS_PAD=$pad MV_PIPELINE=0 -- --envs-per-gpu 1024
done
for pad in 0 6000 15000; do
  run p_pad$pad MV_RASTER_LDS_PAD=$pad -- --envs-per-gpu 1024
done
