#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r07l; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --steps 1600 --warmup 200 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "%.2f M"%(d["value"]/1e6))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
run mixed64_def X=1 -- --scenario Mixed --obs 64 64
run mixed64_ppl2 MV_FAST_PPL=2 -- --scenario Mixed --obs 64 64
run mixed64_s2 MV_RASTER_SPLIT=2 -- --scenario Mixed --obs 64 64
run mixed64_s8 MV_RASTER_SPLIT=8 -- --scenario Mixed --obs 64 64
run mixed64_2048 X=1 -- --scenario Mixed --obs 64 64 --envs-per-gpu 2048
