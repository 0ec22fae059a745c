#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r07g; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
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
for s in 0 1 2 8; do
  run hexmem_s$s MV_RASTER_SPLIT=$s -- --scenario HexMemory
  run collect_s$s MV_RASTER_SPLIT=$s -- --scenario Collect
done
run hexmem_ppl2 MV_FAST_PPL=2 -- --scenario HexMemory
run collect_ppl2 MV_FAST_PPL=2 -- --scenario Collect
run hexmem_ppl2_s2 MV_FAST_PPL=2 MV_RASTER_SPLIT=2 -- --scenario HexMemory
run collect_ppl2_s2 MV_FAST_PPL=2 MV_RASTER_SPLIT=2 -- --scenario Collect
