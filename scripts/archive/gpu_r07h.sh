#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r07h; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
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
for p in 1 2; do
  run hexmem_b1_p$p MV_FAST_PPL=$p -- --scenario HexMemory --batch 1
  run collect_b1_p$p MV_FAST_PPL=$p -- --scenario Collect --batch 1
  run hexexp_p$p MV_FAST_PPL=$p -- --scenario HexExplore
  run hexmem_a2_p$p MV_FAST_PPL=$p -- --scenario HexMemory --agents 2 --envs-per-gpu 512
  run collect72_p$p MV_FAST_PPL=$p -- --scenario Collect --obs 128 72
done
