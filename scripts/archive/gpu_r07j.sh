#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r07j; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --steps 2400 --warmup 200 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "%.2f M"%(d["value"]/1e6))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
for s in Sokoban Collect HexMemory ObstaclesEasy; do
  run ${s}_on X=1 -- --scenario $s --pass-overlap on
  run ${s}_off X=1 -- --scenario $s --pass-overlap off
done
run oh256_on X=1 -- --scenario ObstaclesHard --envs-per-gpu 256 --pass-overlap on
run oh256_off X=1 -- --scenario ObstaclesHard --envs-per-gpu 256 --pass-overlap off
