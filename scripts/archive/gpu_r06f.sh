#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06f; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --warmup 100 --no-cpu-baseline --no-extra-legs > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "%.2f M"%(d["value"]/1e6), d["steps"])
PY
}
for i in 1 2 3; do
run tw512_p0_$i X=1 -- --envs-per-gpu 512 --steps 800 --profile-steps 0
run tw512_p100_$i X=1 -- --envs-per-gpu 512 --steps 800 --profile-steps 100
run tw512_long_$i X=1 -- --envs-per-gpu 512 --steps 4000 --profile-steps 0
done
