#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06j; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --steps 800 --warmup 100 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "%.2f M"%(d["value"]/1e6))
PY
}
run x4_b1_s2 MV_RASTER_SPLIT=2 -- --agents 4 --envs-per-gpu 512 --batch 1
run x4_b1_s1 MV_RASTER_SPLIT=1 -- --agents 4 --envs-per-gpu 512 --batch 1
run x4_b1_s1_t0 MV_RASTER_SPLIT=1 MV_RASTER_TAIL_DIV=0 -- --agents 4 --envs-per-gpu 512 --batch 1
run e4096_b1_s2 MV_RASTER_SPLIT=2 -- --envs-per-gpu 4096 --batch 1
run e4096_b1_s1 MV_RASTER_SPLIT=1 -- --envs-per-gpu 4096 --batch 1
run e4096 X=1 -- --envs-per-gpu 4096
run e2048 X=1 -- --envs-per-gpu 2048
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
