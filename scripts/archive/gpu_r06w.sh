#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06w; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --steps 4000 --warmup 400 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "%.2f M"%(d["value"]/1e6))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for rep in a b; do
for v in 4 3 2; do
  L=X=1; [ $v != 4 ] && L=MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_wps$v.so
  run oh512_w${v}_$rep $L -- --scenario ObstaclesHard --envs-per-gpu 512
  run oh1024_w${v}_$rep $L -- --scenario ObstaclesHard
  run Sokoban_w${v}_$rep $L -- --scenario Sokoban
  run Collect_w${v}_$rep $L -- --scenario Collect
done
done
