#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r07b; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --steps 4000 --warmup 400 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "%.2f M"%(d["value"]/1e6), d["config"].get("ring_slots"), d["config"].get("overlapped_passes"))
PY
}
for rep in a b; do
run oh512_ov_$rep X=1 -- --scenario ObstaclesHard --envs-per-gpu 512 --pass-overlap
run oh512_no_$rep X=1 -- --scenario ObstaclesHard --envs-per-gpu 512
run oh1024_ov_$rep X=1 -- --scenario ObstaclesHard --pass-overlap
run oh1024_no_$rep X=1 -- --scenario ObstaclesHard
run re_ov_$rep X=1 -- --scenario Rearrange --pass-overlap
run re_no_$rep X=1 -- --scenario Rearrange
done
