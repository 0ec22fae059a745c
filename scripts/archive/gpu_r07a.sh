#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r07a; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_pipelining_gpu.py -m gpu -q -x -k "overlapped or one_launch" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log | cut -c1-300
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --steps 3200 --warmup 400 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "%.2f M"%(d["value"]/1e6), d["config"].get("ring_slots"), d["config"].get("overlapped_passes"))
except Exception as e:
    print(sys.argv[2], "failed", e, open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
}
run tw1024_ov X=1 -- --envs-per-gpu 1024
run tw1024_no X=1 -- --envs-per-gpu 1024 --no-pass-overlap
run tw512_ov X=1 -- --envs-per-gpu 512
run tw512_no X=1 -- --envs-per-gpu 512 --no-pass-overlap
run x4_ov X=1 -- --envs-per-gpu 512 --agents 4
run x4_no X=1 -- --envs-per-gpu 512 --agents 4 --no-pass-overlap
run oh512_ov X=1 -- --scenario ObstaclesHard --envs-per-gpu 512
run oh512_no X=1 -- --scenario ObstaclesHard --envs-per-gpu 512 --no-pass-overlap
