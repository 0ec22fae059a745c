#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06q; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_fast_pixels_gpu.py -m gpu -q -x -k "single_face or planar" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log | cut -c1-300
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --steps 1600 --warmup 200 --no-cpu-baseline --profile-steps 100 --no-extra-legs > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "%.2f M"%(d["value"]/1e6), "raster/tick %.1f us"%(d["roofline"]["avg_launch_ms"]*1e3), "step/tick %.1f us"%(d["roofline_physics"]["avg_launch_ms"]*1e3))
PY
}
for v in 1 0; do
  run tw1024_f$v MV_FACES=$v -- --envs-per-gpu 1024
  run tw1024u_f$v MV_FACES=$v MV_PIPELINE=0 -- --envs-per-gpu 1024
  run oh512_f$v MV_FACES=$v -- --scenario ObstaclesHard --envs-per-gpu 512
  run x4_f$v MV_FACES=$v -- --envs-per-gpu 512 --agents 4
  run b1u_f$v MV_FACES=$v MV_PIPELINE=0 -- --batch 1
done
