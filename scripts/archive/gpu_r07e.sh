#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r07e; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 900 python -m pytest tests/test_fast_pixels_gpu.py -m gpu -q -x -k "depth_classes" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log | cut -c1-400
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --steps 1600 --warmup 200 --no-cpu-baseline --profile-steps 100 --no-extra-legs > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "%.2f M"%(d["value"]/1e6), "raster/tick %.1f us"%(d["roofline"]["avg_launch_ms"]*1e3), "step/tick %.1f us"%(d["roofline_physics"]["avg_launch_ms"]*1e3))
except Exception as e:
    print(sys.argv[2], "failed", e, open(sys.argv[1].replace('.json','.err')).read()[-500:])
PY
}
for v in 1 0; do
  run HexMemory_s$v MV_DEPTH_SORT=$v -- --scenario HexMemory
  run HexExplore_s$v MV_DEPTH_SORT=$v -- --scenario HexExplore
  run Collect_s$v MV_DEPTH_SORT=$v -- --scenario Collect
  run Mixed64_s$v MV_DEPTH_SORT=$v -- --scenario Mixed --obs 64 64
done
