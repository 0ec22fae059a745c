#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06v; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --steps 800 --warmup 100 --no-cpu-baseline --profile-steps 100 --no-extra-legs > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "%.2f M"%(d["value"]/1e6), "raster/tick %.1f us"%(d["roofline"]["avg_launch_ms"]*1e3), "step/tick %.1f us"%(d["roofline_physics"]["avg_launch_ms"]*1e3))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for v in 4 3 2; do
  L=X=1; [ $v != 4 ] && L=MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_wps$v.so
  run oh512_w$v $L -- --scenario ObstaclesHard --envs-per-gpu 512
  run oh1024_w$v $L -- --scenario ObstaclesHard
  for s in Rearrange Sokoban Collect HexMemory; do run ${s}_w$v $L -- --scenario $s; done
done
for s in Rearrange Sokoban Collect HexMemory HexExplore; do run ${s}_off MV_STEP_TICKS_OTHERS=0 -- --scenario $s; done
timeout 900 python -m pytest tests/test_pipelining_gpu.py tests/test_step_n_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log | cut -c1-300
