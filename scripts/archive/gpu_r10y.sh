#!/bin/bash
# (GymView::lpt_hint was built for this measurement and not kept: no difference beyond the run-to-run spread; the step launch alone 16.8 us per tick either way)
# r10y: the frame setup's cost-bin atomic issued FIRST, under the bin the frame's previous setup found (GymView::lpt_hint), instead of last with the cost just
# found -- the round trip to L2 leaves the step wave's critical path -- against the previous library (base): tests, then the configurations a step launch bounds
set -u
TAG=${1:-r10y}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 2400 python -m pytest tests/test_pipelining_gpu.py tests/test_fast_pixels_gpu.py tests/test_full_size_oracle_gpu.py tests/test_soak_gpu.py tests/test_multitask_gpu.py tests/test_refill_protocol_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3), {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')})
except Exception as e: print('$name', 'failed', e)
PY
}
for i in 1 2; do
  for V in new base; do
    LIB=""; [ $V = base ] && LIB=$R/megaverse_amd/_variants/libmv_base.so
    MV_LIB_PATH=$LIB run tower_${V}_$i $B
    MV_LIB_PATH=$LIB run tower512_${V}_$i $B --envs-per-gpu 512
    MV_LIB_PATH=$LIB run oh512_${V}_$i $B --scenario ObstaclesHard --envs-per-gpu 512
    MV_LIB_PATH=$LIB run empty_${V}_$i $B --scenario Empty
    MV_LIB_PATH=$LIB run sokoban_${V}_$i $B --scenario Sokoban
    MV_LIB_PATH=$LIB run driver_${V}_$i timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 16
    MV_LIB_PATH=$LIB run alone_${V}_$i env MV_PIPELINE=0 $B --steps 400
  done
done
