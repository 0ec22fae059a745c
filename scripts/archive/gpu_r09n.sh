#!/bin/bash
# r09n: Collect's pass with slab tests specialised by ray sign (box_test_g_signed) against the general test (-DMV_GLIST_SIGNED=0)
set -u
TAG=${1:-r09n}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1800 python -m pytest tests/test_collect_parity_gpu.py tests/test_fast_pixels_gpu.py tests/test_multitask_gpu.py -m gpu -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
for i in 1 2; do
  run collect_signed_$i $B --scenario Collect
  MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_nosigned.so run collect_general_$i $B --scenario Collect
  run collect72_signed_$i $B --scenario Collect --obs 128 72
  MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_nosigned.so run collect72_general_$i $B --scenario Collect --obs 128 72
done
MV_PIPELINE=0 run collect_alone_signed $B --scenario Collect --steps 400
MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_nosigned.so run collect_alone_general $B --scenario Collect --steps 400
