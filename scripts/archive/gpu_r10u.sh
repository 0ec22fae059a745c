#!/bin/bash
# r10u: TowerBuilding's resident step kernel at 96 / 80 VGPRs (-DMV_STEP_TICKS_WAVES_PER_SIMD=5 / 6: 244 / 320 bytes of scratch; the product: 128 VGPRs, 52 bytes)
# -- fewer registers for the wave that stays resident beside the passes, the headline configuration
set -u
TAG=${1:-r10u}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
for i in 1 2 3; do
  run tower_128vgpr_$i $B
  MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_step5.so run tower_96vgpr_$i $B
  MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_step6.so run tower_80vgpr_$i $B
done
