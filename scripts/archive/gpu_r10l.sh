#!/bin/bash
# r10l: Mixed 64 x 64: does the union step launch overlap the union observation launch at all?  Pipelined against one stream (MV_PIPELINE=0); the long-list
# workgroups with two waves instead of four (MV_UNION_TICKS_WAVES=2); the union step kernel at 128 VGPRs (-DMV_UNION_TICKS_WAVES_PER_SIMD=4: 1 KB of scratch)
set -u
TAG=${1:-r10l}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128 --scenario Mixed --obs 64 64"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
for i in 1 2 3; do
  run pipelined_$i $B
  MV_PIPELINE=0 run one_stream_$i $B
  MV_UNION_TICKS_WAVES=2 run two_waves_$i $B
  MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_union128.so run vgpr128_$i $B
  MV_UNION_TICKS_WAVES=2 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_union128.so run vgpr128_two_waves_$i $B
done
