#!/bin/bash
# (the MV_GLIST_LAZY_IH switch was built for this measurement and not kept)
# r10j: the Hex pass's two-pixel variant compiled for six waves per SIMD (80 VGPRs, 32 bytes of scratch; five: 95, none), with and without forming the rotated
# inverse directions where a round meets a wall of that orientation (-DMV_GLIST_LAZY_IH=1) instead of keeping them in registers
set -u
TAG=${1:-r10j}
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
for i in 1 2; do
  for V in w5 w6 w6lazy w5lazy; do
    LIB=""; [ $V != w5 ] && LIB=$R/megaverse_amd/_variants/libmv_$V.so
    MV_LIB_PATH=$LIB run hexmemory_${V}_$i $B --scenario HexMemory
    MV_LIB_PATH=$LIB run hexexplore_${V}_$i $B --scenario HexExplore
  done
done
