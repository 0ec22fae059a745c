#!/bin/bash
# r04k: raised wave priority for the step kernel's waves, now that the step chain is the bound of the pipelined rate
set -u
TAG=${1:-r04k}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs"
run() { local N=$1; shift; env "$@" $B > $OUT/$N.json 2> $OUT/$N.err; echo "$N: $(python -c "import json; d=json.load(open('$OUT/$N.json')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), 'raster', round(d['roofline']['avg_launch_ms'],4), 'step', round(d['roofline_physics']['avg_launch_ms'],4))" 2>/dev/null)"; }
run base0 MV_RASTER_BATCH=0
run base8 MV_RASTER_BATCH=8
for P in 1 3; do
  run prio${P}_b0 MV_RASTER_BATCH=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_prio$P.so
  run prio${P}_b8 MV_RASTER_BATCH=8 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_prio$P.so
  run prio${P}_b2 MV_RASTER_BATCH=2 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_prio$P.so
done
