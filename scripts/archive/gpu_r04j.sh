#!/bin/bash
# r04j: batch raster beside the step kernels: priorities, occupancy caps, chunk sizes
set -u
TAG=${1:-r04j}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0"
run() { local N=$1; shift; env "$@" $B > $OUT/$N.json 2> $OUT/$N.err; echo "$N: $(python -c "import json; d=json.load(open('$OUT/$N.json')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4))" 2>/dev/null)"; }
run base MV_RASTER_BATCH=0
run batch8 MV_RASTER_BATCH=8
run batch4 MV_RASTER_BATCH=4
run batch2 MV_RASTER_BATCH=2
run batch8_simhigh MV_RASTER_BATCH=8 MV_SIM_PRIORITY=high
run batch0_simhigh MV_RASTER_BATCH=0 MV_SIM_PRIORITY=high
run batch8_pad6 MV_RASTER_BATCH=8 MV_RASTER_LDS_PAD=7000
run batch8_pad5 MV_RASTER_BATCH=8 MV_RASTER_LDS_PAD=11000
run batch8_pad5_simhigh MV_RASTER_BATCH=8 MV_RASTER_LDS_PAD=11000 MV_SIM_PRIORITY=high
run batch0_pad6 MV_RASTER_BATCH=0 MV_RASTER_LDS_PAD=7000
run batch0_pad5 MV_RASTER_BATCH=0 MV_RASTER_LDS_PAD=11000
run batch8_waves6 MV_RASTER_BATCH=8 MV_FAST_WAVES=6
