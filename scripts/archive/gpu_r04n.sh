#!/bin/bash
set -u
TAG=${1:-r04n}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0"
run() { local N=$1; shift; env "$@" $B > $OUT/$N.json 2> $OUT/$N.err; echo "$N: $(python -c "import json; d=json.load(open('$OUT/$N.json')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4))" 2>/dev/null)"; }
run base X=1
for V in prio1 prio3 t64 t64prio3; do run $V MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_$V.so; done
run base_pad6 MV_RASTER_LDS_PAD=7000
run prio3_waves6 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_prio3.so MV_FAST_WAVES=6
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0 --gpus 1 --steps 20 --warmup 5"
run driver_base X=1
for V in prio3 t64prio3; do run driver_$V MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_$V.so; done
