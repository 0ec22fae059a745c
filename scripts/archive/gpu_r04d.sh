#!/bin/bash
# r04d: where a raster workgroup's time goes (instrumented build)
set -u
TAG=${1:-r04d}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for P in 1 0; do
  MV_PLANAR=$P MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/rt_planar$P.json 2> $OUT/rt_planar$P.err
  grep "raster timing" $OUT/rt_planar$P.err
done
MV_RASTER_SPLIT=4 MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/rt_split4.json 2> $OUT/rt_split4.err
grep "raster timing" $OUT/rt_split4.err
