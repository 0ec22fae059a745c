#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05b; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
for cfg in "sorted X=1" "nosort MV_RASTER_NOSORT=1" "fb0 MV_RASTER_COST_FEEDBACK=0"; do
  set -- $cfg; N=$1; shift
  env "$@" MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/rt_$N.json 2> $OUT/rt_$N.err
  echo "== $N"; grep "decile" $OUT/rt_$N.err | cut -c1-420
done
