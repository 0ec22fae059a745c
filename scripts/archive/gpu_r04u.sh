#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04u; mkdir -p $OUT; cd $R
MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/rt.json 2> $OUT/rt.err
grep "raster timing" $OUT/rt.err
