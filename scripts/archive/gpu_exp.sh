#!/bin/bash
# experiments: gpu_exp.sh <outdir>   (bench variants controlled by env)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
B="python bench.py --steps 600 --warmup 50 --no-cpu-baseline"
for sp in 4 8 16; do
MV_RASTER_SPLIT=$sp $B > $OUT/bench_sort_s${sp}.json 2>&1
MV_RASTER_NOSORT=1 MV_RASTER_SPLIT=$sp $B > $OUT/bench_nosort_s${sp}.json 2>&1
done
MV_RASTER_NOSORT=1 MV_RASTER_SPLIT=8 $B --scenario Collect > $OUT/bench_nosort_collect_s8.json 2>&1
MV_RASTER_SPLIT=8 $B --scenario Collect > $OUT/bench_sort_collect_s8.json 2>&1
MV_RASTER_SPLIT=4 $B --scenario Collect > $OUT/bench_sort_collect_s4.json 2>&1
