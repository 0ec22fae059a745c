#!/bin/bash
# r10h: census of the long-list pass's tests (instrumented build, -DMV_RASTER_TIMING): slab tests and other primitives (cones, capsules) per tile, hit pixels
set -u
TAG=${1:-r10h}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
for cfg in "hexmem --scenario HexMemory" "hexexplore --scenario HexExplore" "collect --scenario Collect"; do
  set -- $cfg; N=$1; shift
  MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py "$@" --envs-per-gpu 256 --steps 80 --warmup 240 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/c_$N.json 2> $OUT/c_$N.err
  echo "== $N"; grep "census" $OUT/c_$N.err | cut -c1-600
done > $OUT/long_list_census.txt
cat $OUT/long_list_census.txt
