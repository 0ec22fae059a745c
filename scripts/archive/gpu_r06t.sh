#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06t; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
for cfg in "tw1024 --envs-per-gpu 1024" "oh512 --scenario ObstaclesHard --envs-per-gpu 512"; do
  set -- $cfg; N=$1; shift
  MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py "$@" --steps 400 --warmup 800 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/rt_$N.json 2> $OUT/rt_$N.err
  echo "== $N"; grep "census\|cycles per wave" $OUT/rt_$N.err | cut -c1-600
done
