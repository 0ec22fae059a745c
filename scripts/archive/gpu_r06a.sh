#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06a; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
for cfg in "oh512 --scenario ObstaclesHard --envs-per-gpu 512" "oh1024 --scenario ObstaclesHard" "tw512 --envs-per-gpu 512"; do
  set -- $cfg; N=$1; shift
  MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py "$@" --steps 200 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/rt_$N.json 2> $OUT/rt_$N.err
  echo "== $N"; grep -v amdgpu.ids $OUT/rt_$N.err | cut -c1-600 | tail -12
done
