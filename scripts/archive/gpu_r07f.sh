#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r07f; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
for cfg in "hexmem --scenario HexMemory" "collect --scenario Collect"; do
  set -- $cfg; N=$1; shift
  for srt in 1 0; do
  MV_DEPTH_SORT=$srt MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py "$@" --envs-per-gpu 256 --steps 80 --warmup 240 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/c_$N$srt.json 2> $OUT/c_$N$srt.err
  echo "== $N sort=$srt"; grep "census" $OUT/c_$N$srt.err | cut -c1-500
  done
done
