#!/bin/bash
# usage: scripts/gpu_variants.sh "<EXTRA flags variant 1>" "<variant 2>" ...   (run on the GPU box through gpurun)
# rebuilds mv_raster.o / mv_step*.o with each EXTRA and prints the bench kernel times
mkdir -p gpurun_out
for v in "$@"; do
  touch megaverse_amd/csrc/mv_raster.hip megaverse_amd/csrc/mv_step.hip
  make -C megaverse_amd/csrc EXTRA="$v" > gpurun_out/build.log 2>&1 || { echo "build failed for $v"; tail -5 gpurun_out/build.log; continue; }
  echo "== $v"
  timeout 120 python -u bench.py --no-cpu-baseline --steps 600 ${BENCH_ARGS} 2>/dev/null | grep -o "\"value\": [0-9.]*\|avg_launch_ms\": [0-9.]*" | tr '\n' ' '
  echo
done
