#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r07d; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log; tail -12 $OUT/smoke.log | cut -c1-250
# the tile-loop census of the committed instrumented build
MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py --steps 160 --warmup 400 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/census.json 2> $OUT/census.err
grep "census" $OUT/census.err | cut -c1-500
