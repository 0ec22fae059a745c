#!/bin/bash
# r08c: where do the observation pass's vector instructions go?  SQ_INSTS_VALU of measurement builds that leave parts of the pass out
# (skip1: prologue + classification only; skip2: general tiles not drawn; skip3: covered tiles not drawn), and the census of the instrumented build
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08c; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_pipelining_gpu.py -m gpu -q -x -k "late_consumer" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for V in "" skip1 skip2 skip3; do
  L=""; [ -n "$V" ] && L="MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_$V.so"
  (cd /tmp; env $L MV_BENCH_CALL_SCHEDULE=8 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $OUT/db_$V -o run -- python $R/bench.py --batch 8 --steps 64 --warmup 16 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/pmc_$V.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_$V/run_results.db --pmc > $OUT/pmc_$V.csv 2>> $OUT/pmc_$V.log; rm -rf $OUT/db_$V)
  echo "== variant '$V'"; grep -h "raster_fast_batch" $OUT/pmc_$V.csv | cut -c1-200
done
MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py --steps 200 --warmup 48 --no-cpu-baseline --profile-steps 0 --no-extra-legs 2>&1 >/dev/null | grep "raster census\|raster timing (last" > $OUT/raster_census.txt
cat $OUT/raster_census.txt | cut -c1-600
