#!/bin/bash
# r10k: Mixed 64 x 64 (configs[4]): how much of the union observation launch is its long-list frames' work -- the skip builds of raster_glist_body (r10b) under
# the union batch kernel: kernel time and SQ counters
set -u
TAG=${1:-r10k}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for V in full gskip1 gskip2 gskip3; do
  LIB=""; [ $V != full ] && LIB=$R/megaverse_amd/_variants/libmv_$V.so
  (cd /tmp; MV_LIB_PATH=$LIB timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/db_$V -o run -- python $R/bench.py --scenario Mixed --obs 64 64 --steps 64 --warmup 16 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/pmc_Mixed64_$V.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_$V/run_results.db --pmc > $OUT/pmc_Mixed64_$V.csv 2>> $OUT/pmc_Mixed64_$V.log; rm -rf $OUT/db_$V)
  echo "== Mixed64 $V"; grep -h "raster_union_batch\|step_union_ticks" $OUT/pmc_Mixed64_$V.csv | cut -c1-40,95-200
done
