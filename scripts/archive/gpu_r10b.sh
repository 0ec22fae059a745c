#!/bin/bash
# r10b: where do the long-list pass's instructions go?  SQ counters of measurement builds of raster_glist_body that leave parts of it out
# (-DMV_GLIST_DEBUG_SKIP=1: nothing shaded; 2: culled + rays set up, nothing tested; 3: culled only), HexMemory and Collect, batched calls
set -u
TAG=${1:-r10b}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for S in HexMemory Collect; do
  for V in full gskip1 gskip2 gskip3; do
    LIB=""; [ $V != full ] && LIB=$R/megaverse_amd/_variants/libmv_$V.so
    (cd /tmp; MV_LIB_PATH=$LIB timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/db_${S}_$V -o run -- python $R/bench.py --scenario $S --steps 64 --warmup 16 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/pmc_${S}_$V.log 2>&1
     python $R/scripts/rocpd_summary.py $OUT/db_${S}_$V/run_results.db --pmc > $OUT/pmc_${S}_$V.csv 2>> $OUT/pmc_${S}_$V.log; rm -rf $OUT/db_${S}_$V)
    echo "== $S $V"; grep -h "raster_glist_batch" $OUT/pmc_${S}_$V.csv | cut -c1-40,100-200
  done
done
