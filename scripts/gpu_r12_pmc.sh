#!/bin/bash
# the PMC passes of scripts/gpu_r12_final.sh alone (-> profiles/r12z_pmc_*.csv, profiles/pmc_traffic.json): the kernel sources' comments changed after r12z, and the hash that guards the counters' read-back covers the text
set -u
TAG=${1:-r12zp}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
python -c "import bench; print('kernel sources', bench.kernel_sources_sha16())" | tee $OUT/kernel_sources_sha16.txt
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"
pmc() { local C=$1 L=$2
  (cd /tmp; MV_BENCH_CALL_SCHEDULE=16 timeout 300 rocprofv3 --pmc $L --kernel-trace -d $OUT/db_$C -o run -- python $R/bench.py --batch 16 --steps 128 --warmup 32 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/tower_pmc_$C.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_$C/run_results.db --pmc > $OUT/tower_pmc_$C.csv 2>> $OUT/tower_pmc_$C.log; rm -rf $OUT/db_$C) }
pmc FETCH_SIZE FETCH_SIZE
pmc WRITE_SIZE WRITE_SIZE
pmc SQ "$SQ1"
pmc SQ2 "$SQ2"
pmc SQ3 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
find $OUT -name "*.db" -delete
ls -la $OUT/*.csv
