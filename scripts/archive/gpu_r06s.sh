#!/bin/bash
# where the one-launch observation kernel's cycles go: SQ busy / active / wait counters, unpipelined (the kernel alone on the chip) and pipelined
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06s; mkdir -p $OUT; export TMPDIR=/tmp
pmc() { local C=$1 L=$2 E=$3
  (cd /tmp; env $E MV_BENCH_CALL_SCHEDULE=8 timeout 300 rocprofv3 --pmc $L --kernel-trace -d $OUT/db_$C -o run -- python $R/bench.py --batch 8 --steps 64 --warmup 16 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/pmc_$C.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_$C/run_results.db --pmc > $OUT/pmc_$C.csv 2>> $OUT/pmc_$C.log; rm -rf $OUT/db_$C; grep -i "batch_kernel\|ticks_kernel\|^kernel" $OUT/pmc_$C.csv | cut -c1-400) }
pmc A "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" MV_PIPELINE=0
pmc B "SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM" MV_PIPELINE=0
pmc C "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC" MV_PIPELINE=0
pmc D "VALUBusy SALUBusy" MV_PIPELINE=0
pmc E "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_IFETCH SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" MV_PIPELINE=0
