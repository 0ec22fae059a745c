#!/bin/bash
# quick SQ-counter pass over bench.py (60 steps) -> gpurun_out/pmc_SQ
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rm -rf $OUT/pmc_SQ $OUT/pmc_SQ2
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --kernel-trace -d $OUT/pmc_SQ -o run -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 "$@" > $OUT/pmc_SQ.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace -d $OUT/pmc_SQ2 -o run -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 "$@" > $OUT/pmc_SQ2.log 2>&1
