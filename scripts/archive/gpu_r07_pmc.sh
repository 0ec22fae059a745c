#!/bin/bash
# the PMC passes of the headline alone (-> profiles/<tag>_pmc_*.csv, then scripts/make_pmc_traffic.py <tag> 8): after a change of the kernel sources that leaves the kernels' code as it was
set -u
TAG=${1:-r07z}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${TAG}_pmc
mkdir -p $OUT
export TMPDIR=/tmp
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"
pmc() { local C=$1 L=$2
  (cd /tmp; MV_BENCH_CALL_SCHEDULE=8 timeout 300 rocprofv3 --pmc $L --kernel-trace -d $OUT/db_$C -o run -- python $R/bench.py --batch 8 --steps 64 --warmup 16 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/tower_pmc_$C.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_$C/run_results.db --pmc > $OUT/tower_pmc_$C.csv 2>> $OUT/tower_pmc_$C.log; rm -rf $OUT/db_$C) }
pmc FETCH_SIZE FETCH_SIZE
pmc WRITE_SIZE WRITE_SIZE
pmc SQ "$SQ1"
pmc SQ2 "$SQ2"
grep -h "batch_kernel\|ticks_kernel" $OUT/tower_pmc_WRITE_SIZE.csv | cut -c1-200
cd $R; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style.json 2> $OUT/driver_style.err; python -c "
import json; d=json.loads(open('$OUT/driver_style.json').read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), d['roofline']['traffic'])"
