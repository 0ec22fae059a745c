#!/bin/bash
# the round's last pass: the whole GPU suite, the PMC passes of the headline on the final kernel sources (-> profiles/r04z_*, profiles/pmc_traffic.json), the bench lines
set -u
TAG=${1:-r04z}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"
pmc() { local C=$1 L=$2
  (cd /tmp; MV_BENCH_CALL_SCHEDULE=8 timeout 300 rocprofv3 --pmc $L --kernel-trace -d $OUT/db_$C -o run -- python $R/bench.py --batch 8 --steps 64 --warmup 16 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/tower_pmc_$C.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_$C/run_results.db --pmc > $OUT/tower_pmc_$C.csv 2>> $OUT/tower_pmc_$C.log; rm -rf $OUT/db_$C) }
pmc FETCH_SIZE FETCH_SIZE
pmc WRITE_SIZE WRITE_SIZE
pmc SQ "$SQ1"
pmc SQ2 "$SQ2"
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_s -o run -- python $R/bench.py --steps 400 --warmup 48 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/tower_stats.log 2>&1
 python $R/scripts/rocpd_summary.py $OUT/db_s/run_results.db > $OUT/tower_kernel_stats.csv 2>> $OUT/tower_stats.log; rm -rf $OUT/db_s)
(cd /tmp; MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_u -o run -- python $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/tower_unpipelined_stats.log 2>&1
 python $R/scripts/rocpd_summary.py $OUT/db_u/run_results.db > $OUT/tower_unpipelined_kernel_stats.csv 2>> $OUT/tower_unpipelined_stats.log; rm -rf $OUT/db_u)
cd $R
timeout 900 python bench.py > $OUT/tower_bench.json 2> $OUT/tower_bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/tower_bench_driver_style.json 2> $OUT/tower_bench_driver_style.err
find $OUT -name "*.db" -delete
grep -h "raster_fast\|step" $OUT/tower_kernel_stats.csv $OUT/tower_unpipelined_kernel_stats.csv | cut -c1-160
for f in $OUT/tower_bench.json $OUT/tower_bench_driver_style.json; do python -c "import json; d=json.load(open('$f')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')}, d['roofline']['traffic'], d['roofline']['frac'])"; done
