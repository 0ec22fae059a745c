#!/bin/bash
# the four PMC passes of the headline + the bench lines that read them back, on the committed kernel sources (-> profiles/r08z_pmc_*, profiles/pmc_traffic.json)
set -u
TAG=${1:-r08z}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${TAG}_pmc
mkdir -p $OUT
export TMPDIR=/tmp
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"
pmc() { local C=$1 L=$2
  (cd /tmp; MV_BENCH_CALL_SCHEDULE=16 timeout 300 rocprofv3 --pmc $L --kernel-trace -d $OUT/db_$C -o run -- python $R/bench.py --batch 16 --steps 128 --warmup 32 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/tower_pmc_$C.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_$C/run_results.db --pmc > $OUT/tower_pmc_$C.csv 2>> $OUT/tower_pmc_$C.log; rm -rf $OUT/db_$C) }
pmc FETCH_SIZE FETCH_SIZE
pmc WRITE_SIZE WRITE_SIZE
pmc SQ "$SQ1"
pmc SQ2 "$SQ2"
cd $R
# (pmc_traffic.json of THIS tree is written from the passes above before the bench lines read it)
for C in FETCH_SIZE WRITE_SIZE SQ SQ2; do cp $OUT/tower_pmc_$C.csv profiles/${TAG}_pmc_$C.csv; done
python scripts/make_pmc_traffic.py $TAG 16 > /dev/null
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
timeout 900 python bench.py > $OUT/tower_bench.json 2> $OUT/tower_bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/tower_bench_driver_style.json 2> /dev/null
for f in $OUT/tower_bench.json $OUT/tower_bench_driver_style.json; do python -c "import json; d=json.load(open('$f')); print(round(d['value']/1e6,2), 'M', {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')}, d['roofline']['traffic'], round(d['roofline']['frac'],3), d['roofline'].get('valu',{}).get('insts_per_launch'))"; done
