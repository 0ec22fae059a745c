#!/bin/bash
# further evidence on the final code, no code change: kernel tables and SQ counters of the long-list scenarios and of ObstaclesHard 512 / Mixed4, the headline's side variants
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08z_extra; mkdir -p $OUT
export TMPDIR=/tmp BOXOBAN_LEVELS=$R/tests/golden/boxoban
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"
stats() { local tag=$1; shift
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db -o run -- python $R/bench.py "$@" --steps 400 --warmup 96 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/${tag}_stats.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db/run_results.db > $OUT/${tag}_kernel_stats.csv 2>> $OUT/${tag}_stats.log; rm -rf $OUT/db) }
sq() { local tag=$1; shift
  (cd /tmp; timeout 300 rocprofv3 --pmc $SQ1 --kernel-trace -d $OUT/db -o run -- python $R/bench.py "$@" --steps 128 --warmup 32 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/${tag}_pmc_SQ.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db/run_results.db --pmc > $OUT/${tag}_pmc_SQ.csv 2>> $OUT/${tag}_pmc_SQ.log; rm -rf $OUT/db)
  (cd /tmp; timeout 300 rocprofv3 --pmc $SQ2 --kernel-trace -d $OUT/db -o run -- python $R/bench.py "$@" --steps 128 --warmup 32 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/${tag}_pmc_SQ2.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db/run_results.db --pmc > $OUT/${tag}_pmc_SQ2.csv 2>> $OUT/${tag}_pmc_SQ2.log; rm -rf $OUT/db) }
stats Collect --scenario Collect
stats HexMemory --scenario HexMemory
stats obstacles_hard_512 --scenario ObstaclesHard --envs-per-gpu 512
stats mixed4_64 --scenario Mixed4 --obs 64 64
stats Empty --scenario Empty
sq Collect --scenario Collect
sq HexMemory --scenario HexMemory
sq Empty --scenario Empty
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
$B --envs-per-gpu 16384 > $OUT/tower_16384_bench.json 2> /dev/null
$B --obs 128 72 > $OUT/tower_128x72_bench.json 2> /dev/null
$B --pixels exact > $OUT/tower_exact_pixels_bench.json 2> /dev/null
$B --policy single-bit > $OUT/tower_single_bit_bench.json 2> /dev/null
MV_PLANAR=0 $B > $OUT/tower_planar_off_bench.json 2> /dev/null
$B --scenario Mixed --obs 64 64 --envs-per-gpu 2048 > $OUT/mixed_64_2048_bench.json 2> /dev/null
$B --scenario Empty --envs-per-gpu 64 --obs 128 72 > $OUT/empty_64x128x72_bench.json 2> /dev/null
$B --scenario Collect --obs 128 72 > $OUT/Collect_128x72_bench.json 2> /dev/null
for f in $OUT/*_bench.json; do python -c "
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3), d['config'].get('ticks_per_call'))
except Exception as e: print('$f', 'failed', e)
"; done
for t in Collect HexMemory obstacles_hard_512 mixed4_64 Empty; do echo "== $t"; grep 'raster_\|step_' $OUT/${t}_kernel_stats.csv | head -3 | cut -c1-150; done
for t in Collect HexMemory Empty; do echo "== $t"; grep 'raster_.*SQ_INSTS_VALU\|raster_.*SQ_ACTIVE_INST_VALU' $OUT/${t}_pmc_SQ.csv | cut -c1-40,100-200; grep 'raster_.*SQ_INSTS_SALU' $OUT/${t}_pmc_SQ2.csv | cut -c1-40,100-200; done
