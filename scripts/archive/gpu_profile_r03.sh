#!/bin/bash
# Round-3 profile pass on the GPU box (via gpurun): gpu_profile_r03.sh <tag>
#   headline (TowerBuilding 1024x1 128x128): bench line (2000 steps, all legs, CPU baseline) + driver-style short run + rocprofv3 kernel stats +
#   PMC passes (FETCH_SIZE, WRITE_SIZE, SQ, SQ2: separate runs, never combined with other trace domains)
#   Collect, HexMemory: bench + kernel stats + FETCH / WRITE / SQ / SQ2 passes (the long-list raster variant)
#   Mixed 64x64 (configs[4], one GPU's share): bench (with CPU baseline) + kernel stats;  other configs: bench + kernel stats or bench only
# Summaries (CSV, from the rocpd databases with scripts/rocpd_summary.py) land in gpurun_out/<tag>/ and are copied into profiles/ by hand.
set -u
TAG=${1:-r03p}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"
stats() {   # stats <name> <bench args...>: kernel table of the same command, shorter
  local N=$1; shift
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_${N}_stats -o run -- python $R/bench.py "$@" --steps 400 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/${N}_stats.log 2>&1
  python $R/scripts/rocpd_summary.py $OUT/db_${N}_stats/run_results.db > $OUT/${N}_kernel_stats.csv 2>> $OUT/${N}_stats.log
  rm -rf $OUT/db_${N}_stats
}
prof() {    # prof <name> <bench args...>: bench line (no CPU baseline) + kernel table
  local N=$1; shift
  cd $R; timeout 400 python bench.py "$@" --no-cpu-baseline > $OUT/${N}_bench.json 2> $OUT/${N}_bench.err
  stats $N "$@"
}
pmc() {     # pmc <name> <counter-set-name> "<counters>" <bench args...>
  local N=$1 C=$2 L=$3; shift 3
  cd /tmp
  timeout 300 rocprofv3 --pmc $L --kernel-trace -d $OUT/db_${N}_$C -o run -- python $R/bench.py "$@" --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/${N}_pmc_$C.log 2>&1
  python $R/scripts/rocpd_summary.py $OUT/db_${N}_$C/run_results.db --pmc > $OUT/${N}_pmc_$C.csv 2>> $OUT/${N}_pmc_$C.log
  rm -rf $OUT/db_${N}_$C
}
cd $R; timeout 600 python bench.py > $OUT/tower_bench.json 2> $OUT/tower_bench.err
cd $R; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/tower_bench_driver_style.json 2> $OUT/tower_bench_driver_style.err
stats tower
pmc tower FETCH_SIZE FETCH_SIZE
pmc tower WRITE_SIZE WRITE_SIZE
pmc tower SQ "$SQ1"
pmc tower SQ2 "$SQ2"
for sc in Collect HexMemory; do
  prof $sc --scenario $sc
  pmc $sc FETCH_SIZE FETCH_SIZE --scenario $sc
  pmc $sc WRITE_SIZE WRITE_SIZE --scenario $sc
  pmc $sc SQ "$SQ1" --scenario $sc
  pmc $sc SQ2 "$SQ2" --scenario $sc
done
cd $R; timeout 600 python bench.py --scenario Mixed --obs 64 64 > $OUT/mixed_64_bench.json 2> $OUT/mixed_64_bench.err
stats mixed_64 --scenario Mixed --obs 64 64
prof obstacles_hard_512 --scenario ObstaclesHard --envs-per-gpu 512
prof tower_512x4 --agents 4 --envs-per-gpu 512
prof hexexplore --scenario HexExplore
prof rearrange --scenario Rearrange
prof sokoban --scenario Sokoban
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs"
$B --scenario Empty > $OUT/empty_bench.json 2> $OUT/empty_bench.err
$B --scenario Empty --envs-per-gpu 64 --obs 128 72 > $OUT/empty_64x128x72_bench.json 2> $OUT/empty_64_bench.err
$B --scenario Collect --envs-per-gpu 64 --obs 128 72 > $OUT/collect_64x128x72_bench.json 2> $OUT/collect_64_bench.err
$B --envs-per-gpu 4096 > $OUT/tower_4096_bench.json 2> $OUT/tower_4096_bench.err
$B --obs 128 72 > $OUT/tower_128x72_bench.json 2> $OUT/tower_128x72_bench.err
$B --pixels exact > $OUT/tower_exact_bench.json 2> $OUT/tower_exact_bench.err
$B --policy single-bit > $OUT/tower_single_bit_bench.json 2> $OUT/tower_single_bit_bench.err
$B --scenario Mixed --obs 128 128 > $OUT/mixed_128_bench.json 2> $OUT/mixed_128_bench.err
$B --scenario Mixed --obs 64 64 --envs-per-gpu 2048 > $OUT/mixed_64_n2048_bench.json 2> $OUT/mixed_64_n2048_bench.err
cd /tmp; MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_tu -o run -- python $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/tower_unpipelined_stats.log 2>&1
python $R/scripts/rocpd_summary.py $OUT/db_tu/run_results.db > $OUT/tower_unpipelined_kernel_stats.csv 2>> $OUT/tower_unpipelined_stats.log
rm -rf $OUT/db_tu
unpip() {   # unpip <name> <bench args...>: kernel table with the pipelining off (every kernel alone on the device)
  local N=$1; shift
  cd /tmp; MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_u -o run -- python $R/bench.py "$@" --steps 400 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/${N}_unpipelined_stats.log 2>&1
  python $R/scripts/rocpd_summary.py $OUT/db_u/run_results.db > $OUT/${N}_unpipelined_kernel_stats.csv 2>> $OUT/${N}_unpipelined_stats.log
  rm -rf $OUT/db_u
}
unpip tower_512x4 --agents 4 --envs-per-gpu 512
unpip obstacles_hard_512 --scenario ObstaclesHard --envs-per-gpu 512
# where a tick's time goes (an instrumented build of the library, -DMV_TICK_TIMING): phase cycles, cast statistics, launch lifetimes
if [ -f $R/megaverse_amd/_variants/libmv_ticktiming.so ]; then
  cd $R
  (export MV_TICK_TIMING=1 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_ticktiming.so
   PROBE_MODE=tick python scripts/probe_tick.py TowerBuilding 1024 1 2>&1 | grep "tick timing" > $OUT/tick_timing_tower_tick_only.txt
   PROBE_MODE=fused python scripts/probe_tick.py TowerBuilding 1024 1 2>&1 | grep "tick timing" > $OUT/tick_timing_tower_fused.txt)
fi
# timelines (kernel start / duration / queue): the driver-style 20-step run and the double-buffered closed loop
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/db_t20 -o run -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --profile-steps 0 > $OUT/timeline_20_steps.log 2>&1
python $R/scripts/kernel_timeline.py $OUT/db_t20/run_results.db 62 0 > $OUT/timeline_20_steps.txt 2>> $OUT/timeline_20_steps.log
timeout 300 rocprofv3 --kernel-trace -d $OUT/db_dbuf -o run -- python $R/scripts/probe_double_buffer.py 1024 100 > $OUT/timeline_double_buffered.log 2>&1
python $R/scripts/kernel_timeline.py $OUT/db_dbuf/run_results.db 36 > $OUT/timeline_double_buffered.txt 2>> $OUT/timeline_double_buffered.log
rm -rf $OUT/db_t20 $OUT/db_dbuf
cd $R; (timeout 900 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log)
find $OUT -name "*.db" -delete
ls $OUT
