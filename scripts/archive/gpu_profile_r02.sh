#!/bin/bash
# Round-2 profile pass on the GPU box (via gpurun): gpu_profile_r02.sh <tag>
#   (r02b: + HexMemory / HexExplore / Empty, the unpipelined headline, smoke())
#   headline (TowerBuilding 1024x1 128x128): bench line (2000 steps) + rocprofv3 kernel stats + PMC passes (FETCH_SIZE, WRITE_SIZE, SQ, SQ2: separate runs)
#   Collect 1024x1: bench + kernel stats + FETCH/WRITE/SQ passes;  other configs: bench + kernel stats
# Summaries (CSV, from the rocpd databases with scripts/rocpd_summary.py) land in gpurun_out/<tag>/ and are copied into profiles/ by hand.
set -u
TAG=${1:-r02p}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"
prof() {   # prof <name> <bench args...>
  local N=$1; shift
  cd $R; timeout 400 python bench.py "$@" > $OUT/${N}_bench.json 2> $OUT/${N}_bench.err
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_${N}_stats -o run -- python $R/bench.py "$@" --steps 400 --warmup 50 --no-cpu-baseline --profile-steps 0 > $OUT/${N}_stats.log 2>&1
  python $R/scripts/rocpd_summary.py $OUT/db_${N}_stats/run_results.db > $OUT/${N}_kernel_stats.csv 2>> $OUT/${N}_stats.log
}
pmc() {    # pmc <name> <counter-set-name> "<counters>" <bench args...>
  local N=$1 C=$2 L=$3; shift 3
  cd /tmp
  timeout 300 rocprofv3 --pmc $L --kernel-trace -d $OUT/db_${N}_$C -o run -- python $R/bench.py "$@" --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 > $OUT/${N}_pmc_$C.log 2>&1
  python $R/scripts/rocpd_summary.py $OUT/db_${N}_$C/run_results.db --pmc > $OUT/${N}_pmc_$C.csv 2>> $OUT/${N}_pmc_$C.log
}
prof tower
pmc tower FETCH_SIZE FETCH_SIZE
pmc tower WRITE_SIZE WRITE_SIZE
pmc tower SQ "$SQ1"
pmc tower SQ2 "$SQ2"
prof collect --scenario Collect
pmc collect FETCH_SIZE FETCH_SIZE --scenario Collect
pmc collect WRITE_SIZE WRITE_SIZE --scenario Collect
pmc collect SQ "$SQ1" --scenario Collect
prof obstacles_hard_512 --scenario ObstaclesHard --envs-per-gpu 512
prof tower_512x4 --agents 4 --envs-per-gpu 512
prof rearrange --scenario Rearrange
prof sokoban --scenario Sokoban
prof hexmemory --scenario HexMemory
prof hexexplore --scenario HexExplore
cd $R; timeout 300 python bench.py --scenario Empty --no-cpu-baseline > $OUT/empty_bench.json 2> $OUT/empty_bench.err
timeout 300 python bench.py --scenario Empty --envs-per-gpu 64 --obs 128 72 --no-cpu-baseline > $OUT/empty_64x128x72_bench.json 2> $OUT/empty_64_bench.err
timeout 300 python bench.py --scenario Collect --envs-per-gpu 64 --obs 128 72 --no-cpu-baseline > $OUT/collect_64x128x72_bench.json 2> $OUT/collect_64_bench.err
MV_PIPELINE=0 timeout 300 python bench.py --no-cpu-baseline > $OUT/tower_unpipelined_bench.json 2> $OUT/tower_unpipelined_bench.err
cd /tmp; MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_tower_unpipelined_stats -o run -- python $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline --profile-steps 0 > $OUT/tower_unpipelined_stats.log 2>&1
python $R/scripts/rocpd_summary.py $OUT/db_tower_unpipelined_stats/run_results.db > $OUT/tower_unpipelined_kernel_stats.csv 2>> $OUT/tower_unpipelined_stats.log
cd $R; (timeout 900 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log)
cd $R; timeout 300 python bench.py --scenario Mixed --obs 64 64 --no-cpu-baseline > $OUT/mixed_64_bench.json 2> $OUT/mixed_64_bench.err
timeout 300 python bench.py --envs-per-gpu 4096 --no-cpu-baseline > $OUT/tower_4096_bench.json 2> $OUT/tower_4096_bench.err
timeout 300 python bench.py --obs 128 72 --no-cpu-baseline > $OUT/tower_128x72_bench.json 2> $OUT/tower_128x72_bench.err
timeout 300 python bench.py --pixels exact --no-cpu-baseline > $OUT/tower_exact_bench.json 2> $OUT/tower_exact_bench.err
rm -rf $OUT/db_*
ls $OUT
