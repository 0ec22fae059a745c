#!/bin/bash
# Round-4 profile pass on the GPU box (via gpurun): gpu_profile_r04.sh <tag>
#   headline (TowerBuilding 1024x1 128x128): bench line (2000 steps, all legs, CPU baseline) + driver-style short run + rocprofv3 kernel stats
#   (pipelined default: the batched kernels; unpipelined one tick per call: every kernel alone) + PMC passes (FETCH_SIZE, WRITE_SIZE, SQ, SQ2: separate
#   runs, never combined with other trace domains) on the default (batched) path; other configs: bench + kernel stats or bench only; timelines; smoke.
# Summaries (CSV, from the rocpd databases with scripts/rocpd_summary.py) land in gpurun_out/<tag>/ and are copied into profiles/ by hand.
set -u
TAG=${1:-r04p}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"
stats() {   # stats <name> <bench args...>: kernel table of the same command, shorter
  local N=$1; shift
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_${N}_stats -o run -- python $R/bench.py "$@" --steps 400 --warmup 48 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/${N}_stats.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_${N}_stats/run_results.db > $OUT/${N}_kernel_stats.csv 2>> $OUT/${N}_stats.log; rm -rf $OUT/db_${N}_stats)
}
unpip() {   # unpip <name> <bench args...>: kernel table with the pipelining off, one tick per call (every kernel alone on the device)
  local N=$1; shift
  (cd /tmp; MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_u -o run -- python $R/bench.py "$@" --steps 400 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/${N}_unpipelined_stats.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_u/run_results.db > $OUT/${N}_unpipelined_kernel_stats.csv 2>> $OUT/${N}_unpipelined_stats.log; rm -rf $OUT/db_u)
}
prof() {    # prof <name> <bench args...>: bench line (no CPU baseline) + kernel table
  local N=$1; shift
  (cd $R; timeout 400 python bench.py "$@" --no-cpu-baseline > $OUT/${N}_bench.json 2> $OUT/${N}_bench.err)
  stats $N "$@"
}
pmc() {     # pmc <name> <counter-set-name> "<counters>" <bench args...>
  local N=$1 C=$2 L=$3; shift 3
  (cd /tmp; timeout 300 rocprofv3 --pmc $L --kernel-trace -d $OUT/db_${N}_$C -o run -- python $R/bench.py "$@" --steps 64 --warmup 16 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/${N}_pmc_$C.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_${N}_$C/run_results.db --pmc > $OUT/${N}_pmc_$C.csv 2>> $OUT/${N}_pmc_$C.log; rm -rf $OUT/db_${N}_$C)
}
cd $R; timeout 900 python bench.py > $OUT/tower_bench.json 2> $OUT/tower_bench.err
cd $R; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/tower_bench_driver_style.json 2> $OUT/tower_bench_driver_style.err
stats tower
unpip tower
pmc tower FETCH_SIZE FETCH_SIZE --batch 8
pmc tower WRITE_SIZE WRITE_SIZE --batch 8
pmc tower SQ "$SQ1" --batch 8
pmc tower SQ2 "$SQ2" --batch 8
prof tower_512x4 --agents 4 --envs-per-gpu 512
unpip tower_512x4 --agents 4 --envs-per-gpu 512
prof obstacles_hard_512 --scenario ObstaclesHard --envs-per-gpu 512
unpip obstacles_hard_512 --scenario ObstaclesHard --envs-per-gpu 512
for sc in Collect HexMemory; do
  prof $sc --scenario $sc
done
cd $R; timeout 900 python bench.py --scenario Mixed --obs 64 64 > $OUT/mixed_64_bench.json 2> $OUT/mixed_64_bench.err
stats mixed_64 --scenario Mixed --obs 64 64
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs"
$B --scenario HexExplore > $OUT/hexexplore_bench.json 2> $OUT/hexexplore_bench.err
$B --scenario Rearrange > $OUT/rearrange_bench.json 2> $OUT/rearrange_bench.err
$B --scenario Sokoban > $OUT/sokoban_bench.json 2> $OUT/sokoban_bench.err
$B --scenario Empty > $OUT/empty_bench.json 2> $OUT/empty_bench.err
$B --scenario ObstaclesHard > $OUT/obstacles_hard_1024_bench.json 2> $OUT/obstacles_hard_1024_bench.err
$B --scenario Empty --envs-per-gpu 64 --obs 128 72 > $OUT/empty_64x128x72_bench.json 2> $OUT/empty_64_bench.err
$B --scenario Collect --envs-per-gpu 64 --obs 128 72 > $OUT/collect_64x128x72_bench.json 2> $OUT/collect_64_bench.err
$B --envs-per-gpu 4096 > $OUT/tower_4096_bench.json 2> $OUT/tower_4096_bench.err
$B --envs-per-gpu 16384 --steps 400 --warmup 48 > $OUT/tower_16384_bench.json 2> $OUT/tower_16384_bench.err
$B --obs 128 72 > $OUT/tower_128x72_bench.json 2> $OUT/tower_128x72_bench.err
$B --pixels exact > $OUT/tower_exact_bench.json 2> $OUT/tower_exact_bench.err
$B --policy single-bit > $OUT/tower_single_bit_bench.json 2> $OUT/tower_single_bit_bench.err
$B --scenario Mixed --obs 128 128 > $OUT/mixed_128_bench.json 2> $OUT/mixed_128_bench.err
$B --scenario Mixed --obs 64 64 --envs-per-gpu 2048 > $OUT/mixed_64_n2048_bench.json 2> $OUT/mixed_64_n2048_bench.err
MV_PLANAR=0 $B > $OUT/tower_no_planar_bench.json 2> $OUT/tower_no_planar_bench.err
MV_STEP_TICKS=0 $B > $OUT/tower_no_multitick_bench.json 2> $OUT/tower_no_multitick_bench.err
# timelines (kernel start / duration / queue): the batched default, the driver-style 20-step run, the closed loop
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/db_t -o run -- python $R/bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-extra-legs --profile-steps 0 > $OUT/timeline_batched.log 2>&1
python $R/scripts/kernel_timeline.py $OUT/db_t/run_results.db 30 10 > $OUT/timeline_batched.txt 2>> $OUT/timeline_batched.log; rm -rf $OUT/db_t
timeout 300 rocprofv3 --kernel-trace -d $OUT/db_t20 -o run -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --profile-steps 0 > $OUT/timeline_20_steps.log 2>&1
python $R/scripts/kernel_timeline.py $OUT/db_t20/run_results.db 40 0 > $OUT/timeline_20_steps.txt 2>> $OUT/timeline_20_steps.log; rm -rf $OUT/db_t20
timeout 300 rocprofv3 --kernel-trace -d $OUT/db_cl -o run -- python $R/scripts/probe_closed_loop.py 1024 100 > $OUT/timeline_closed_loop.log 2>&1
python $R/scripts/kernel_timeline.py $OUT/db_cl/run_results.db 30 30 > $OUT/timeline_closed_loop.txt 2>> $OUT/timeline_closed_loop.log; rm -rf $OUT/db_cl
# where a raster workgroup's time goes (instrumented build, if shipped)
if [ -f $R/megaverse_amd/_variants/libmv_rtiming.so ]; then
  cd $R; MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 2>&1 >/dev/null | grep "raster timing" > $OUT/raster_timing.txt
fi
cd $R; timeout 300 python scripts/probe_cpu_scaling.py 1 4 16 32 64 > $OUT/cpu_scaling.txt 2>&1
cd $R; (timeout 900 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log)
find $OUT -name "*.db" -delete
ls $OUT | wc -l
for f in $OUT/*_bench.json $OUT/tower_bench_driver_style.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), (d.get('roofline') or {}).get('avg_launch_ms'), (d.get('roofline_physics') or {}).get('avg_launch_ms'), {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')})" 2>/dev/null)"; done
