#!/bin/bash
# round 6, the closing pass (after r10a-u): the whole GPU suite, the PMC passes of the headline on the final kernel sources (-> profiles/r10z_*, profiles/pmc_traffic.json), the bench lines
set -u
TAG=${1:-r10z}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 1200 python scripts/soak.py 6000 > $OUT/soak.log 2>&1; tail -2 $OUT/soak.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
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
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_s -o run -- python $R/bench.py --steps 800 --warmup 96 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/tower_stats.log 2>&1
 python $R/scripts/rocpd_summary.py $OUT/db_s/run_results.db > $OUT/tower_kernel_stats.csv 2>> $OUT/tower_stats.log
 python $R/scripts/kernel_timeline.py $OUT/db_s/run_results.db 40 20 > $OUT/timeline_batched.txt 2>/dev/null; rm -rf $OUT/db_s)
(cd /tmp; MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_u -o run -- python $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/tower_unpipelined_stats.log 2>&1
 python $R/scripts/rocpd_summary.py $OUT/db_u/run_results.db > $OUT/tower_unpipelined_kernel_stats.csv 2>> $OUT/tower_unpipelined_stats.log; rm -rf $OUT/db_u)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace -d $OUT/db_c -o run -- python $R/scripts/probe_closed_loop.py 1024 200 > $OUT/closed_loop.log 2>&1
 python $R/scripts/kernel_timeline.py $OUT/db_c/run_results.db 60 40 > $OUT/timeline_closed_loop.txt 2>/dev/null; rm -rf $OUT/db_c)
(cd /tmp; export BOXOBAN_LEVELS=$R/tests/golden/boxoban; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_m -o run -- python $R/bench.py --scenario Mixed --obs 64 64 --steps 240 --warmup 48 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/mixed_64_stats.log 2>&1
 python $R/scripts/rocpd_summary.py $OUT/db_m/run_results.db > $OUT/mixed_64_kernel_stats.csv 2>> $OUT/mixed_64_stats.log
 python $R/scripts/kernel_timeline.py $OUT/db_m/run_results.db 60 40 > $OUT/timeline_mixed_64.txt 2>/dev/null; rm -rf $OUT/db_m)
cd $R
timeout 900 python bench.py > $OUT/tower_bench.json 2> $OUT/tower_bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/tower_bench_driver_style.json 2> $OUT/tower_bench_driver_style.err
find $OUT -name "*.db" -delete
grep -h "raster_fast\|step_ticks\|tower_draw" $OUT/tower_kernel_stats.csv $OUT/tower_unpipelined_kernel_stats.csv | cut -c1-160
for f in $OUT/tower_bench.json $OUT/tower_bench_driver_style.json; do python -c "import json; d=json.load(open('$f')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')}, d['roofline']['traffic'], d['roofline']['frac'], d.get('host_enqueue_ms_per_step_closed_loop_double_buffered'))"; done
# the other configurations, one bench line each (no CPU baseline, no extra legs)
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
$B --envs-per-gpu 512 > $OUT/tower_512_bench.json 2> /dev/null
$B --envs-per-gpu 512 --agents 4 > $OUT/tower_512x4_bench.json 2> /dev/null
$B --envs-per-gpu 4096 > $OUT/tower_4096_bench.json 2> /dev/null
$B --batch 8 > $OUT/tower_8_ticks_per_call_bench.json 2> /dev/null
MV_SIM_PRIORITY=normal $B > $OUT/tower_normal_priority_bench.json 2> /dev/null
MV_SIM_PRIORITY=normal $B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/obstacles_hard_512_normal_priority_bench.json 2> /dev/null
$B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/obstacles_hard_512_bench.json 2> /dev/null
$B --scenario ObstaclesHard --envs-per-gpu 512 --pass-overlap off > $OUT/obstacles_hard_512_no_overlap_bench.json 2> /dev/null
$B --scenario ObstaclesHard > $OUT/obstacles_hard_1024_bench.json 2> /dev/null
MV_STEP_PIPE=0 $B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/obstacles_hard_512_one_wave_step_bench.json 2> /dev/null
MV_STEP_PIPE=0 $B --envs-per-gpu 512 > $OUT/tower_512_one_wave_step_bench.json 2> /dev/null
$B --envs-per-gpu 256 > $OUT/tower_256_bench.json 2> /dev/null
for s in Collect HexMemory HexExplore Rearrange Sokoban Empty; do $B --scenario $s > $OUT/${s}_bench.json 2> /dev/null; done
$B --obs 128 72 > $OUT/tower_128x72_bench.json 2> /dev/null
$B --obs 64 64 > $OUT/tower_64x64_bench.json 2> /dev/null
$B --scenario Collect --obs 128 72 > $OUT/Collect_128x72_bench.json 2> /dev/null
$B --pixels exact --steps 400 > $OUT/tower_exact_pixels_bench.json 2> /dev/null
$B --policy single-bit > $OUT/tower_single_bit_bench.json 2> /dev/null
MV_STEP_PIPE=1 $B > $OUT/tower_step_pipe_bench.json 2> /dev/null
MV_STEP_PIPE=0 $B --scenario Empty > $OUT/Empty_one_wave_step_bench.json 2> /dev/null
MV_PLANAR=0 $B > $OUT/tower_planar_off_bench.json 2> /dev/null
$B --scenario Empty --steps 800 > $OUT/Empty_800_steps_bench.json 2> /dev/null
$B --scenario Mixed --obs 128 128 > $OUT/mixed_128_bench.json 2> /dev/null
$B --scenario Mixed --obs 64 64 > $OUT/mixed_64_bench.json 2> /dev/null
$B --scenario Mixed4 --obs 64 64 > $OUT/mixed4_64_bench.json 2> /dev/null
MV_STEP_TICKS=0 $B --batch 8 > $OUT/tower_no_multitick_bench.json 2> /dev/null
for f in $OUT/*_bench.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$f', 'failed', e)
"; done
