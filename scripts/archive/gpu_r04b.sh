#!/bin/bash
# r04b: planar tiles, classified lane-per-tile in the workgroup's prologue: byte equality, then A/B timing (MV_PLANAR=0 / 1)
set -u
TAG=${1:-r04b}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_fast_pixels_gpu.py -x -q -k "planar or per_lane or full_size" > $OUT/pytest_planar.log 2>&1; echo "rc=$?" >> $OUT/pytest_planar.log
tail -5 $OUT/pytest_planar.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs"
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"
for P in 0 1; do
  MV_PLANAR=$P $B > $OUT/tower_planar$P.json 2> $OUT/tower_planar$P.err
  (cd /tmp; MV_PLANAR=$P MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_u$P -o run -- python $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/tower_unpipelined_planar$P.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_u$P/run_results.db > $OUT/tower_unpipelined_planar${P}_kernel_stats.csv 2>> $OUT/tower_unpipelined_planar$P.log; rm -rf $OUT/db_u$P)
  (cd /tmp; MV_PLANAR=$P timeout 300 rocprofv3 --pmc $SQ1 --kernel-trace -d $OUT/db_sq$P -o run -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/tower_pmc_SQ_planar$P.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_sq$P/run_results.db --pmc > $OUT/tower_pmc_SQ_planar$P.csv 2>> $OUT/tower_pmc_SQ_planar$P.log; rm -rf $OUT/db_sq$P)
  (cd /tmp; MV_PLANAR=$P timeout 300 rocprofv3 --pmc $SQ2 --kernel-trace -d $OUT/db_sq2$P -o run -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/tower_pmc_SQ2_planar$P.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_sq2$P/run_results.db --pmc > $OUT/tower_pmc_SQ2_planar$P.csv 2>> $OUT/tower_pmc_SQ2_planar$P.log; rm -rf $OUT/db_sq2$P)
done
for P in 0 1; do
  MV_PLANAR=$P $B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/obstacles_hard_512_planar$P.json 2> $OUT/obst_planar$P.err
  MV_PLANAR=$P $B --agents 4 --envs-per-gpu 512 > $OUT/tower_512x4_planar$P.json 2>> $OUT/obst_planar$P.err
  MV_PLANAR=$P $B --scenario Sokoban > $OUT/sokoban_planar$P.json 2>> $OUT/obst_planar$P.err
  MV_PLANAR=$P $B --scenario Rearrange > $OUT/rearrange_planar$P.json 2>> $OUT/obst_planar$P.err
  MV_PLANAR=$P $B --scenario Empty > $OUT/empty_planar$P.json 2>> $OUT/obst_planar$P.err
done
find $OUT -name "*.db" -delete
for f in $OUT/*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), (d.get('roofline') or {}).get('avg_launch_ms'))" 2>/dev/null)"; done
grep -h "raster_fast" $OUT/*kernel_stats.csv | cut -c1-200
grep -h "raster_fast.*INSTS_VALU\|raster_fast.*INSTS_SALU\|raster_fast.*INSTS_LDS" $OUT/*pmc*.csv | cut -c60-200
