#!/bin/bash
# round 2, first GPU pass: numerics probe, full GPU test suite, fast-raster variants, rocprofv3 stats
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02a
mkdir -p $OUT
$R/scripts/_bin/probe_numerics > $OUT/probe.log 2>&1
cd $R
timeout 900 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
B="python bench.py --steps 600 --warmup 50 --no-cpu-baseline"
$B > $OUT/bench_fast_default.json 2> $OUT/bench_fast_default.err
MV_RASTER_SPLIT=4 $B > $OUT/bench_fast_split4.json 2>&1
MV_RASTER_SPLIT=16 $B > $OUT/bench_fast_split16.json 2>&1
MV_FAST_WAVES=6 $B > $OUT/bench_fast_waves6.json 2>&1
$B --pixels exact > $OUT/bench_exact.json 2>&1
$B --scenario Collect > $OUT/bench_fast_collect.json 2>&1
$B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/bench_fast_obst.json 2>&1
$B --agents 4 --envs-per-gpu 512 > $OUT/bench_fast_a4.json 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o run -- python $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline --profile-steps 0 > $OUT/prof_stats.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --kernel-trace -d $OUT/pmc_SQ -o run -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 > $OUT/pmc_SQ.log 2>&1
find $OUT -name "*.db" -size +20M -delete
ls -la $OUT
