#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 900 python -m pytest tests/test_fast_pixels_gpu.py tests/test_parity_gpu.py -m gpu -q -x -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
B="python bench.py --steps 600 --warmup 50 --no-cpu-baseline"
$B > $OUT/bench_tower.json 2>&1
$B --scenario Collect > $OUT/bench_collect.json 2>&1
$B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/bench_obst.json 2>&1
$B --agents 4 --envs-per-gpu 512 > $OUT/bench_a4.json 2>&1
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --kernel-trace -d $OUT/db -o run -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 > $OUT/pmc.log 2>&1
python $R/scripts/rocpd_summary.py $OUT/db/run_results.db --pmc | grep raster_fast > $OUT/pmc_raster.csv
rm -rf $OUT/db
tail -3 $OUT/pytest.log
