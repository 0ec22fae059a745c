#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r02c}
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_fast_pixels_gpu.py tests/test_parity_gpu.py -m gpu -q -s > $OUT/pytest_fast.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_fast.log
B="python bench.py --steps 600 --warmup 50 --no-cpu-baseline"
for sp in 2 4 8 16; do
MV_RASTER_SPLIT=$sp $B > $OUT/bench_fast_s${sp}.json 2>&1
done
$B --scenario Collect > $OUT/bench_fast_collect.json 2>&1
$B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/bench_fast_obst.json 2>&1
$B --agents 4 --envs-per-gpu 512 > $OUT/bench_fast_a4.json 2>&1
$B --scenario Rearrange > $OUT/bench_fast_rearr.json 2>&1
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc_SQ -o run -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 > $OUT/pmc_SQ.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/pmc_SQ2 -o run -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 > $OUT/pmc_SQ2.log 2>&1
find $OUT -name "*.db" -size +20M -delete
