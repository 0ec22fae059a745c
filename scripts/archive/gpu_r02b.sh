#!/bin/bash
# round 2: fast-raster correctness + variants
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r02b}
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_fast_pixels_gpu.py -m gpu -q -s > $OUT/pytest_fast.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_fast.log
B="python bench.py --steps 600 --warmup 50 --no-cpu-baseline"
for sp in 2 4 8; do for wv in 6 8; do
MV_RASTER_SPLIT=$sp MV_FAST_WAVES=$wv $B > $OUT/bench_fast_s${sp}_w${wv}.json 2>&1
done; done
$B --pixels exact > $OUT/bench_exact.json 2>&1
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc_SQ -o run -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 > $OUT/pmc_SQ.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/pmc_SQ2 -o run -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 > $OUT/pmc_SQ2.log 2>&1
find $OUT -name "*.db" -size +20M -delete
