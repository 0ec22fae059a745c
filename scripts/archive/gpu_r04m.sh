#!/bin/bash
# r04m: k ticks of every env with one step launch + the k passes with one raster launch
set -u
TAG=${1:-r04m}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_pipelining_gpu.py tests/test_parity_gpu.py tests/test_full_size_gpu.py -x -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0"
run() { local N=$1; shift; env "$@" $B > $OUT/$N.json 2> $OUT/$N.err; echo "$N: $(python -c "import json; d=json.load(open('$OUT/$N.json')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4))" 2>/dev/null)"; }
run ticks1_batch8 MV_STEP_TICKS=1 MV_RASTER_BATCH=8
run ticks1_batch0 MV_STEP_TICKS=1 MV_RASTER_BATCH=0
run ticks0_batch8 MV_STEP_TICKS=0 MV_RASTER_BATCH=8
run ticks0_batch0 MV_STEP_TICKS=0 MV_RASTER_BATCH=0
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0 --gpus 1 --steps 20 --warmup 5"
run driver_ticks1_batch8 MV_STEP_TICKS=1 MV_RASTER_BATCH=8
run driver_ticks0_batch0 MV_STEP_TICKS=0 MV_RASTER_BATCH=0
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0 --gpus 1 --steps 20 --warmup 5 --batch 4"
run driver_b4_ticks1_batch8 MV_STEP_TICKS=1 MV_RASTER_BATCH=8
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0 --agents 4 --envs-per-gpu 512"
run a4_ticks1_batch8 MV_STEP_TICKS=1 MV_RASTER_BATCH=8
run a4_ticks0_batch0 MV_STEP_TICKS=0 MV_RASTER_BATCH=0
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/db_t -o run -- python $R/bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-extra-legs --profile-steps 0 > $OUT/timeline.log 2>&1
python $R/scripts/kernel_timeline.py $OUT/db_t/run_results.db 24 20 > $OUT/timeline.txt 2>> $OUT/timeline.log
rm -rf $OUT/db_t
cat $OUT/timeline.txt
