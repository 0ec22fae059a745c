#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_empty_parity_gpu.py tests/test_obstacles_parity_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
B="python bench.py --steps 600 --warmup 50 --no-cpu-baseline"
$B --scenario Empty --obs 128 72 --envs-per-gpu 64 > $OUT/bench_empty_64.json 2>&1
$B --scenario Empty --obs 128 72 > $OUT/bench_empty_1024.json 2>&1
$B --scenario Collect --obs 128 72 --envs-per-gpu 64 > $OUT/bench_collect_64.json 2>&1
$B --scenario Collect --obs 128 72 > $OUT/bench_collect_1024_72.json 2>&1
tail -4 $OUT/pytest.log
