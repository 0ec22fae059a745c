#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
B="python bench.py --steps 600 --warmup 50 --no-cpu-baseline"
$B > $OUT/bench_tower.json 2>&1
$B --scenario Collect > $OUT/bench_collect.json 2>&1
python -m pytest tests/test_collect_parity_gpu.py tests/test_fast_pixels_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db -o run -- python $R/bench.py --scenario Collect --steps 300 --warmup 50 --no-cpu-baseline --profile-steps 0 > $OUT/stats.log 2>&1
python $R/scripts/rocpd_summary.py $OUT/db/run_results.db > $OUT/collect_kernel_stats.csv
rm -rf $OUT/db
