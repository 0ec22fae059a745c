#!/bin/bash
# gpu_quick.sh <outdir> "<pytest targets>" ; then tower / a4 benches
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest $2 -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
B="python bench.py --steps 600 --warmup 50 --no-cpu-baseline"
$B > $OUT/bench_tower.json 2>&1
$B --agents 4 --envs-per-gpu 512 > $OUT/bench_a4.json 2>&1
$B --agents 8 --envs-per-gpu 256 > $OUT/bench_a8.json 2>&1
tail -5 $OUT/pytest.log
