#!/bin/bash
# tests + standard bench set: gpu_tb.sh <outdir> [pytest targets...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest "${@:-tests}" -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
B="python bench.py --steps 600 --warmup 50 --no-cpu-baseline"
$B > $OUT/bench_tower.json 2>&1
$B --scenario Collect > $OUT/bench_collect.json 2>&1
$B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/bench_obst.json 2>&1
$B --agents 4 --envs-per-gpu 512 > $OUT/bench_a4.json 2>&1
$B --scenario Rearrange > $OUT/bench_rearr.json 2>&1
$B --scenario Mixed --obs 64 64 > $OUT/bench_mixed.json 2>&1
tail -5 $OUT/pytest.log
