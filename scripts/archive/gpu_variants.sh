#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
B="python bench.py --steps 600 --warmup 50 --no-cpu-baseline"
$B > $OUT/bench_tower_t128.json 2>&1
$B --scenario Collect > $OUT/bench_collect_t128.json 2>&1
$B --agents 4 --envs-per-gpu 512 > $OUT/bench_a4_t128.json 2>&1
$B --envs-per-gpu 4096 > $OUT/bench_n4096_t128.json 2>&1
for t in 64 192 256; do
export MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_t$t.so
$B > $OUT/bench_tower_t$t.json 2>&1
$B --scenario Collect > $OUT/bench_collect_t$t.json 2>&1
$B --agents 4 --envs-per-gpu 512 > $OUT/bench_a4_t$t.json 2>&1
$B --envs-per-gpu 4096 > $OUT/bench_n4096_t$t.json 2>&1
done
