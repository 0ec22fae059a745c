#!/bin/bash
# r08a: the new tests (Python surface vs the reference's classes, overlapped passes with a late consumer) + the headline as the tree stands
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08a; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_py_surface_gpu.py tests/test_pipelining_gpu.py tests/test_env_surface_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline"
$B > $OUT/tower_bench.json 2> $OUT/tower_bench.err
$B --gpus 1 --steps 20 --warmup 5 > $OUT/tower_bench_driver_style.json 2>/dev/null
$B --no-extra-legs --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/obst512_bench.json 2>/dev/null
$B --no-extra-legs --scenario Empty > $OUT/empty_bench.json 2>/dev/null
$B --no-extra-legs --scenario Empty --steps 2000 > $OUT/empty2000_bench.json 2>/dev/null
bash scripts/show_bench.sh $OUT/*_bench*.json
