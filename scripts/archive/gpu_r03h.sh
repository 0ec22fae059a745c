#!/bin/bash
# round 3, eighth GPU pass: split by frame count (Mixed), sanity of the other configs
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r03h}
mkdir -p $OUT
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 900 python -m pytest tests/test_fast_pixels_gpu.py tests/test_multitask_gpu.py -m gpu -q -x > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_new.log
tail -5 $OUT/pytest_new.log
B="python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --profile-steps 64 --no-extra-legs"
timeout 300 $B --scenario Mixed --obs 64 64 > $OUT/bench_mixed64.json 2> $OUT/bench_mixed64.err
timeout 300 $B --scenario Mixed --obs 128 128 > $OUT/bench_mixed128.json 2> $OUT/bench_mixed128.err
timeout 300 $B --scenario Mixed --obs 64 64 --envs-per-gpu 2048 > $OUT/bench_mixed64_n2048.json 2> $OUT/bench_mixed64_n2048.err
timeout 300 $B > $OUT/bench_tower.json 2> $OUT/bench_tower.err
timeout 300 $B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/bench_obst512.json 2> $OUT/bench_obst512.err
timeout 300 $B --agents 4 --envs-per-gpu 512 > $OUT/bench_a4.json 2> $OUT/bench_a4.err
timeout 300 $B --scenario Collect --envs-per-gpu 256 > $OUT/bench_collect256.json 2> $OUT/bench_collect256.err
cd /tmp
MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_mixed_64 -o run -- python $R/bench.py --scenario Mixed --obs 64 64 --steps 200 --warmup 20 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/mixed_stats_64.log 2>&1
python $R/scripts/rocpd_summary.py $OUT/db_mixed_64/run_results.db > $OUT/mixed64_unpipelined_kernel_stats.csv 2>> $OUT/mixed_stats_64.log
rm -rf $OUT/db_*
cd $R
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        l=json.loads(open(f).read().strip().splitlines()[-1])
        r=l.get("roofline",{}); p=l.get("roofline_physics",{})
        print(os.path.basename(f), "%.2fM %.4fms"%(l["value"]/1e6,l["ms_per_step"]), "raster %.4f step %.4f"%(r.get("avg_launch_ms",0),p.get("avg_launch_ms",0)))
    except Exception as e:
        print(os.path.basename(f), "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
for f in $OUT/*kernel_stats.csv; do echo $f; sed -n 3,6p $f | cut -c1-150; done
