#!/bin/bash
# r08m: per-tick views derived in the kernels (k up to 16 ticks per launch): parity of batched calls, headline at 8 and 16 ticks per call
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r08m}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 1500 python -m pytest tests/test_pipelining_gpu.py tests/test_multitask_gpu.py tests/test_fast_pixels_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
$B > $OUT/tower_b8_bench.json 2> $OUT/err.txt
MV_PIPE_BATCH=16 $B --batch 16 > $OUT/tower_b16_bench.json 2>> $OUT/err.txt
$B --gpus 1 --steps 20 --warmup 5 > $OUT/tower_driver_style_bench.json 2>/dev/null
$B --scenario ObstaclesHard --envs-per-gpu 512 --pass-overlap off > $OUT/obst512_b8_nooverlap_bench.json 2>/dev/null
MV_PIPE_BATCH=16 $B --batch 16 --scenario ObstaclesHard --envs-per-gpu 512 --pass-overlap off > $OUT/obst512_b16_nooverlap_bench.json 2>/dev/null
MV_PIPE_BATCH=16 $B --batch 16 --scenario Collect > $OUT/collect_b16_bench.json 2>/dev/null
$B --scenario Collect > $OUT/collect_b8_bench.json 2>/dev/null
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_s -o run -- python $R/bench.py --steps 400 --warmup 48 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/tower_stats.log 2>&1
 python $R/scripts/rocpd_summary.py $OUT/db_s/run_results.db > $OUT/tower_kernel_stats.csv 2>> $OUT/tower_stats.log; rm -rf $OUT/db_s)
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*_bench*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]/1e6,2), "M", "raster/tick %.1f us step/tick %.1f us" % (d["roofline"]["avg_launch_ms"]*1e3, d["roofline_physics"]["avg_launch_ms"]*1e3), d["config"].get("ticks_per_call"))
    except Exception as e: print(f, "failed", e)
PY
grep -h "raster_fast\|step_ticks" $OUT/tower_kernel_stats.csv | cut -c1-200; tail -3 $OUT/err.txt
