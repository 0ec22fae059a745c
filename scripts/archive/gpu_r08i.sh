#!/bin/bash
# r08i: batched group calls -- one union step launch + one union observation launch per call (configs[4])
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r08i}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 1200 python -m pytest tests/test_multitask_gpu.py tests/test_full_size_gpu.py tests/test_full_size_oracle_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 100"
$B --scenario Mixed --obs 64 64 > $OUT/mixed_64_bench.json 2> $OUT/mixed_64.err
$B --scenario Mixed --obs 128 128 > $OUT/mixed_128_bench.json 2> /dev/null
$B --scenario Mixed4 --obs 64 64 > $OUT/mixed4_64_bench.json 2> /dev/null
MV_STEP_TICKS=0 $B --scenario Mixed --obs 64 64 > $OUT/mixed_64_tickbytick_bench.json 2> /dev/null
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_s -o run -- python $R/bench.py --scenario Mixed --obs 64 64 --steps 400 --warmup 48 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/mixed_stats.log 2>&1
 python $R/scripts/rocpd_summary.py $OUT/db_s/run_results.db > $OUT/mixed_64_kernel_stats.csv 2>> $OUT/mixed_stats.log; rm -rf $OUT/db_s)
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*_bench*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]/1e6,2), "M", "raster/tick %.1f us step/tick %.1f us" % (d["roofline"]["avg_launch_ms"]*1e3, d["roofline_physics"]["avg_launch_ms"]*1e3))
    except Exception as e: print(f, "failed", e)
PY
head -8 $OUT/mixed_64_kernel_stats.csv | cut -c1-200
tail -3 $OUT/mixed_64.err
