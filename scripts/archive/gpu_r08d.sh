#!/bin/bash
# r08d: TowerBuilding episodes drawn ahead of time on a stream of their own (tower_draw_kernel) -- the whole GPU suite, the headline, kernel stats
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r08d}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
B="timeout 300 python bench.py --no-cpu-baseline"
$B > $OUT/tower_bench.json 2> $OUT/tower_bench.err
$B --gpus 1 --steps 20 --warmup 5 > $OUT/tower_bench_driver_style.json 2>/dev/null
$B --no-extra-legs --envs-per-gpu 512 --agents 4 > $OUT/tower_512x4_bench.json 2>/dev/null
$B --no-extra-legs --envs-per-gpu 4096 > $OUT/tower_4096_bench.json 2>/dev/null
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_s -o run -- python $R/bench.py --steps 400 --warmup 48 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/tower_stats.log 2>&1
 python $R/scripts/rocpd_summary.py $OUT/db_s/run_results.db > $OUT/tower_kernel_stats.csv 2>> $OUT/tower_stats.log; rm -rf $OUT/db_s)
(cd /tmp; MV_BENCH_CALL_SCHEDULE=8 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace -d $OUT/db_p -o run -- python $R/bench.py --batch 8 --steps 64 --warmup 16 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/pmc.log 2>&1
 python $R/scripts/rocpd_summary.py $OUT/db_p/run_results.db --pmc > $OUT/pmc.csv 2>> $OUT/pmc.log; rm -rf $OUT/db_p)
find $OUT -name "*.db" -delete
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*_bench*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]/1e6,2), "M", {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith("value_")}, "raster/tick %.1f us step/tick %.1f us" % (d["roofline"]["avg_launch_ms"]*1e3, d["roofline_physics"]["avg_launch_ms"]*1e3))
    except Exception as e: print(f, "failed", e)
PY
grep -h "raster_fast\|step_ticks\|tower_draw" $OUT/tower_kernel_stats.csv $OUT/pmc.csv | cut -c1-200
