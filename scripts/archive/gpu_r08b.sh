#!/bin/bash
# r08b: raster -- highlight bound, overlay tiles, one-round general path, 16-byte clears, v_and_or: parity of the paths, the headline, VALU count
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r08b}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_fast_pixels_gpu.py tests/test_canonical_frames_gpu.py tests/test_py_surface_gpu.py tests/test_pipelining_gpu.py tests/test_full_size_gpu.py tests/test_multitask_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs"
$B > $OUT/tower_bench.json 2> $OUT/tower_bench.err
MV_PLANAR=2 $B > $OUT/tower_nooverlay_bench.json 2>/dev/null
$B --gpus 1 --steps 20 --warmup 5 > $OUT/tower_bench_driver_style.json 2>/dev/null
$B --scenario Empty --steps 400 > $OUT/empty_bench.json 2>/dev/null
$B --envs-per-gpu 512 --agents 4 > $OUT/tower_512x4_bench.json 2>/dev/null
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
(cd /tmp; MV_BENCH_CALL_SCHEDULE=8 timeout 300 rocprofv3 --pmc $SQ1 --kernel-trace -d $OUT/db_SQ -o run -- python $R/bench.py --batch 8 --steps 64 --warmup 16 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/tower_pmc_SQ.log 2>&1
 python $R/scripts/rocpd_summary.py $OUT/db_SQ/run_results.db --pmc > $OUT/tower_pmc_SQ.csv 2>> $OUT/tower_pmc_SQ.log; rm -rf $OUT/db_SQ)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_s -o run -- python $R/bench.py --steps 400 --warmup 48 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/tower_stats.log 2>&1
 python $R/scripts/rocpd_summary.py $OUT/db_s/run_results.db > $OUT/tower_kernel_stats.csv 2>> $OUT/tower_stats.log; rm -rf $OUT/db_s)
find $OUT -name "*.db" -delete
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*_bench*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]/1e6,2), "M", "raster/tick %.1f us step/tick %.1f us" % (d["roofline"]["avg_launch_ms"]*1e3, d["roofline_physics"]["avg_launch_ms"]*1e3))
    except Exception as e: print(f, "failed", e)
PY
grep -h "raster_fast\|step_ticks" $OUT/tower_kernel_stats.csv $OUT/tower_pmc_SQ.csv | cut -c1-220
