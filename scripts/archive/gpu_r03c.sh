#!/bin/bash
# round 3, third GPU pass: mv_group (union step + union raster) tests, Mixed bench, raster split sweep, sim-stream priority
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r03c}
mkdir -p $OUT
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 900 python -m pytest tests/test_multitask_gpu.py tests/test_parity_gpu.py tests/test_pipelining_gpu.py -m gpu -q -x > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_new.log
tail -15 $OUT/pytest_new.log
B="python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --profile-steps 128"
timeout 300 $B --scenario Mixed --obs 64 64 > $OUT/bench_mixed64.json 2> $OUT/bench_mixed64.err
MV_MULTITASK_UNION=0 timeout 300 $B --scenario Mixed --obs 64 64 --batch 1 --profile-steps 0 > $OUT/bench_mixed64_r02scheme.json 2> $OUT/bench_mixed64_r02scheme.err
timeout 300 $B --scenario Mixed --obs 128 128 --no-extra-legs > $OUT/bench_mixed128.json 2> $OUT/bench_mixed128.err
for sp in 2 8; do
  MV_RASTER_SPLIT=$sp timeout 200 $B --no-extra-legs > $OUT/bench_split$sp.json 2> $OUT/bench_split$sp.err
done
MV_SIM_PRIORITY=low timeout 200 $B --no-extra-legs > $OUT/bench_simlow.json 2> $OUT/bench_simlow.err
MV_SIM_PRIORITY=high timeout 200 $B --no-extra-legs > $OUT/bench_simhigh.json 2> $OUT/bench_simhigh.err
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_all.log
tail -5 $OUT/pytest_all.log
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        l=json.loads(open(f).read().strip().splitlines()[-1])
        r=l.get("roofline",{}); p=l.get("roofline_physics",{})
        print(os.path.basename(f), "%.2fM %.4fms"%(l["value"]/1e6,l["ms_per_step"]), {k:round(v/1e6,2) for k,v in l.items() if k.startswith("value_")}, "raster %.4f step %.4f"%(r.get("avg_launch_ms",0),p.get("avg_launch_ms",0)))
    except Exception as e:
        print(os.path.basename(f), "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
