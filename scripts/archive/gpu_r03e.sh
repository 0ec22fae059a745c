#!/bin/bash
# round 3, fifth GPU pass: capacity-flag tests, two ranks on one device, pybind placement; split rule; pixels per lane at 64x64
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r03e}
mkdir -p $OUT
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 1200 python -m pytest tests/test_capacity_flags_gpu.py tests/test_distributed_gpu.py tests/test_env_surface_gpu.py tests/test_multitask_gpu.py -m gpu -q -s > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_new.log
tail -30 $OUT/pytest_new.log
B="python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --profile-steps 64 --no-extra-legs"
timeout 200 $B > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 300 $B --scenario Mixed --obs 64 64 > $OUT/bench_mixed64.json 2> $OUT/bench_mixed64.err
MV_FAST_PPL=1 timeout 300 $B --scenario Mixed --obs 64 64 > $OUT/bench_mixed64_ppl1.json 2> $OUT/bench_mixed64_ppl1.err
for sc in HexMemory Collect; do
  for p in 1 2; do
    MV_FAST_PPL=$p timeout 300 $B --scenario $sc --obs 64 64 > $OUT/bench_${sc}64_ppl$p.json 2> $OUT/bench_${sc}64_ppl$p.err
  done
done
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
