#!/bin/bash
# round 3, sixth GPU pass: the long-list ("global list") raster variant: whole suite + Collect / Hex / Mixed timings
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r03f}
mkdir -p $OUT
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 1200 python -m pytest tests/test_fast_pixels_gpu.py tests/test_hex_parity_gpu.py tests/test_collect_parity_gpu.py tests/test_capacity_flags_gpu.py tests/test_multitask_gpu.py -m gpu -q -x > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_new.log
tail -25 $OUT/pytest_new.log
B="python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --profile-steps 64 --no-extra-legs"
for sc in Collect HexMemory HexExplore; do
  timeout 300 $B --scenario $sc > $OUT/bench_${sc}.json 2> $OUT/bench_${sc}.err
  MV_FAST_PPL=1 timeout 300 $B --scenario $sc > $OUT/bench_${sc}_ppl1.json 2> $OUT/bench_${sc}_ppl1.err
done
timeout 300 $B --scenario Mixed --obs 64 64 > $OUT/bench_mixed64.json 2> $OUT/bench_mixed64.err
timeout 300 $B --scenario Mixed --obs 128 128 > $OUT/bench_mixed128.json 2> $OUT/bench_mixed128.err
timeout 300 $B --scenario HexMemory --obs 64 64 > $OUT/bench_HexMemory64.json 2> $OUT/bench_HexMemory64.err
timeout 300 $B --scenario Collect --obs 64 64 > $OUT/bench_Collect64.json 2> $OUT/bench_Collect64.err
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_all.log
tail -8 $OUT/pytest_all.log
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
