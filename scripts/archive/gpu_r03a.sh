#!/bin/bash
# round 3, first GPU pass: new tests (two pixels per lane, mv_step_n, output ring, single-bit policy, warnings), then the whole suite,
# then A/B bench lines for the raster variants and the batched stepping
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r03a}
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_fast_pixels_gpu.py tests/test_pipelining_gpu.py tests/test_refill_protocol_gpu.py -m gpu -q -x > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_new.log
tail -15 $OUT/pytest_new.log
B="python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --profile-steps 128"
MV_FAST_PPL=1 timeout 200 $B > $OUT/bench_ppl1.json 2> $OUT/bench_ppl1.err
MV_FAST_PPL=2 timeout 200 $B > $OUT/bench_ppl2_w7.json 2> $OUT/bench_ppl2_w7.err
MV_FAST_PPL=2 MV_FAST_WAVES=6 timeout 200 $B --no-extra-legs > $OUT/bench_ppl2_w6.json 2> $OUT/bench_ppl2_w6.err
MV_FAST_PPL=2 timeout 200 $B --batch 16 --no-extra-legs > $OUT/bench_ppl2_b16.json 2> $OUT/bench_ppl2_b16.err
MV_FAST_PPL=2 timeout 200 $B --batch 4 --no-extra-legs > $OUT/bench_ppl2_b4.json 2> $OUT/bench_ppl2_b4.err
for sc in Collect HexMemory ObstaclesHard; do
  for p in 1 2; do
    MV_FAST_PPL=$p timeout 200 $B --scenario $sc --no-extra-legs > $OUT/bench_${sc}_ppl$p.json 2> $OUT/bench_${sc}_ppl$p.err
  done
done
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_all.log
tail -5 $OUT/pytest_all.log
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        l=json.loads(open(f).read().strip().splitlines()[-1])
        r=l.get("roofline",{}); p=l.get("roofline_physics",{})
        print(os.path.basename(f), "%.2fM %.4fms"%(l["value"]/1e6,l["ms_per_step"]), {k:round(v/1e6,2) for k,v in l.items() if k.startswith("value_")}, "raster %.4f step %.4f"%(r.get("avg_launch_ms",0),p.get("avg_launch_ms",0)))
    except Exception as e:
        print(os.path.basename(f), "ERR", e, open(f.replace(".json",".err")).read()[-400:])
PY
