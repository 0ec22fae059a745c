#!/bin/bash
# round 3, second GPU pass: one divide per conservative-advancement iteration + exact cast pre-reject (parity suite, canonical poses), then
# timing experiments: step workgroups of 64 threads, the TowerBuilding generator's LDS
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r03b}
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_all.log
tail -5 $OUT/pytest_all.log
B="python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --profile-steps 128 --no-extra-legs"
timeout 200 $B > $OUT/bench_default.json 2> $OUT/bench_default.err
MV_PIPELINE=0 timeout 200 $B --batch 1 > $OUT/bench_unpipelined.json 2> $OUT/bench_unpipelined.err
for v in t64 noreset noreset_t64; do
  MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_$v.so timeout 200 $B > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_$v.so timeout 200 $B --batch 1 > $OUT/bench_${v}_unpipelined.json 2> $OUT/bench_${v}_unpipelined.err
done
timeout 200 $B --agents 4 --envs-per-gpu 512 > $OUT/bench_a4.json 2> $OUT/bench_a4.err
MV_PIPELINE=0 timeout 200 $B --agents 4 --envs-per-gpu 512 --batch 1 > $OUT/bench_a4_unpipelined.json 2> $OUT/bench_a4_unpipelined.err
timeout 200 $B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/bench_obst512.json 2> $OUT/bench_obst512.err
MV_PIPELINE=0 timeout 200 $B --scenario ObstaclesHard --envs-per-gpu 512 --batch 1 > $OUT/bench_obst512_unpipelined.json 2> $OUT/bench_obst512_unpipelined.err
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
