#!/bin/bash
# round 3, tenth GPU pass: after the forward-and-strafe early-out -- headline with all legs (incl. the double-buffered closed loop), step-bound configs
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r03j}
mkdir -p $OUT
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
B="python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --profile-steps 64"
timeout 300 $B > $OUT/bench_tower.json 2> $OUT/bench_tower.err
timeout 300 $B --no-extra-legs --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/bench_obst512.json 2> $OUT/bench_obst512.err
timeout 300 $B --no-extra-legs --scenario ObstaclesHard > $OUT/bench_obst1024.json 2> $OUT/bench_obst1024.err
timeout 300 $B --agents 4 --envs-per-gpu 512 > $OUT/bench_a4.json 2> $OUT/bench_a4.err
timeout 300 $B --no-extra-legs --scenario Mixed --obs 64 64 > $OUT/bench_mixed64.json 2> $OUT/bench_mixed64.err
timeout 300 $B --no-extra-legs --scenario Collect > $OUT/bench_collect.json 2> $OUT/bench_collect.err
timeout 300 $B --no-extra-legs --scenario HexMemory > $OUT/bench_hexmemory.json 2> $OUT/bench_hexmemory.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_tower_driver.json 2> $OUT/bench_tower_driver.err
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        l=json.loads(open(f).read().strip().splitlines()[-1])
        r=l.get("roofline",{}); p=l.get("roofline_physics",{})
        print(os.path.basename(f), "%.2fM %.4fms"%(l["value"]/1e6,l["ms_per_step"]), "raster %.4f step %.4f"%(r.get("avg_launch_ms",0),p.get("avg_launch_ms",0)),
              " ".join("%s=%.2fM"%(k[6:],v/1e6) for k,v in l.items() if k.startswith("value_")))
    except Exception as e:
        print(os.path.basename(f), "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
