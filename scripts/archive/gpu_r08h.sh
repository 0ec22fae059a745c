#!/bin/bash
# r08h: every configuration's bench line on the current tree (no CPU baseline), the headline with its extra legs
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r08h}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 300 python bench.py --no-cpu-baseline > $OUT/tower_bench.json 2> $OUT/tower_bench.err
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 100"
$B --envs-per-gpu 512 > $OUT/tower_512_bench.json 2> /dev/null
$B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/obstacles_hard_512_bench.json 2> /dev/null
$B --scenario ObstaclesHard --envs-per-gpu 512 --pass-overlap off > $OUT/obstacles_hard_512_no_overlap_bench.json 2> /dev/null
$B --scenario ObstaclesHard > $OUT/obstacles_hard_1024_bench.json 2> /dev/null
for s in Collect HexMemory HexExplore Rearrange Sokoban Empty; do $B --scenario $s > $OUT/${s}_bench.json 2> /dev/null; done
$B --scenario Empty --steps 2000 > $OUT/Empty2000_bench.json 2> /dev/null
$B --scenario Mixed --obs 128 128 > $OUT/mixed_128_bench.json 2> /dev/null
$B --scenario Mixed --obs 64 64 > $OUT/mixed_64_bench.json 2> /dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*_bench*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]/1e6,2), "M", {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith("value_")}, "raster/tick %.1f us step/tick %.1f us" % (d["roofline"]["avg_launch_ms"]*1e3, d["roofline_physics"]["avg_launch_ms"]*1e3), d.get("host_enqueue_ms_per_step"))
    except Exception as e: print(f, "failed", e)
PY
