#!/bin/bash
# bench.py's "alone" figures (roofline.alone, roofline_physics.alone: the same launches with pipelining off): headline, driver's form, MV_STEP_PIPE=1, Collect, ObstaclesHard 512, Empty
set -u
TAG=${1:-r12za}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs"
$B > $OUT/tower.json 2> $OUT/tower.err
$B --gpus 1 --steps 20 --warmup 5 > $OUT/driver.json 2> /dev/null
MV_STEP_PIPE=1 $B > $OUT/tower_step_pipe.json 2> /dev/null
$B --scenario Collect > $OUT/collect.json 2> /dev/null
$B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/oh512.json 2> /dev/null
$B --scenario Empty > $OUT/empty.json 2> /dev/null
for f in $OUT/*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r, p = d['roofline'], d['roofline_physics']
    print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M; pass per tick %.1f us; step per tick %.1f us (alone %.1f us, frac_survey %.3f)' % (r['avg_launch_ms']*1e3, p['avg_launch_ms']*1e3, p['alone']['avg_launch_ms']*1e3, p['alone']['frac_survey_bytes']))
except Exception as e: print('$f', 'failed', repr(e))
"; done
tail -2 $OUT/tower.err
