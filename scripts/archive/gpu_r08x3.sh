#!/bin/bash
# r08x3: the settled scheme (a fresh status read-back every period, forced wait after max(32, 4 k) ticks): refill tests incl. the new run-ahead test, soak, rates
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08x3; mkdir -p $OUT; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 900 python -m pytest tests/test_refill_protocol_gpu.py tests/test_pipelining_gpu.py -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -40 $OUT/pytest.log | cut -c1-220
timeout 1200 python scripts/soak.py 6000 > $OUT/soak.log 2>&1; tail -3 $OUT/soak.log
B="timeout 300 python bench.py --no-cpu-baseline --profile-steps 0"
$B > $OUT/tower_bench.json 2> /dev/null
$B --no-extra-legs --scenario Empty > $OUT/Empty_bench.json 2> /dev/null
$B --no-extra-legs --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/obstacles_hard_512_bench.json 2> /dev/null
$B --no-extra-legs --scenario Collect > $OUT/Collect_bench.json 2> /dev/null
$B --no-extra-legs --scenario Sokoban > $OUT/Sokoban_bench.json 2> /dev/null
$B --no-extra-legs --steps 20 --warmup 5 > $OUT/driver_style_bench.json 2> /dev/null
for f in $OUT/*_bench.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')})
"; done
