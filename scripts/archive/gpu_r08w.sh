#!/bin/bash
# r08w: the status read-back's age bounded in ticks (one in flight): the soak (r08z's failed on a HexExplore starvation in the batched, overlapped run), refill tests, and
# what the tighter bound does to the rates (headline, 4096 envs, single ticks, ObstaclesHard 512, Collect, Mixed)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08w; mkdir -p $OUT; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 900 python -m pytest tests/test_refill_protocol_gpu.py tests/test_pipelining_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
timeout 1200 python scripts/soak.py 6000 > $OUT/soak.log 2>&1; tail -4 $OUT/soak.log
B="timeout 300 python bench.py --no-cpu-baseline --profile-steps 0"
$B > $OUT/tower_bench.json 2> /dev/null
$B --no-extra-legs --envs-per-gpu 4096 > $OUT/tower_4096_bench.json 2> /dev/null
$B --no-extra-legs --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/obstacles_hard_512_bench.json 2> /dev/null
$B --no-extra-legs --scenario Collect > $OUT/Collect_bench.json 2> /dev/null
$B --no-extra-legs --scenario Empty > $OUT/Empty_bench.json 2> /dev/null
$B --no-extra-legs --scenario Mixed --obs 64 64 > $OUT/mixed_64_bench.json 2> /dev/null
$B --no-extra-legs --steps 20 --warmup 5 > $OUT/driver_style_bench.json 2> /dev/null
for f in $OUT/*_bench.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')})
"; done
