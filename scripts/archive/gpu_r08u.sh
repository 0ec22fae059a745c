#!/bin/bash
# r08u: the policy-in-the-loop legs (one gym; two halves on two streams) with the simulation stream at default and at high priority -- the legs never use that
# stream (a tick with a policy in the loop runs on the caller's), but a high-priority stream is a hardware queue of its own
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08u; mkdir -p $OUT; cd $R
for i in 1 2; do
  MV_SIM_PRIORITY=normal timeout 300 python bench.py --no-cpu-baseline --profile-steps 0 --steps 1000 > $OUT/normal_${i}_bench.json 2> /dev/null
  timeout 300 python bench.py --no-cpu-baseline --profile-steps 0 --steps 1000 > $OUT/high_${i}_bench.json 2> /dev/null
done
for f in $OUT/*_bench.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')})
"; done
