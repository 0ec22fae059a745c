#!/bin/bash
# the whole GPU suite + smoke on the committed tree
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08z_suite; mkdir -p $OUT; cd $R
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style.json 2> /dev/null; python -c "import json; d=json.load(open('$OUT/driver_style.json')); print(round(d['value']/1e6,2), d['roofline']['traffic'])"
