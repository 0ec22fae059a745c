#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r07k; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver.json 2> $OUT/driver.err; echo "rc=$?"
python - <<'PY'
import json,os
d=json.loads(open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r07k/driver.json").read().strip().splitlines()[-1])
print(round(d["value"]/1e6,2), d["roofline"]["traffic"], d["roofline"].get("valu",{}).get("valu_busy_frac_at_2.4GHz"), d["cpu_baseline"]["value"], d["config"]["first_calls"])
PY
timeout 120 python bench.py --scenario Sokoban --steps 800 --warmup 100 --no-cpu-baseline --no-extra-legs --profile-steps 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Sokoban', round(d['value']/1e6,2), d['config']['overlapped_passes'])"
