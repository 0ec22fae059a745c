#!/bin/bash
set -u
TAG=${1:-r04r}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python bench.py > $OUT/tower_bench.json 2> $OUT/tower_bench.err; echo "rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/tower_bench.json"))
print(round(d["value"]/1e6,2), "M", d["ms_per_step"], {k: round(v/1e6,2) for k,v in d.items() if k.startswith("value_")})
print("roofline", {k:v for k,v in d["roofline"].items() if k not in ("note","valu")})
print("physics", {k:v for k,v in d["roofline_physics"].items() if k!="note"})
cb=d["cpu_baseline"]; print("cpu", cb["value"], cb["cores"], cb["physics_scaling"], {k:(round(v["value"]),v["threads"],v["steps"],v["seconds"]) for k,v in cb.items() if isinstance(v,dict)})
PY
timeout 300 python -m pytest tests/test_distributed_gpu.py -x -q 2>&1 | tail -3
