#!/bin/bash
# r08p: views by value up to 8 ticks, in device memory beyond: batched-call parity, the 20-step form, the headline
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08p; mkdir -p $OUT; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 1500 python -m pytest tests/test_pipelining_gpu.py tests/test_full_size_oracle_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs"
for i in 1 2 3; do $B --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style_$i.json 2>/dev/null; done
$B > $OUT/tower_bench.json 2>/dev/null
$B --envs-per-gpu 4096 > $OUT/tower_4096_bench.json 2>/dev/null
$B --envs-per-gpu 4096 --batch 8 > $OUT/tower_4096_b8_bench.json 2>/dev/null
$B --envs-per-gpu 512 --agents 4 --batch 8 > $OUT/tower_512x4_b8_bench.json 2>/dev/null
$B --envs-per-gpu 512 --agents 4 > $OUT/tower_512x4_bench.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]/1e6,2), "M", d["config"].get("ticks_per_call"), d["config"].get("ring_slots"))
    except Exception as e: print(f, "failed", e)
PY
