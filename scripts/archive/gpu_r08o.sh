#!/bin/bash
# r08o: the 20-step form with 16 ticks per call by default -- what slowed it down?  and: more hardware queues for the double-buffered learner leg
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08o; mkdir -p $OUT; cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --gpus 1 --steps 20 --warmup 5"
for i in 1 2; do
$B > $OUT/d_default_$i.json 2>/dev/null
$B --batch 8 > $OUT/d_batch8_$i.json 2>/dev/null
MV_PIPE_BATCH=8 $B --batch 8 > $OUT/d_pipe8_batch8_$i.json 2>/dev/null
MV_BENCH_CALL_SCHEDULE=2,6 $B > $OUT/d_sched26_$i.json 2>/dev/null
MV_BENCH_CALL_SCHEDULE=4 $B > $OUT/d_sched4_$i.json 2>/dev/null
done
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --no-cpu-baseline > $OUT/q8_bench.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --steps 400 > $OUT/s400_bench.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]/1e6,2), "M", {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith("value_")}, d["config"].get("ticks_per_call"), d["config"].get("ring_slots"))
    except Exception as e: print(f, "failed", e)
PY
