#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06i; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --steps 800 --warmup 100 --no-cpu-baseline --profile-steps 0 > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "%.2f M"%(d["value"]/1e6), " ".join("%s %.2f"%(k[6:],v/1e6) for k,v in d.items() if k.startswith("value_")), "enq", d.get("host_enqueue_ms_per_step_closed_loop_double_buffered"))
PY
}
run def X=1 --
run s2 MV_RASTER_SPLIT=2 --
run def_b X=1 --
run s2_b MV_RASTER_SPLIT=2 --
