#!/bin/bash
# r12y: mv_raster.hip built with LLVM's other AMDGPU scheduling strategies (scripts/build_variant.sh: max-ilp, max-memory-clause, iterative-minreg, iterative-ilp, no post-RA scheduler) against the product's build: headline, Empty, Collect, two runs each
set -u
TAG=${1:-r12y}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
for i in 1 2; do
for V in product maxilp memclause itminreg itilp nopostsched; do
  L=""; [ $V != product ] && L=$R/megaverse_amd/_variants/libmv_$V.so
  for S in TowerBuilding Empty Collect; do
    MV_LIB_PATH=$L $B --scenario $S > $OUT/${S}_${V}_$i.json 2> $OUT/${S}_${V}_$i.err
  done
done; done
for f in $OUT/*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M', 'raster/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3))
except Exception as e: print('$f', 'failed', e)
"; done
