#!/bin/bash
# Is a kernel's gfx950 code the same in the working tree's mv_raster.hip as in a git revision's?  (What profiles/pmc_traffic.json's source hash stands for: the
# counters were measured on THIS code.  A change elsewhere in the file moves the hash; this says whether the measured kernel moved.)
# usage: scripts/same_kernel_isa.sh [rev=HEAD] [mangled-name substring=the headline's one-launch pass]
REV=${1:-HEAD}; PAT=${2:-raster_fast_batch_kernelILi256ELb0ELi7ELi2ELi256}
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
git -C $R show $REV:megaverse_amd/csrc/mv_raster.hip > $T/mv_raster_rev.hip
cp $R/megaverse_amd/csrc/mv_raster.hip $T/mv_raster_tree.hip
for v in rev tree; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-pass-failed -fno-slp-vectorize \
    -mllvm -amdgpu-atomic-optimizer-strategy=None -I$R/megaverse_amd/csrc -S --cuda-device-only -o $T/$v.s $T/mv_raster_$v.hip || exit 2
  python $R/scripts/kernel_asm.py $T/$v.s $PAT dump | grep -v "^==" | grep -v "^\s*;" | sed 's/;.*//' > $T/$v.txt
done
if cmp -s $T/rev.txt $T/tree.txt; then echo "same code: $PAT ($(wc -l < $T/tree.txt) lines) in $REV and in the working tree"; rm -rf $T; exit 0; fi
echo "DIFFERENT: $PAT"; diff $T/rev.txt $T/tree.txt | head -20; rm -rf $T; exit 1
