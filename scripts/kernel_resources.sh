#!/bin/bash
# VGPRs / SGPRs / scratch / LDS / occupancy of every kernel in a .hip file (hipcc remarks): scripts/kernel_resources.sh megaverse_amd/csrc/mv_raster.hip [filter] [extra flags]
f=${1:-megaverse_amd/csrc/mv_raster.hip}; filt=${2:-.}; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-pass-failed -Wno-unused-function "$@" \
  -Rpass-analysis=kernel-resource-usage -c -o /tmp/kr_$$.o "$f" 2>&1 | grep -E "Function Name|VGPRs:|SGPRs:|Occupancy|LDS Size|ScratchSize|error" \
  | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste -sd' ' | sed 's/Function Name: /\n/g' | grep -E "$filt" | c++filt | cut -c1-260
rm -f /tmp/kr_$$.o
