#!/bin/bash
# scripts/build_variant.sh <name> "<extra hipcc flags>" [file.hip ...]: megaverse_amd/_variants/libmv_<name>.so = the library with the named sources
# (default mv_raster.hip) recompiled with the extra flags; run it with MV_LIB_PATH=...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; shift 2
SRCS=${@:-mv_raster.hip}
make -C $R/megaverse_amd/csrc -j8 > /dev/null
mkdir -p $R/megaverse_amd/_variants $R/megaverse_amd/csrc/_obj_var_$NAME
OBJS=""
for o in $R/megaverse_amd/csrc/_obj/*.o; do
  b=$(basename $o .o); skip=0
  for s in $SRCS; do [ "$b" = "$(basename $s .hip)" ] && skip=1; done
  [ $skip = 0 ] && OBJS="$OBJS $o"
done
for s in $SRCS; do
  b=$(basename $s .hip)
  EXTRA=""; [ "$b" = "mv_raster" ] && EXTRA="-fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-pass-failed $EXTRA $FLAGS -c -o $R/megaverse_amd/csrc/_obj_var_$NAME/$b.o $R/megaverse_amd/csrc/$s
  OBJS="$OBJS $R/megaverse_amd/csrc/_obj_var_$NAME/$b.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/megaverse_amd/_variants/libmv_$NAME.so $OBJS
echo built megaverse_amd/_variants/libmv_$NAME.so
