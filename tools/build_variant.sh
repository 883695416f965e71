#!/bin/bash
# tools/build_variant.sh NAME FILE.hip [extra hipcc flags]  ->  _ab/NAME.so : the library with ONE translation unit rebuilt with extra flags
# (A/B experiments: NRC_LIB_PATH=_ab/NAME.so python tools/...).  Developer aid; _ab/ is not tracked.
set -e
cd "$(dirname "$0")/.."
name=$1; file=$2; shift 2
stem=$(basename "$file" .hip)
extra=""
case "$stem" in
  gs_raster) extra="-ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None";;
  ngp_march|adam|knn|gs_densify) extra="-ffp-contract=off";;
  ngp_net) extra="-fno-slp-vectorize";;
esac
mkdir -p _ab
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value -fno-gpu-rdc -DNDEBUG \
  $extra "$@" -c nerficg_amd/csrc/$stem.hip -o _ab/$name.$stem.o
objs=""
for o in nerficg_amd/lib/obj/*.o; do
  if [ "$(basename $o .o)" == "$stem" ]; then objs="$objs _ab/$name.$stem.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o _ab/$name.so $objs
echo _ab/$name.so
