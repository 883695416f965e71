#!/bin/bash
# tools/build_all_variant.sh NAME [extra hipcc flags]  ->  _ab/NAME.so : the WHOLE library rebuilt with extra flags (e.g. -DNRC_TILE_W_LOG2=4)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p _ab/$name
objs=""
for f in nerficg_amd/csrc/*.hip; do
  stem=$(basename "$f" .hip)
  extra=""
  case "$stem" in
    gs_raster) extra="-ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None";;
    ngp_march|adam|knn|gs_densify) extra="-ffp-contract=off";;
  ngp_net) extra="-fno-slp-vectorize";;
  esac
  /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wno-unused-function -Wno-unused-result -Wno-unused-value -fno-gpu-rdc -DNDEBUG $extra "$@" -c $f -o _ab/$name/$stem.o 2>/dev/null
  objs="$objs _ab/$name/$stem.o"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o _ab/$name.so $objs
echo _ab/$name.so
