#!/bin/bash
# Timing-experiment builds of the library with parts of one kernel compiled out (results are garbage).
# The knock-out sites (#ifdef METRO_DBG_*) are NOT in the product sources: apply them first, build, and revert --
#   patch -p0 < tools/knockouts_r02_r04.patch     (ring kernel, conv_gemm4w, conv_b1, head_f16, stem_pool_f16: rounds 2-4)
#   patch -p0 < tools/knockouts_r05.patch         (conv_pws, conv3x3_f16_slab: round 5)
#   ... tools/build_dbg_variants.sh ... ; git checkout metro_pose3d_amd/csrc
# Usage:
#   tools/build_dbg_variants.sh <source.hip> MACRO [MACRO...]   ->  metro_pose3d_amd/ab/libmetro_<MACRO>.so
# use with METRO_HIP_LIB=$PWD/metro_pose3d_amd/ab/libmetro_<MACRO>.so python bench.py --layer-report ...
set -e
cd "$(dirname "$0")/.."
python -m metro_pose3d_amd.build >/dev/null
cd metro_pose3d_amd
src=$1; shift
stem=$(basename "$src" .hip)
mkdir -p ab
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-function $(echo $v | tr "+" "\n" | sed "s/^/-DMETRO_DBG_/" | tr "\n" " ") -I../include \
      -c csrc/$stem.hip -o ab/${stem}_$v.o &
done
wait
for v in "$@"; do
  objs=$(ls build/*.o | grep -v "/$stem.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libmetro_$v.so $objs ab/${stem}_$v.o
done
ls ab/*.so
