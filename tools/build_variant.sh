#!/bin/bash
# A/B library build with extra hipcc flags on SOME translation units (the rest as the product builds them):
#   tools/build_variant.sh <name> "<flags>" file1.hip file2.hip ...   ->  metro_pose3d_amd/ab/libmetro_<name>.so
#   tools/build_variant.sh agpr_slab "-DMETRO_AGPR_ACC=1" conv3x3_f16_slab.hip
# then: tools/ab_libs.sh 64 3 metro_pose3d_amd/libmetro_hip.so metro_pose3d_amd/ab/libmetro_agpr_slab.so
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../metro_pose3d_amd"
mkdir -p ab/$name
srcs=$(cd .. && python -c "from metro_pose3d_amd.build import SOURCES; print(' '.join(SOURCES))")
pids=()
for s in $srcs; do
  extra=""
  for f in "$@"; do [ "$f" = "$s" ] && extra="$flags"; done
  o=ab/$name/$(basename ${s%.*}).o
  if [ -z "$extra" ] && [ -f build/$(basename ${s%.*}).o ] && [ build/$(basename ${s%.*}).o -nt csrc/$s ] && [ build/$(basename ${s%.*}).o -nt csrc/metro_common.h ]; then
    cp build/$(basename ${s%.*}).o $o           # untouched unit: the product's object
    continue
  fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-function $extra -c csrc/$s -o $o &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libmetro_$name.so ab/$name/*.o
rm -rf ab/$name
ls -la ab/libmetro_$name.so
