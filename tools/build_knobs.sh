#!/bin/bash
# A/B timing build: the library with -DMETRO_TUNING_KNOBS, i.e. the METRO_* tuning knobs (compile-time constants in the
# product build, metro_common.h: tuning_knob) read from the environment.
#   tools/build_knobs.sh   ->  metro_pose3d_amd/ab/libmetro_knobs.so
#   METRO_HIP_LIB=$PWD/metro_pose3d_amd/ab/libmetro_knobs.so METRO_CONV_C64=0 python bench.py --no-extras --cpu-seconds 0
set -e
cd "$(dirname "$0")/../metro_pose3d_amd"
mkdir -p ab/knobs   # ab/ travels to the GPU box (dbg/ is in .gpurunignore); delete it when done
srcs=$(python -c "from metro_pose3d_amd.build import SOURCES; print(' '.join(SOURCES))" 2>/dev/null || (cd .. && python -c "from metro_pose3d_amd.build import SOURCES; print(' '.join(SOURCES))"))
pids=()
for s in $srcs; do
  o=ab/knobs/$(basename ${s%.*}).o
  if [ ! -f $o ] || [ csrc/$s -nt $o ] || [ csrc/metro_common.h -nt $o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-function -DMETRO_TUNING_KNOBS -c csrc/$s -o $o &
    pids+=($!)
    if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libmetro_knobs.so ab/knobs/*.o
ls -la ab/libmetro_knobs.so
