#!/bin/bash
# Builds the library from the csrc/ of a git revision (default HEAD) into metro_pose3d_amd/dbg/libmetro_<name>.so:
# the baseline arm of an in-box A/B (tools/ab_libs.sh).      tools/build_ref_lib.sh [rev] [name] [extra hipcc flags...]
set -e
rev=${1:-HEAD}; name=${2:-base}; shift 2 || true
cd "$(dirname "$0")/.."
tmp=$(mktemp -d /tmp/metro_ref.XXXX)
git archive $rev metro_pose3d_amd/csrc metro_pose3d_amd/build.py include | tar -x -C $tmp
srcs=$(cd $tmp && python -c "
import re;t=open('metro_pose3d_amd/build.py').read();m=re.search(r'SOURCES\s*=\s*\[(.*?)\]',t,re.S);print(' '.join(re.findall(r'\'([^\']+)\'',m.group(1))))")
mkdir -p metro_pose3d_amd/dbg/$name
pids=()
for s in $srcs; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-function -I$tmp/include "$@" -c $tmp/metro_pose3d_amd/csrc/$s -o metro_pose3d_amd/dbg/$name/$(basename ${s%.*}).o &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o metro_pose3d_amd/dbg/libmetro_$name.so metro_pose3d_amd/dbg/$name/*.o
rm -rf $tmp
ls -la metro_pose3d_amd/dbg/libmetro_$name.so
