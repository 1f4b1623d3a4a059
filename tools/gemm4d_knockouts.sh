#!/bin/bash
# (the METRO_DBG_* sites live in tools/knockouts_r02_r04.patch, not in the product sources: see tools/build_dbg_variants.sh)
# Timing experiments on conv_gemm4d (results of the knock-out builds are garbage): which part of the K loop costs what.
#   tools/gemm4d_knockouts.sh    (on the GPU box; builds the variants first if hipcc is there)
cd "$(dirname "$0")/.."
V="G4D_BUFFER G4D_BUFFER+G4D_SAME_TILE G4D_SAME_TILE"
ls metro_pose3d_amd/dbg/libmetro_G4D_NO_DMA.so >/dev/null 2>&1 || tools/build_dbg_variants.sh conv_gemm4d.hip $V >/dev/null
echo "== product"; python tools/gemm8p_probe.py 256 1 2>&1 | grep "n=256"
for v in $V; do
  echo "== $v"; METRO_HIP_LIB=$PWD/metro_pose3d_amd/dbg/libmetro_$v.so python tools/gemm8p_probe.py 256 1 2>&1 | grep "n=256" | sed 's/gemm8p.*gemm4d/gemm4d/; s/hipblaslt.*//'
done
