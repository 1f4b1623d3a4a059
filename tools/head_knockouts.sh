#!/bin/bash
# (the METRO_DBG_* sites live in tools/knockouts_r02_r04.patch, not in the product sources: see tools/build_dbg_variants.sh)
# Timing experiments on the 256-pixel K-halves head (results of the knock-out builds are garbage): which part costs what.
#   tools/head_knockouts.sh    (on the GPU box; the variants are built here first: tools/build_dbg_variants.sh head_f16.hip ...)
cd "$(dirname "$0")/.."
V="HD3_NO_X HD3_NO_W HD3_NO_X+HD3_NO_W HD3_NO_MFMA HD3_NO_SOFTMAX HD3_NO_X+HD3_NO_W+HD3_NO_SOFTMAX"
ls metro_pose3d_amd/dbg/libmetro_HD3_NO_X.so >/dev/null 2>&1 || tools/build_dbg_variants.sh head_f16.hip $V >/dev/null
echo "== product"; python tools/head_probe.py 2>&1 | grep "b64\|b256\|C4\|C5"
for v in $V; do
  echo "== $v"; METRO_HIP_LIB=$PWD/metro_pose3d_amd/dbg/libmetro_$v.so python tools/head_probe.py 2>&1 | grep "b64\|b256\|C4\|C5"
done
