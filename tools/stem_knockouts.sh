#!/bin/bash
# (the METRO_DBG_* sites live in tools/knockouts_r02_r04.patch, not in the product sources: see tools/build_dbg_variants.sh)
# Timing experiments on the rows stem (results of the knock-out builds are garbage): which part of a conv-row iteration costs what.
#   tools/stem_knockouts.sh   (variants built here first: tools/build_dbg_variants.sh stem_pool_f16.hip SP2_NO_MFMA ...)
cd "$(dirname "$0")/.."
V="SP2_NO_MFMA SP2_NO_POOL SP2_NO_CAST SP2_NO_DMA SP2_NO_MFMA+SP2_NO_POOL SP2_NO_POOL+SP2_NO_CAST+SP2_NO_DMA"
echo "== product: $(python tools/stem_probe.py 2 16 32 64 128 256 2>&1 | tail -1)"
for v in $V; do
  echo "== $v: $(METRO_HIP_LIB=$PWD/metro_pose3d_amd/dbg/libmetro_$v.so python tools/stem_probe.py 2 32 64 256 2>&1 | tail -1)"
done
