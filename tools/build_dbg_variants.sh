#!/bin/bash
# Timing-experiment builds of the library with parts of the LDS-DMA conv kernel compiled out
# (results are garbage; use with METRO_HIP_LIB=metro_pose3d_amd/dbg/libmetro_<variant>.so bench.py --layer-report).
set -e
cd "$(dirname "$0")/../metro_pose3d_amd"
python -m metro_pose3d_amd.build >/dev/null 2>&1 || (cd .. && python -m metro_pose3d_amd.build >/dev/null)
mkdir -p dbg
for v in SKIP_STORE SKIP_LOAD SKIP_MFMA; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-function -DMETRO_DBG_$v -I../include \
      -c csrc/conv_igemm_f16_dma.hip -o dbg/dma_$v.o &
done
wait
for v in SKIP_STORE SKIP_LOAD SKIP_MFMA; do
  objs=$(ls build/*.o | grep -v conv_igemm_f16_dma.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o dbg/libmetro_$v.so $objs dbg/dma_$v.o
done
ls -la dbg/*.so
