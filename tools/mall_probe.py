#!/usr/bin/env python3
"""HBM vs Infinity-Cache rates of tools/peak_probe.hip's read / copy kernels over buffer sizes (16 MiB .. 2 GiB): does a tensor
that fits the 256 MiB Infinity Cache stream faster than one that does not?  (python tools/mall_probe.py, on the GPU box)"""
import ctypes as C
import os
import torch  # noqa: F401  (HIP runtime first)

lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libmetro_probe.so'))
lib.metro_probe_hbm.argtypes = [C.c_int, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
lib.metro_probe_mfma_f16.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
for kind, name in ((1, 'N(0,1)'), (2, 'relu x He'), (0, 'zeros')):
    tf, mhz, ms = C.c_double(), C.c_double(), C.c_double()
    assert lib.metro_probe_mfma_f16(kind, 3.0, C.byref(tf), C.byref(mhz), C.byref(ms)) == 0
    print(f'mfma f16 32x32x16, {name:10s}: {tf.value:7.1f} TFLOP/s  sclk {mhz.value:6.0f} MHz  ({ms.value:.2f} ms)')
for mib in (16, 32, 64, 128, 192, 256, 384, 512, 1024, 2048):
    row = []
    for kind in (0, 1):
        tb, us = C.c_double(), C.c_double()
        assert lib.metro_probe_hbm(kind, mib << 20, C.byref(tb), C.byref(us)) == 0
        row.append(f'{"read" if kind == 0 else "copy"} {tb.value:5.2f} TB/s ({us.value:7.1f} us)')
    print(f'{mib:5d} MiB: ' + '   '.join(row))
