#!/usr/bin/env python3
"""tools/peak_probe.hip on the instruction forms of VERDICT r5 item 8: TFLOP/s and the clock held by a register-resident MFMA loop
for 32x32x16 / 16x16x32, accumulators in VGPRs / AGPRs, on zeros, N(0,1) and relu x He operands."""
import ctypes as C
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, 'tools', 'libmetro_probe.so'))
lib.metro_probe_mfma_f16_variant.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
names = {0: 'v_mfma_f32_32x32x16_f16, VGPR acc', 1: 'v_mfma_f32_16x16x32_f16, VGPR acc', 2: 'v_mfma_f32_32x32x16_f16, AGPR acc', 3: 'v_mfma_f32_16x16x32_f16, AGPR acc'}
kinds = {2: 'relu(N(0,1)) x He', 1: 'N(0,1)', 0: 'zeros'}
print('variant\tdata\tTFLOP/s\tsclk_MHz\tms')
for rep in range(2):
    for kind in (2, 1, 0):
        for v in range(4):
            tf, mhz, ms = C.c_double(), C.c_double(), C.c_double()
            rc = lib.metro_probe_mfma_f16_variant(kind, v, 2.0, C.byref(tf), C.byref(mhz), C.byref(ms))
            print(f'{names[v]}\t{kinds[kind]}\t{tf.value:.1f}\t{mhz.value:.0f}\t{ms.value:.2f}' if rc == 0 else f'{names[v]}\t{kinds[kind]}\tERROR')
