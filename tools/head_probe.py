#!/usr/bin/env python3
"""Times metro_head_f16 (the one-launch volumetric head + finalize) on the head shapes of the BASELINE configs.
    python tools/head_probe.py            (METRO_HIP_LIB=... for knock-out builds: results are then garbage, times are not)"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from metro_pose3d_amd import ModelSpec, _lib
from tests import helpers as H

lib = _lib.load()
dev = torch.device('cuda', 0)
for name, spec, n in (('C2 b64', ModelSpec(50, 16, 'h36m'), 64), ('C2 b256', ModelSpec(50, 16, 'h36m'), 256), ('C3 b256 J19', ModelSpec(50, 16, 'many19'), 256),
                      ('C4 b32', ModelSpec(101, 8, 'many19'), 32), ('C5 b16', ModelSpec(50, 4, 'h36m'), 16)):
    side, k, c = spec.heatmap_side, 2048, spec.n_head_channels
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((n, side, side, k), generator=g, device=dev).half()
    w = (torch.randn((c, k), generator=g, device=dev) * 0.03).half()
    b = torch.zeros(c, dtype=torch.float32, device=dev)
    sc = torch.ones(k, dtype=torch.float16, device=dev)
    sh = torch.zeros(k, dtype=torch.float16, device=dev)
    cs = spec.to_c(_lib.METRO_PREC_F16)
    scratch = torch.empty(lib.metro_head_f16_scratch_bytes(n, side, spec.skeleton.n_head), dtype=torch.uint8, device=dev)
    poses = torch.empty((n, spec.skeleton.n_out, 3), dtype=torch.float32, device=dev)
    fn = lambda: lib.metro_head_f16(H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(sc), H.ptr(sh), n, k, C.byref(cs), H.ptr(scratch), None, H.ptr(poses), None)
    for _ in range(3):
        assert fn() == 0, lib.metro_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    gf = 2.0 * n * side * side * k * c / 1e9
    print(f'{name}: head + finalize {us:7.1f} us  ({gf / us * 1e3:5.0f} TFLOP/s, activations {n * side * side * k * 2 / 1e6:.0f} MB)', flush=True)
