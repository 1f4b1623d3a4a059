#!/usr/bin/env python3
"""Times the fused conv2+conv3 launch of a block3 unit against the two separate launches.   python tools/conv23_probe.py [batch]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from metro_pose3d_amd import _lib
from tests import helpers as H

lib = _lib.load()
dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(0)
mk = lambda shape, s=1.0, dt=np.float16: torch.from_numpy((rng.standard_normal(shape) * s).astype(dt)).to(dev)
t1 = torch.relu(mk((n, 16, 16, 256)))
w2, b2 = mk((256, 3, 3, 256), 0.03), mk((256,), 0.1, np.float32)
w3, b3 = mk((1024, 1, 1, 256), 0.06), mk((1024,), 0.1, np.float32)
res = mk((n, 16, 16, 1024))
t2 = torch.empty((n, 16, 16, 256), dtype=torch.float16, device=dev)
out = torch.empty((n, 16, 16, 1024), dtype=torch.float16, device=dev)
flags = torch.zeros(2 * n, dtype=torch.int32, device=dev)
d2 = H.conv_desc(n, 16, 256, 16, 256, 3, 1, 1, 1, relu=True, in_dtype=_lib.METRO_F16)
d3 = H.conv_desc(n, 16, 256, 16, 1024, 1, residual=True, res_h=16, in_dtype=_lib.METRO_F16)
z = C.c_void_p(0)


def fused():
    assert lib.metro_conv_f16_conv2_conv3(C.byref(d2), H.ptr(t1), H.ptr(w2), H.ptr(b2), H.ptr(t2), C.byref(d3), H.ptr(w3), H.ptr(b3),
                                          H.ptr(res), H.ptr(out), H.ptr(flags), z) == 0, lib.metro_last_error()


def separate():
    assert lib.metro_conv_f16(C.byref(d2), H.ptr(t1), H.ptr(w2), H.ptr(b2), None, None, None, H.ptr(t2), z) == 0
    assert lib.metro_conv_f16(C.byref(d3), H.ptr(t2), H.ptr(w3), H.ptr(b3), None, None, H.ptr(res), H.ptr(out), z) == 0


for name, fn in (('separate', separate), ('fused', fused), ('separate', separate), ('fused', fused)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f'n={n} {name:9s} {e0.elapsed_time(e1) / reps * 1e3:7.1f} us per unit (conv2 + conv3)', flush=True)
