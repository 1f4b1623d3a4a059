#!/usr/bin/env python3
"""Times the 3x3 stride-1 layers of blocks 2-4 alone (conv3x3_f16_slab.hip), e.g. under the knock-out builds of
tools/knockouts_r05.patch + tools/build_dbg_variants.sh conv3x3_f16_slab.hip SLAB_NO_DMA SLAB_NO_MFMA SLAB_NO_FRAG SLAB_NO_BARRIER:   python tools/slab_probe.py [batch]"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from metro_pose3d_amd import _lib
from tests import helpers as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = _lib.load(); dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(1)
P = H.ptr
for c, h, dil in ((512, 16, 2), (256, 16, 1), (128, 32, 1)):
    x = torch.randn((n, h, h, c), generator=g, device=dev).clamp_min(0).half()
    w = (torch.randn((c, 3, 3, c), generator=g, device=dev) * (2.0 / (9 * c)) ** 0.5).half()
    b = torch.randn(c, generator=g, device=dev) * 0.1
    o = torch.empty_like(x)
    d = H.conv_desc(n, h, c, h, c, 3, 1, dil, dil, relu=True, in_dtype=_lib.METRO_F16)
    fn = lambda: lib.metro_conv_f16(C.byref(d), P(x), P(w), P(b), None, None, None, P(o), None)
    lib.metro_kernel_notes(1)
    for _ in range(3):
        assert fn() == 0, lib.metro_last_error()
    kid = lib.metro_last_kernel_id().decode().split(' & ')[0]
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, bb in ev:
        a.record(); fn(); bb.record()
    torch.cuda.synchronize()
    us = float(np.median([a.elapsed_time(bb) for a, bb in ev])) * 1e3
    gf = 2.0 * n * h * h * 9 * c * c / 1e9
    print(f'batch {n}  3x3 {c:4d} -> {c:4d} on {h} x {h} rate {dil}  {us:8.1f} us  {gf / us * 1e-3:6.2f} PFLOP/s  {kid}')
