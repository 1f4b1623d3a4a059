#!/usr/bin/env python3
"""Times conv3 + shortcut of blocks 3-4 (1x1, K -> 4 K + residual, 16 x 16 maps) alone: conv_pws.hip's skewed kernel and conv_pw64.hip's
lock-step one (metro_conv_b1_form(1)).   python tools/pws_probe.py [batch]"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from metro_pose3d_amd import _lib
from tests import helpers as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load(); dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(1)
P = H.ptr


def timeit(fn, reps=20):
    for _ in range(3):
        assert fn() == 0, lib.metro_last_error()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3


for k, h in ((512, 16), (256, 16), (256, 32)):
    x = (torch.randn((n, h, h, k), generator=g, device=dev)).clamp_min(0).half()
    w = (torch.randn((4 * k, k), generator=g, device=dev) * (2.0 / k) ** 0.5).half()
    b = torch.randn(4 * k, generator=g, device=dev) * 0.1
    r = torch.randn((n, h, h, 4 * k), generator=g, device=dev).half()
    o = torch.empty_like(r)
    d = H.conv_desc(n, h, k, h, 4 * k, 1, residual=True, res_h=h, in_dtype=_lib.METRO_F16)
    fn = lambda: lib.metro_conv_f16(C.byref(d), P(x), P(w), P(b), None, None, P(r), P(o), None)
    outs = []
    for classic in (1, 0):
        lib.metro_conv_b1_form(classic)
        lib.metro_kernel_notes(1)
        us = timeit(fn)
        kid = lib.metro_last_kernel_id().decode().split(' & ')[0]
        torch.cuda.synchronize()
        outs.append(o.clone())
        gf = 2.0 * n * h * h * k * 4 * k / 1e9
        mb = (x.numel() + 2 * r.numel()) * 2 / 1e6
        print(f'batch {n}  {k:4d} -> {4 * k:4d} on {h} x {h}  {us:8.1f} us  {gf / us * 1e3 / 1e3:6.2f} PFLOP/s  {mb / us / 1e3:5.2f} TB/s  {kid}')
    lib.metro_conv_b1_form(0)
    print('   same bits:', bool(torch.equal(outs[0], outs[1])))
