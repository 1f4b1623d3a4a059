#!/usr/bin/env python3
"""Times the conv3 + next conv1 launches of block1 (64 x 64 maps) alone, in the forms metro_forward / metro_forward_upto run:
    python tools/b1_probe.py [batch]            (METRO_HIP_LIB=... for knock-out builds: tools/build_dbg_variants.sh conv_b1.hip B1_NO_DMA ...)
unit 1: metro_conv_f16_next_proj with the sum stored (classic) / on chip (producer-consumer);
unit 2: metro_conv_f16_next_rebuild storing (classic) / sub-sampled copy only (producer-consumer)."""
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
h = 64
rnd = lambda *s, sc=1.0: (torch.randn(s, generator=g, device=dev) * sc)
t2 = rnd(n, h, h, 64).clamp_min(0).half(); t2p = rnd(n, h, h, 64).clamp_min(0).half(); x0 = rnd(n, h, h, 64).half()
w3 = rnd(256, 64, sc=0.18).half(); w3p = rnd(256, 64, sc=0.18).half(); wsc = rnd(256, 64, sc=0.18).half(); w1 = rnd(64, 256, sc=0.09).half()
b3, b3p, bsc, b1 = rnd(256, sc=0.1), rnd(256, sc=0.1), rnd(256, sc=0.1), rnd(64, sc=0.1)
ps, pb = (1 + rnd(64, sc=0.1)).half(), rnd(64, sc=0.1).half(); s2, sh2 = (1 + rnd(256, sc=0.1)).half(), rnd(256, sc=0.1).half()
out = torch.empty(n, h, h, 256, dtype=torch.float16, device=dev); sub = torch.empty(n, h // 2, h // 2, 256, dtype=torch.float16, device=dev)
out2 = torch.empty(n, h, h, 64, dtype=torch.float16, device=dev)
d = H.conv_desc(n, h, 64, h, 256, 1, in_dtype=_lib.METRO_F16)
P = H.ptr


def timeit(fn, reps=20):
    for _ in range(3):
        assert fn() == 0, lib.metro_last_error()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3


def proj(o):
    return lib.metro_conv_f16_next_proj(C.byref(d), P(t2), P(w3), P(b3), P(x0), P(wsc), P(bsc), P(ps), P(pb), P(o), P(w1), P(b1), P(s2), P(sh2), P(out2), 64, None)


def reb(o, osub):
    return lib.metro_conv_f16_next_rebuild(C.byref(d), P(t2), P(w3), P(b3), P(x0), P(wsc), P(bsc), P(ps), P(pb), P(t2p), P(w3p), P(b3p), P(o), P(osub), 0,
                                           P(w1), P(b1), P(s2), P(sh2), P(out2), 64, None)


lib.metro_kernel_notes(1)
rows = []
for name, fn, classic in (('unit 1, sum stored (classic)', lambda: proj(out), 1), ('unit 1, sum on chip (classic)', lambda: proj(None), 1),
                          ('unit 1, sum on chip', lambda: proj(None), 0),
                          ('unit 2, rebuilt, sum stored (classic)', lambda: reb(out, None), 1), ('unit 2, rebuilt, sub copy (classic)', lambda: reb(None, sub), 1),
                          ('unit 2, rebuilt, sum stored', lambda: reb(out, None), 0), ('unit 2, rebuilt, sub copy', lambda: reb(None, sub), 0)):
    lib.metro_conv_b1_form(classic)
    lib.metro_kernel_notes(1)
    us = timeit(fn)
    kid = lib.metro_last_kernel_id().decode().split(' & ')[0]
    print(f'batch {n}  {name:42s} {us:8.1f} us   {kid}')
lib.metro_conv_b1_form(0)
