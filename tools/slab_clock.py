#!/usr/bin/env python3
"""Reads the cycle sums of a METRO_DBG_SLAB_CLOCK build of the tap-reuse kernel (patch -p0 < tools/knockouts_r06.patch; tools/build_dbg_variants.sh
conv3x3_f16_slab.hip SLAB_CLOCK; METRO_HIP_LIB=.../ab/libmetro_SLAB_CLOCK.so python tools/slab_clock.py [batch]): per tile, waves 0 and 5 --
cycles before the first step, in `s_waitcnt vmcnt + s_barrier` over all steps, in the K loop as a whole, in the epilogue."""
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
    o = torch.zeros_like(x)
    d = H.conv_desc(n, h, c, h, c, 3, 1, dil, dil, relu=True, in_dtype=_lib.METRO_F16)
    lib.metro_kernel_notes(1)
    for _ in range(3):
        assert lib.metro_conv_f16(C.byref(d), P(x), P(w), P(b), None, None, None, P(o), None) == 0, lib.metro_last_error()
    torch.cuda.synchronize()
    kid = lib.metro_last_kernel_id().decode().split(' & ')[0]
    tn = 512 if '512' in kid.split(',')[0] else 256
    tm = int(kid.split('<')[1].split('x')[0])
    rows = o.view(-1, c)                                   # [pixels][c]
    recs = []
    for m0 in range(0, rows.shape[0], tn):
        for n0 in range(0, c, tm):
            v = rows[m0, n0:n0 + 64].contiguous().view(torch.int64).cpu().numpy()
            if v[5] == 0x600DC10C and v[13] == 0x600DC10C:
                recs.append(np.concatenate([v[:5], v[8:13]]))
    r = np.array(recs, dtype=np.float64)
    steps = r[:, 4].mean()
    print(f'batch {n}  3x3 {c} -> {c} on {h}x{h} rate {dil}  {kid}: {len(recs)} tiles, {steps:.0f} steps per tile')
    for wv, off in ((0, 0), (5, 5)):
        pro, wait, loop, epi = r[:, off], r[:, off + 1], r[:, off + 2], r[:, off + 3]
        print(f'   wave {wv}: before the first step {pro.mean():8.0f} cycles | K loop {loop.mean():8.0f} = {loop.mean() / steps:6.0f} per step, of which '
              f'wait + barrier {wait.mean() / steps:6.0f} ({100 * wait.mean() / loop.mean():.0f} %) | epilogue {epi.mean():8.0f} | '
              f'tile {(pro + loop + epi).mean():8.0f} cycles')
