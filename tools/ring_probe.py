#!/usr/bin/env python3
"""Times the deep-K pre-activated 1x1 layers the ring kernel serves at batch 64 (conv_igemm_f16_dma.hip), e.g. under the knock-out builds
(tools/knockouts_r02_r04.patch + tools/build_dbg_variants.sh conv_igemm_f16_dma.hip SKIP_LOAD SKIP_MFMA SKIP_STORE):  python tools/ring_probe.py [batch]"""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '.')
from metro_pose3d_amd import _lib
from tests import helpers as H
lib = _lib.load(); dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device=dev); g.manual_seed(0)
for c_in, c_out in ((2048, 512), (1024, 256), (512, 1280)):
    x = torch.randn((n, 16, 16, c_in), generator=g, device=dev).half()
    w = (torch.randn((c_out, c_in), generator=g, device=dev) * (2.0 / c_in) ** 0.5).half()
    b = torch.zeros(c_out, dtype=torch.float32, device=dev)
    sc = (torch.rand(c_in, generator=g, device=dev) + 0.5).half(); sh = (torch.randn(c_in, generator=g, device=dev) * 0.3).half()
    pair = c_out == 1280
    out = torch.empty((n, 16, 16, 1024 if pair else c_out), dtype=torch.float16, device=dev)
    out2 = torch.empty((n, 16, 16, 256), dtype=torch.float16, device=dev)
    d = H.conv_desc(n, 16, c_in, 16, c_out, 1, prologue=True, in_dtype=_lib.METRO_F16)
    if pair:
        fn = lambda: lib.metro_conv_f16_pair(C.byref(d), H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(sc), H.ptr(sh), H.ptr(out), 1024, H.ptr(out2), None)
    else:
        fn = lambda: lib.metro_conv_f16(C.byref(d), H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(sc), H.ptr(sh), None, H.ptr(out), None)
    lib.metro_kernel_notes(1)
    for _ in range(3): assert fn() == 0, lib.metro_last_error()
    kid = lib.metro_last_kernel_id().decode().split(' & ')[0]
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, bb in ev:
        a.record(); fn(); bb.record()
    torch.cuda.synchronize()
    us = float(np.median([a.elapsed_time(bb) for a, bb in ev])) * 1e3
    print(f'batch {n} {c_in:4d} -> {c_out:4d} pro  {us:7.1f} us  {2.0*n*256*c_in*c_out/us*1e-9:6.2f} PFLOP/s  {kid}')
