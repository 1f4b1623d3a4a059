#!/usr/bin/env python3
"""Is the deep-K GEMM power-bound?  conv_gemm4w (pre-activated, through the C ABI) and torch.mm (hipBLASLt, bare GEMM) on the same shapes with
zero / N(0,1) / relu(N(0,1)) x He operands: the same instructions, other switching activity."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '.')
from metro_pose3d_amd import _lib
from tests import helpers as H
lib = _lib.load(); dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device=dev); g.manual_seed(0)
def timed(fn, reps=20):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3
for c_in, c_out in ((2048, 512), (1024, 2048), (1024, 256)):
    for kind in ('zeros', 'relu x He', 'N(0,1)'):
        if kind == 'zeros':
            x = torch.zeros((n, 16, 16, c_in), device=dev).half(); w = torch.zeros((c_out, c_in), device=dev).half()
        elif kind == 'N(0,1)':
            x = torch.randn((n, 16, 16, c_in), generator=g, device=dev).half(); w = torch.randn((c_out, c_in), generator=g, device=dev).half()
        else:
            x = torch.randn((n, 16, 16, c_in), generator=g, device=dev).clamp_min(0).half()
            w = (torch.randn((c_out, c_in), generator=g, device=dev) * (2.0 / c_in) ** 0.5).half()
        b = torch.zeros(c_out, dtype=torch.float32, device=dev)
        sc = torch.ones(c_in, device=dev).half(); sh = torch.zeros(c_in, device=dev).half()
        out = torch.empty((n, 16, 16, c_out), dtype=torch.float16, device=dev)
        d = H.conv_desc(n, 16, c_in, 16, c_out, 1, prologue=True)
        us = timed(lambda: lib.metro_conv_f16_gemm4w(C.byref(d), H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(sc), H.ptr(sh), None, H.ptr(out), 0, None, C.c_void_p(0)))
        x2 = x.view(-1, c_in); wt = w.t().contiguous()
        us2 = timed(lambda: torch.mm(x2, wt))
        gf = 2.0 * n * 256 * c_in * c_out * 1e-9
        print(f'batch {n} {c_in:4d} -> {c_out:4d}  {kind:10s}  conv_gemm4w {us:7.1f} us {gf / us:5.2f} PF | torch.mm (hipBLASLt) {us2:7.1f} us {gf / us2:5.2f} PF')
