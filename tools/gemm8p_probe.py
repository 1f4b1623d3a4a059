#!/usr/bin/env python3
"""Times metro_conv_f16_gemm8p, metro_conv_f16_gemm4w and torch.mm (hipBLASLt, bare GEMM) on the deep-K 1x1 shapes of blocks 3-4.
    python tools/gemm8p_probe.py [batch]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from metro_pose3d_amd import _lib
from tests import helpers as H

lib = _lib.load(); xlib = _lib.load_experimental()
dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(2048, 512), (1024, 256), (1024, 2048), (512, 1024), (1024, 512)]
if len(sys.argv) > 2:
    SHAPES = SHAPES[:int(sys.argv[2])]
rng = np.random.default_rng(0)
for c_in, c_out in SHAPES:
    x = torch.from_numpy(rng.standard_normal((n, 16, 16, c_in)).astype(np.float16)).to(dev)
    w = torch.from_numpy((rng.standard_normal((c_out, c_in)) * np.sqrt(2.0 / c_in)).astype(np.float16)).to(dev)
    b = torch.zeros(c_out, dtype=torch.float32, device=dev)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, c_in).astype(np.float16)).to(dev)
    sh = torch.from_numpy((rng.standard_normal(c_in) * 0.3).astype(np.float16)).to(dev)
    out = torch.empty((n, 16, 16, c_out), dtype=torch.float16, device=dev)
    gf = 2.0 * n * 256 * c_in * c_out / 1e9
    for pro in (False, True):
        d = H.conv_desc(n, 16, c_in, 16, c_out, 1, prologue=pro)
        o8, o4 = torch.empty_like(out), torch.empty_like(out)
        a8 = xlib.metro_conv_f16_gemm8p(C.byref(d), H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(sc) if pro else None, H.ptr(sh) if pro else None, None, H.ptr(o8), 0, None, C.c_void_p(0))
        a4 = lib.metro_conv_f16_gemm4w(C.byref(d), H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(sc) if pro else None, H.ptr(sh) if pro else None, None, H.ptr(o4), 0, None, C.c_void_p(0))
        od = torch.empty_like(out)
        ad = xlib.metro_conv_f16_gemm4d(C.byref(d), H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(sc) if pro else None, H.ptr(sh) if pro else None, None, H.ptr(od), 0, None, C.c_void_p(0))
        op = torch.empty_like(out)
        ap = xlib.metro_conv_f16_gemm4d_geo(C.byref(d), H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(sc), H.ptr(sh), None, H.ptr(op), 0, None, 3, C.c_void_p(0)) if pro else 0
        torch.cuda.synchronize()
        print('   bits equal to gemm8p:', a8, a4, ad, ap, bool(torch.equal(o8, o4)), bool(torch.equal(o8, od)), float((o8.float() - od.float()).abs().max()),
              'in-place form:', bool(torch.equal(o8, op)) if pro else None, float((o8.float() - op.float()).abs().max()) if pro else None)
        res = {}
        for name, fn in (('gemm8p', lambda: xlib.metro_conv_f16_gemm8p(C.byref(d), H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(sc) if pro else None,
                                                                     H.ptr(sh) if pro else None, None, H.ptr(out), 0, None, C.c_void_p(0))),
                         ('gemm4w', lambda: lib.metro_conv_f16_gemm4w(C.byref(d), H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(sc) if pro else None,
                                                                     H.ptr(sh) if pro else None, None, H.ptr(out), 0, None, C.c_void_p(0))),
                         ('gemm4d', lambda: xlib.metro_conv_f16_gemm4d(C.byref(d), H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(sc) if pro else None,
                                                                     H.ptr(sh) if pro else None, None, H.ptr(out), 0, None, C.c_void_p(0))),
                         ('gemm4p', (lambda: xlib.metro_conv_f16_gemm4d_geo(C.byref(d), H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(sc), H.ptr(sh), None,
                                                                                     H.ptr(out), 0, None, 3, C.c_void_p(0))) if pro else None),
                         ('hipblaslt', lambda: (torch.mm(x.view(-1, c_in), w.t(), out=out.view(-1, c_out)), 0)[1]),
                         ):
            if fn is None:
                continue
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
            res[name] = e0.elapsed_time(e1) / reps * 1e3
        if 'ref' not in res:
            pass
        print(f'n={n} {c_in:5d}->{c_out:5d} pro={int(pro)}  ' + '  '.join(f'{k} {v:7.1f} us {gf / v * 1e3:6.0f} TF/s' for k, v in res.items()), flush=True)
