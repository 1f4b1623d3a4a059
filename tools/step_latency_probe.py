"""Per-K-step latency of the conv kernels on an otherwise idle GPU (tiny M, deep K)."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from metro_pose3d_amd import _lib
from tests import helpers as H
lib = _lib.load(); dev = torch.device('cuda', 0)
def run(name, n, h, c_in, c_out, k, dil, pad, reps=200):
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((n, h, h, c_in)).astype(np.float16)).to(dev)
    w = torch.from_numpy((rng.standard_normal((c_out, k, k, c_in)) * 0.02).astype(np.float16)).to(dev)
    b = torch.zeros(c_out, dtype=torch.float32, device=dev)
    out = torch.empty((n, h, h, c_out), dtype=torch.float16, device=dev)
    d = H.conv_desc(n, h, c_in, h, c_out, k, 1, dil, pad, in_dtype=_lib.METRO_F16)
    f = lambda: lib.metro_conv_f16(C.byref(d), H.ptr(x), H.ptr(w), H.ptr(b), None, None, None, H.ptr(out), None)
    for _ in range(10): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); us = (time.perf_counter() - t) / reps * 1e6
    steps = k * k * ((c_in + 63) // 64)
    print('%-34s M=%6d  %7.1f us  %3d steps  -> %.3f us/step' % (name, n * h * h, us, steps, us / steps))
run('1x1 K=2048 N=128, 1 image', 1, 16, 2048, 128, 1, 1, 0)
run('1x1 K=2048 N=128, 8 images', 8, 16, 2048, 128, 1, 1, 0)
run('1x1 K=2048 N=512, 64 images', 64, 16, 2048, 512, 1, 1, 0)
run('3x3 K=512 N=128, 1 image', 1, 16, 512, 128, 3, 2, 2)
run('3x3 K=512 N=512, 64 images', 64, 16, 512, 512, 3, 2, 2)
run('1x1 K=64 N=128, 1 image (launch floor)', 1, 16, 64, 128, 1, 1, 0)
