"""Effective HBM rate of the memory-bound 1x1 convolutions of block1/block2 (batch 64) through metro_conv_f16."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, '.')
from metro_pose3d_amd import _lib
from tests import helpers as H
lib = _lib.load(); dev = torch.device('cuda', 0)
def run(name, n, h, c_in, c_out, pro, res, relu=False, reps=30):
    g = torch.Generator(device='cpu').manual_seed(0)
    x = torch.randn((n, h, h, c_in), generator=g).half().to(dev)
    w = (torch.randn((c_out, 1, 1, c_in), generator=g) * 0.05).half().to(dev)
    b = torch.zeros(c_out, dtype=torch.float32, device=dev)
    sc = torch.ones(c_in, dtype=torch.float16, device=dev) if pro else None
    sh = torch.zeros(c_in, dtype=torch.float16, device=dev) if pro else None
    r = torch.randn((n, h, h, c_out), generator=g).half().to(dev) if res else None
    out = torch.empty((n, h, h, c_out), dtype=torch.float16, device=dev)
    d = H.conv_desc(n, h, c_in, h, c_out, 1, prologue=pro, relu=relu, residual=res, res_h=h, in_dtype=_lib.METRO_F16)
    f = lambda: lib.metro_conv_f16(C.byref(d), H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(sc), H.ptr(sh), H.ptr(r), H.ptr(out), None)
    # rotate through a scratch buffer between launches so no launch finds its operands in the 256 MB MALL
    junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]; e1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    for i in range(reps):
        junk.add_(1)
        e0[i].record(); f(); e1[i].record()
    torch.cuda.synchronize()
    us = float(np.median([a.elapsed_time(b_) for a, b_ in zip(e0, e1)][5:])) * 1e3
    mb = (x.numel() + out.numel() + (r.numel() if res else 0)) * 2 / 1e6
    print('%-44s %7.1f us  %6.1f MB  %5.2f TB/s' % (name, us, mb, mb / us))
run('b1 shortcut 64->256 pro', 64, 64, 64, 256, True, False)
run('b1 shortcut 64->256 nopro', 64, 64, 64, 256, False, False)
run('b1 conv1 64->64 pro relu', 64, 64, 64, 64, True, False, True)
run('b1 conv1 256->64 pro relu', 64, 64, 256, 64, True, False, True)
run('b1 conv3 64->256 +res', 64, 64, 64, 256, False, True)
run('b2 conv3 128->512 +res', 64, 32, 128, 512, False, True)
run('b2 conv1 512->128 pro relu', 64, 32, 512, 128, True, False, True)
run('b3 conv3 256->1024 +res', 64, 16, 256, 1024, False, True)
