#!/usr/bin/env python3
"""Yardstick for the HBM-bound launches: what pure-read and copy kernels of the ROCm stack reach on the same bytes (torch.sum /
torch.max / clone of a 285 MB fp32 tensor = the configs[4] soft-argmax volume).  Never part of the product.
    python tools/hbm_read_probe.py"""
import torch

dev = torch.device('cuda', 0)
x = torch.randn(128 * 64 * 64 * 136, device=dev)
mb = x.numel() * 4 / 1e6


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


y = torch.empty_like(x)
for name, fn, traffic in (('torch.sum (read)', lambda: x.sum(), 1), ('torch.max (read)', lambda: x.max(), 1),
                          ('x.view(-1, 136).sum(0) (read)', lambda: x.view(-1, 136).sum(0), 1),
                          ('y.copy_(x) (read + write)', lambda: y.copy_(x), 2), ('x.exp() (read + write)', lambda: torch.exp(x, out=y), 2)):
    us = timed(fn)
    print(f'{name:34s} {us:8.1f} us  {mb * traffic / us:6.2f} TB/s')
