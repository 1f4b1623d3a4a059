#!/usr/bin/env python3
"""What the ROCm libraries reach on the GEMM / conv shapes of blocks 3-4 (a yardstick for the hand-written kernels,
never part of the product): torch.mm -> hipBLASLt/rocBLAS fp16 for the 1x1 layers (bare GEMM: no pre-activation,
bias, ReLU or residual), torch conv2d -> MIOpen for the 3x3 layers (channels_last fp16).
    python tools/library_probe.py [batch]"""
import sys
import time

import torch
import torch.nn.functional as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda', 0)
torch.manual_seed(0)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


GEMMS = [('block3 conv1', 1024, 256), ('block3 conv3', 256, 1024), ('block3 pair', 512, 1024 + 256), ('block4 shortcut', 1024, 2048),
         ('block4 conv1 u1', 1024, 512), ('block4 conv1', 2048, 512), ('block4 conv3', 512, 2048), ('head', 2048, 136)]
m = n * 256
for name, k, c in GEMMS:
    x = torch.randn(m, k, device=dev, dtype=torch.float16)
    w = torch.randn(c, k, device=dev, dtype=torch.float16) * 0.02
    out = torch.empty(m, c, device=dev, dtype=torch.float16)
    wt = w.t()
    us = timed(lambda: torch.mm(x, wt, out=out))
    gf = 2.0 * m * k * c / 1e9
    print(f'n={n} GEMM {name:16s} M={m} K={k:4d} N={c:4d}: {us:7.1f} us {gf / us * 1e3:6.0f} TFLOP/s', flush=True)

CONVS = [('block3 conv2', 256, 16, 1), ('block4 conv2', 512, 16, 2), ('block2 conv2', 128, 32, 1), ('block1 conv2', 64, 64, 1)]
for name, c, side, rate in CONVS:
    x = torch.randn(n, c, side, side, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, c, 3, 3, device=dev, dtype=torch.float16) * 0.02).contiguous(memory_format=torch.channels_last)
    t0 = time.time()
    try:
        us = timed(lambda: F.conv2d(x, w, padding=rate, dilation=rate))
    except Exception as e:                                                       # MIOpen may lack a solver
        print(f'n={n} CONV {name}: {type(e).__name__}: {e}', flush=True)
        continue
    gf = 2.0 * n * side * side * c * c * 9 / 1e9
    print(f'n={n} CONV {name:16s} {c}ch {side}x{side} rate {rate}: {us:7.1f} us {gf / us * 1e3:6.0f} TFLOP/s  (setup {time.time() - t0:.1f} s)', flush=True)
